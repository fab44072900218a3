"""Parity checks shared by the CPU-emulation tests and the -m gpu tests.

Every function drives the product path through the C-ABI (via obs_color_monitor_amd.ScopeContext)
and compares with oracle/ on the same seeded input.  Integer tables must be bit-exact.
"""
import numpy as np
import pytest

DISTS = {1: "uniform", 2: "concentrated", 3: "flat", 4: "bars", 5: "alpha_holes"}
# (components) the reference UI exposes + single channels (BASELINE config 3) + odd masks
COMPONENTS = (0x07, 0x20, 0x50, 0x70, 0x01, 0x02, 0x04, 0x13)


DEFAULTS = {"vss.merge": 3, "vss.unroll": 0, "vss.fused_unroll": 0, "vss.hash": 0, "vss.fused_hash": -1, "vss.groups": -1, "his.unroll": 0, "wvs.wide": 1, "wvs.strip": 0, "wvs.deep": 1, "dock": 1, "wvs.persist": 1, "wvs.per_cu": 1, "fold": 1, "vss.sparse": 1, "vss.fold": 1, "vss.block": 0, "his.block": 0,
            "nt": 1, "slices": 0, "trio.strip": 0, "vss.rle": 0}


class options:
    """with options(ctx, **{"vss.merge": 0}): ...  -- tunables restored to the defaults afterwards."""

    def __init__(self, ctx, **kv):
        self.ctx, self.kv = ctx, kv

    def __enter__(self):
        for k, v in self.kv.items():
            self.ctx.set_option(k, v)
        return self.ctx

    def __exit__(self, *exc):
        for k in self.kv:
            self.ctx.set_option(k, DEFAULTS[k])


def frames(oracle, w, h, dist, linesize=None, seed=0, colorspace=2):
    rgb = oracle.fill_frame(w, h, dist, seed, linesize)
    yuv = oracle.rgb_to_yuv(rgb, w, colorspace)
    return rgb, yuv


def assert_same(got, want, what):
    if not np.array_equal(got, want):
        bad = np.argwhere(np.asarray(got) != np.asarray(want))
        first = tuple(bad[0])
        raise AssertionError(f"{what}: {len(bad)} mismatching cells, first at {first}: "
                             f"got {np.asarray(got)[first]} want {np.asarray(want)[first]}")


def check_k0(ctx, oracle, w, h, dist, colorspace, linesize=None):
    rgb = oracle.fill_frame(w, h, dist, 0, linesize)
    want = oracle.rgb_to_yuv(rgb, w, colorspace)
    got = ctx.convert_rgb_yuv(rgb, w, colorspace)
    assert_same(got[:, : 4 * w], want[:, : 4 * w], f"K0 {w}x{h} d{dist} cs{colorspace}")


def check_vectorscope(ctx, oracle, w, h, dist, linesize=None, fuse=False, colorspace=2):
    rgb, yuv = frames(oracle, w, h, dist, linesize, colorspace=colorspace)
    want = oracle.vectorscope(yuv, w)
    if fuse:
        got, st = ctx.vectorscope(rgb=rgb, width=w, colorspace=colorspace, fuse_k0=True)
    else:
        got, st = ctx.vectorscope(yuv=yuv, width=w)
    assert st == 0
    assert_same(got, want, f"vectorscope {w}x{h} d{dist} ls{linesize} fuse{fuse}")


def screen_frame(w, h, seed, detail=0.03, rects=12, linesize=None, grey=False, alpha_noise=False):
    """Screen-capture-like content for the absorb mode of the vectorscope kernels: a flat background, flat rectangles, a sprinkle of
    single-texel detail; grey=True: every colour a grey (the chroma never changes while the RGB does); alpha_noise: random alpha
    bytes (the vectorscope ignores alpha, src/vectorscope.c:217-238: runs must not break on it)."""
    rng = np.random.default_rng(seed)

    def colour():
        c = rng.integers(0, 256, 3, dtype=np.uint8)
        return np.array([c[0]] * 3, np.uint8) if grey else c

    img = np.empty((h, w, 4), np.uint8)
    img[..., :3] = colour()
    for _ in range(rects):
        x0, y0 = int(rng.integers(0, max(w - 1, 1))), int(rng.integers(0, max(h - 1, 1)))
        ww, hh = int(rng.integers(1, max(w // 2, 2))), int(rng.integers(1, max(h // 2, 2)))
        img[y0:y0 + hh, x0:x0 + ww, :3] = colour()
    mask = rng.random((h, w)) < detail
    n = int(mask.sum())
    d = rng.integers(0, 256, (n, 3), dtype=np.uint8)
    img[..., :3][mask] = np.repeat(d[:, :1], 3, axis=1) if grey else d
    img[..., 3] = rng.integers(0, 256, (h, w), dtype=np.uint8) if alpha_noise else 255
    ls = linesize or 4 * w
    out = np.zeros((h, ls), np.uint8)
    out[:, : 4 * w] = img.reshape(h, 4 * w)
    return out


def check_vectorscope_rgb_frame(ctx, oracle, rgb, w, colorspace=2, what=""):
    """The fused-K0 vectorscope of an arbitrary RGB plane against oracle K0 + oracle K1."""
    want = oracle.vectorscope(oracle.rgb_to_yuv(rgb, w, colorspace), w)
    got, st = ctx.vectorscope(rgb=rgb, width=w, colorspace=colorspace, fuse_k0=True)
    assert st == 0
    assert_same(got, want, f"fused vectorscope {w}x{rgb.shape[0]} cs{colorspace} {what}")


def check_fused_screen_content(ctx, oracle, sizes, forced):
    """The fused vectorscope on screen-like content -- flat areas, rectangles, sprinkled detail, greys (the chroma never changes
    while the RGB does), random alpha -- and on the oracle's distributions.  forced 0: the modes by wave vote (the default);
    1: run-length mode on ANY content (vss.rle = 1: on noise every cell is rough and goes through the packed road, three
    rounds per wave-cell; the mode lives in the kernel's batch form, vss.fused_unroll 4); 2: the per-lane absorb mode on any
    content (vss.merge 2)."""
    opts = {1: {"vss.rle": 1}, 2: {"vss.merge": 2}}.get(forced, {})
    with options(ctx, **opts):
        for (w, h, ls) in sizes:
            for seed, kw in enumerate(({}, {"detail": 0.0}, {"detail": 0.3}, {"grey": True}, {"alpha_noise": True, "detail": 0.01},
                                       {"rects": 200, "detail": 0.0})):
                rgb = screen_frame(w, h, 77 + seed, linesize=ls, **kw)
                check_vectorscope_rgb_frame(ctx, oracle, rgb, w, 1 + seed % 2, f"screen {kw} forced={forced}")
            for dist in (1, 2, 3, 4, 5):
                check_vectorscope(ctx, oracle, w, h, dist, ls, fuse=True, colorspace=2)


def check_reference_colour_targets(ctx):
    """The library against the vectorscope graticule's colour targets, the one table of expected chroma the reference holds
    (src/vectorscope.c:283-316; data and rule in tests/test_oracle.py): the 100 % and 75 % colour bars through
    scope_convert_rgb_yuv, and through the fused-K0 and the staged vectorscope -- the twelve bins that light up lie on the
    twelve targets of the matrix in use.  No oracle call."""
    import test_oracle as to

    bars = np.repeat(np.repeat(to.colour_bars_plane().reshape(1, 12, 4), 16, axis=1), 8, axis=0).reshape(8, 12 * 16 * 4)  # 192 x 8 texels
    bars = np.ascontiguousarray(bars)
    w = 12 * 16
    to.check_bars_against_the_reference_targets(
        lambda cs: [(int(p[0]), int(p[2])) for p in ctx.convert_rgb_yuv(bars, w, cs)[0].reshape(w, 4)[::16]])
    for cs in (1, 2):
        tables = [ctx.vectorscope(rgb=bars, width=w, colorspace=cs, fuse_k0=True), ctx.vectorscope(yuv=ctx.convert_rgb_yuv(bars, w, cs), width=w)]
        for vs, st in tables:
            assert st == 0
            rows, cols = np.nonzero(vs)
            lit = {(int(c), 255 - int(r)) for r, c in zip(rows, cols)}
            assert len(lit) == 12 and all(vs[255 - v, u] == 128 for (u, v) in lit)  # 16 x 8 texels per bar
            for t in to.REF_COLOUR_TARGETS[cs]:
                assert any(max(abs(u - t[0]), abs(v - t[1])) <= 2 for (u, v) in lit), (cs, t, sorted(lit))


def check_plain_screen_content(ctx, oracle, sizes, forced=0):
    """The plain (staged-plane) vectorscope on the Y'CbCr planes of screen-like content: windows of cells where some lanes are
    flat and some are not take the run-length form of the absorb mode (scope_kernels.hip, keys_rle; the batch form, vss.unroll 4),
    all-flat and all-rough windows the forms they took before.  forced 1: that form on any content (vss.rle = 1)."""
    with options(ctx, **({"vss.rle": 1} if forced else {})):
        _plain_screen_content(ctx, oracle, sizes, forced)


def _plain_screen_content(ctx, oracle, sizes, forced):
    for (w, h, ls) in sizes:
        for seed, kw in enumerate(({}, {"detail": 0.0}, {"detail": 0.3}, {"grey": True}, {"alpha_noise": True, "detail": 0.01},
                                   {"rects": 200, "detail": 0.0}, {"rects": 60, "detail": 0.1})):
            rgb = screen_frame(w, h, 177 + seed, linesize=ls, **kw)
            yuv = oracle.rgb_to_yuv(rgb, w, 1 + seed % 2)
            got, st = ctx.vectorscope(yuv=yuv, width=w)
            assert st == 0
            assert_same(got, oracle.vectorscope(yuv, w), f"plain vectorscope {w}x{h} ls{ls} screen {kw} forced={forced}")
        for dist in (1, 2, 3, 4, 5):
            check_vectorscope(ctx, oracle, w, h, dist, ls, fuse=False)


def check_waveform(ctx, oracle, w, h, dist, components, linesize=None, fuse=False, colorspace=2):
    rgb, yuv = frames(oracle, w, h, dist, linesize, colorspace=colorspace)
    want = oracle.waveform(rgb, yuv, w, components)
    if fuse:
        got, st = ctx.waveform(rgb=rgb, yuv=None, width=w, components=components, colorspace=colorspace, fuse_k0=True)
    else:
        got, st = ctx.waveform(rgb=rgb, yuv=yuv, width=w, components=components)
    assert st == 0
    assert_same(got, want, f"waveform {w}x{h} d{dist} c{components:#x} ls{linesize} fuse{fuse}")


def check_histogram_counts(ctx, oracle, w, h, dist, components, linesize=None, fuse=False, colorspace=2):
    rgb, yuv = frames(oracle, w, h, dist, linesize, colorspace=colorspace)
    want = oracle.histogram_counts(rgb, yuv, w, components)
    if fuse:
        got, st = ctx.histogram_counts(rgb=rgb, yuv=None, width=w, components=components, colorspace=colorspace,
                                       fuse_k0=True)
    else:
        got, st = ctx.histogram_counts(rgb=rgb, yuv=yuv, width=w, components=components)
    assert st == 0
    assert_same(got, want, f"histogram {w}x{h} d{dist} c{components:#x} ls{linesize} fuse{fuse}")


def check_histogram_full(ctx, oracle, w, h, dist, components, logscale, level_fixed, level_ratio):
    """K3 + K3b.  Counts and hi_max are integers (bit-exact); the float table uses the same libm
    logf on the same host as the oracle, so it is compared bit-exactly as well."""
    rgb, yuv = frames(oracle, w, h, dist)
    want_f, want_hi = oracle.histogram(rgb, yuv, w, components, logscale, level_fixed, level_ratio, (7, 8, 9))
    got_f, got_hi, st = ctx.histogram(rgb=rgb, yuv=yuv, width=w, components=components, logscale=logscale,
                                      level_fixed=level_fixed, level_ratio=level_ratio, hi_max_in=(7, 8, 9))
    assert st == 0
    assert_same(got_hi, want_hi, "hi_max")
    assert_same(got_f.view(np.uint32), want_f.view(np.uint32), f"histogram float table c{components:#x} log{logscale}")


def check_batches(ctx, oracle, w, h, n):
    """n frames of mixed content in one batched call each."""
    rgbs, yuvs = [], []
    for i in range(n):
        r, y = frames(oracle, w, h, 1 + i % 5, seed=1000 + i)
        rgbs.append(r)
        yuvs.append(y)
    got = ctx.vectorscope_batch_host(yuvs, w, planes_are_rgb=False)
    for i in range(n):
        assert_same(got[i], oracle.vectorscope(yuvs[i], w), f"vectorscope batch frame {i}")
    got = ctx.vectorscope_batch_host(rgbs, w, planes_are_rgb=True, colorspace=2)
    for i in range(n):
        assert_same(got[i], oracle.vectorscope(yuvs[i], w), f"vectorscope batch (fused K0) frame {i}")
    got, st = ctx.waveform_batch_host(rgbs, w, 0x07, planes_are_rgb=True)
    assert st == 0
    for i in range(n):
        assert_same(got[i], oracle.waveform(rgbs[i], None, w, 0x07), f"waveform batch frame {i}")
    got, st = ctx.waveform_batch_host(rgbs, w, 0x20, planes_are_rgb=True, colorspace=2)
    assert st == 0
    for i in range(n):
        assert_same(got[i], oracle.waveform(None, yuvs[i], w, 0x20), f"waveform batch luma (fused K0) frame {i}")
    got, st = ctx.histogram_counts_batch_host(rgbs, w, 0x07, planes_are_rgb=True)
    assert st == 0
    for i in range(n):
        assert_same(got[i], oracle.histogram_counts(rgbs[i], None, w, 0x07), f"histogram batch frame {i}")


def check_error_behaviour(ctx, oracle):
    """NULL planes leave the caller's table untouched (the reference's callbacks return early:
    vectorscope.c:252, waveform.c:276-281, histogram.c:436-441); masks that select no plane give
    a zeroed table (waveform.c:225-234, histogram.c:363-373)."""
    import obs_color_monitor_amd as ocm

    w, h = 32, 8
    rgb, yuv = frames(oracle, w, h, 1)
    out, st = ctx.vectorscope(yuv=None, rgb=rgb, width=w)  # no yuv, no FUSE flag
    assert st == ocm.ENODATA and (out == 0xEE).all()
    out, st = ctx.waveform(rgb=None, yuv=yuv, width=w, components=0x07)
    assert st == ocm.ENODATA and (out == 0xEE).all()
    out, st = ctx.waveform(rgb=rgb, yuv=None, width=w, components=0x20)
    assert st == ocm.ENODATA and (out == 0xEE).all()
    out, st = ctx.histogram_counts(rgb=None, yuv=yuv, width=w, components=0x07)
    assert st == ocm.ENODATA and (out == 0xEEEEEEEE).all()
    out, st = ctx.waveform(rgb=rgb, yuv=yuv, width=w, components=0x08)
    assert st == 0 and (out == 0).all()
    out, st = ctx.histogram_counts(rgb=rgb, yuv=yuv, width=w, components=0x80)
    assert st == 0 and (out == 0).all()
    # the full histogram with an unselectable mask: zero table, hi_max untouched
    f, hi, st = ctx.histogram(rgb=rgb, yuv=yuv, width=w, components=0x00, hi_max_in=(7, 8, 9))
    assert st == 0 and (f.view(np.uint32) == 0).all() and tuple(hi) == (7, 8, 9)
    # linesize smaller than the row is rejected
    bad = np.zeros((4, 16), dtype=np.uint8)
    with __import__("pytest").raises(ocm.ScopeError):
        ctx.vectorscope(yuv=bad, width=8)


def check_device_path(ctx, oracle, to_dev, to_host, w, h, linesize, n=2, offset=0):
    """Device-resident planes/tables through the batched entry points (IN_DEVICE | OUT_DEVICE).
    to_dev(np.ndarray) -> (handle, address); to_host(handle) -> np.ndarray.  `offset` shifts the
    plane start by that many bytes (4-byte aligned but not 16) to force the unaligned pixel path."""
    import numpy as np

    rgbs, yuvs = [], []
    for i in range(n):
        r, y = frames(oracle, w, h, 1 + i % 5, linesize, seed=77 + i)
        rgbs.append(r)
        yuvs.append(y)

    def upload(fr):
        flat = np.concatenate([np.zeros(offset, np.uint8)] + [f.reshape(-1) for f in fr])
        hnd, addr = to_dev(flat)
        return hnd, [addr + offset + i * h * linesize for i in range(n)]

    hy, py = upload(yuvs)
    hr, pr = upload(rgbs)
    by = ctx._batch(py, linesize, w, h, False, 2)
    br = ctx._batch(pr, linesize, w, h, True, 2)

    out_h, out_a = to_dev(np.full(n * 65536, 0xEE, np.uint8))
    ctx.vectorscope_batch_dev(by, out_a)
    got = to_host(out_h).reshape(n, 256, 256)
    for i in range(n):
        assert_same(got[i], oracle.vectorscope(yuvs[i], w), f"device vectorscope frame {i}")
    ctx.vectorscope_batch_dev(br, out_a)
    got = to_host(out_h).reshape(n, 256, 256)
    for i in range(n):
        assert_same(got[i], oracle.vectorscope(yuvs[i], w), f"device vectorscope fused frame {i}")

    # the multi-GPU wire format straight from the kernels: two bins per 32-bit word, lo | hi << 16
    pk_h, pk_a = to_dev(np.full(n * 65536 * 2, 0xEE, np.uint8))
    for batch, what in ((by, "packed"), (br, "packed fused")):
        ctx.vectorscope_batch_dev(batch, pk_a, packed16=True)
        got = to_host(pk_h).view(np.uint32).reshape(n, 32768)
        for i in range(n):
            t = oracle.vectorscope(yuvs[i], w).reshape(-1).astype(np.uint32)
            assert_same(got[i], t[0::2] | (t[1::2] << 16), f"device vectorscope {what} frame {i}")

    out_h, out_a = to_dev(np.full(n * 256 * w * 4, 0xEE, np.uint8))
    ctx.waveform_batch_dev(br, 0x07, out_a)
    got = to_host(out_h).reshape(n, 256, w, 4)
    for i in range(n):
        assert_same(got[i], oracle.waveform(rgbs[i], None, w, 0x07), f"device waveform frame {i}")
    ctx.waveform_batch_dev(by, 0x70, out_a)
    got = to_host(out_h).reshape(n, 256, w, 4)
    for i in range(n):
        assert_same(got[i], oracle.waveform(None, yuvs[i], w, 0x70), f"device waveform yuv frame {i}")

    out_h, out_a = to_dev(np.full(n * 4096, 0xEE, np.uint8))
    ctx.histogram_counts_batch_dev(br, 0x07, out_a)
    got = to_host(out_h).view(np.uint32).reshape(n, 256, 4)
    for i in range(n):
        assert_same(got[i], oracle.histogram_counts(rgbs[i], None, w, 0x07), f"device histogram frame {i}")
    return hy, hr


def check_surface_fanout(ctx, oracle, w, h, linesize=None):
    """scope_surface_begin/end: the dock's three scopes on one staged surface, planes uploaded once."""
    import obs_color_monitor_amd as ocm

    rgb, yuv = frames(oracle, w, h, 5, linesize=linesize)

    def three(colorspace=2):
        v, st = ctx.vectorscope(rgb=rgb, yuv=yuv, width=w, colorspace=colorspace)
        assert st == 0
        assert_same(v, oracle.vectorscope(yuv, w), "fan-out vectorscope")
        wv, st = ctx.waveform(rgb=rgb, yuv=yuv, width=w, components=0x07, colorspace=colorspace)
        assert_same(wv, oracle.waveform(rgb, yuv, w, 0x07), "fan-out waveform RGB")
        wl, st = ctx.waveform(rgb=rgb, yuv=yuv, width=w, components=0x20, colorspace=colorspace)
        assert_same(wl, oracle.waveform(rgb, yuv, w, 0x20), "fan-out waveform luma")
        hc, st = ctx.histogram_counts(rgb=rgb, yuv=yuv, width=w, components=0x70, colorspace=colorspace)
        assert_same(hc, oracle.histogram_counts(rgb, yuv, w, 0x70), "fan-out histogram YUV")

    ctx.surface_begin()
    three()
    assert ctx.surface_end() == (2, 2)  # RGB and YUV uploaded once each, two uploads saved
    ctx.surface_begin(fuse_k0=True)  # YUV never uploaded: converted from the cached RGB plane on the fly
    three()
    assert ctx.surface_end() == (1, 3)
    three()  # outside begin/end nothing is cached
    with pytest.raises(ocm.ScopeError):
        ctx.surface_end()
    ctx.surface_begin()
    with pytest.raises(ocm.ScopeError):
        ctx.surface_begin()
    v, st = ctx.vectorscope(yuv=yuv, width=w)
    other = oracle.fill_frame(w + 1, h, 1, 3)
    with pytest.raises(ocm.ScopeError):  # one geometry per surface
        ctx.histogram_counts(rgb=other, yuv=None, width=w + 1, components=0x07)
    assert ctx.surface_end() == (1, 0)


def check_waveform_histogram(ctx, oracle, w, h, dist, linesize=None, colorspace=2):
    """scope_waveform_histogram: K2 + K3 of one surface, one pass where the masks allow (same plane kind, histogram
    channels a subset of the waveform's), two passes otherwise -- always the tables of the two separate scopes."""
    rgb, yuv = frames(oracle, w, h, dist, linesize, colorspace=colorspace)
    for wc, hc in ((0x07, 0x07), (0x70, 0x70), (0x70, 0x20), (0x07, 0x02), (0x05, 0x04), (0x13, 0x01),
                   (0x20, 0x70), (0x07, 0x70), (0x50, 0x07), (0x01, 0x06)):
        wv, cnt, st = ctx.waveform_histogram(rgb=rgb, yuv=yuv, width=w, wvs_components=wc, his_components=hc, colorspace=colorspace)
        assert st == 0
        assert_same(wv, oracle.waveform(rgb, yuv, w, wc), f"waveform+histogram: waveform {wc:#x}/{hc:#x} {w}x{h} d{dist}")
        assert_same(cnt, oracle.histogram_counts(rgb, yuv, w, hc), f"waveform+histogram: counts {wc:#x}/{hc:#x} {w}x{h} d{dist}")
    # fused K0: only the RGB plane exists
    for wc, hc in ((0x70, 0x70), (0x20, 0x20), (0x70, 0x50), (0x07, 0x07)):
        wv, cnt, st = ctx.waveform_histogram(rgb=rgb, yuv=None, width=w, wvs_components=wc, his_components=hc,
                                             colorspace=colorspace, fuse_k0=True)
        assert st == 0
        assert_same(wv, oracle.waveform(rgb, yuv, w, wc), f"waveform+histogram fused K0: waveform {wc:#x}/{hc:#x}")
        assert_same(cnt, oracle.histogram_counts(rgb, yuv, w, hc), f"waveform+histogram fused K0: counts {wc:#x}/{hc:#x}")


def check_surface_waveform_feeds_histogram(ctx, oracle, w, h):
    """Inside scope_surface_begin/end the waveform pass leaves the histogram of its channels behind: the dock's
    histogram callback on the same surface costs neither an upload nor a kernel."""
    rgb, yuv = frames(oracle, w, h, 5)
    ctx.surface_begin()
    wv, st = ctx.waveform(rgb=rgb, yuv=yuv, width=w, components=0x07)
    assert st == 0
    assert_same(wv, oracle.waveform(rgb, yuv, w, 0x07), "dock waveform")
    for hc in (0x07, 0x02, 0x05):  # every subset of the waveform's channels comes from the same side table
        cnt, st = ctx.histogram_counts(rgb=rgb, yuv=yuv, width=w, components=hc)
        assert st == 0
        assert_same(cnt, oracle.histogram_counts(rgb, yuv, w, hc), f"dock histogram {hc:#x} from the waveform pass")
    f, hi, st = ctx.histogram(rgb=rgb, yuv=yuv, width=w, components=0x07, logscale=1)
    want_f, want_hi = oracle.histogram(rgb, yuv, w, 0x07, 1, 0, 0, (0, 0, 0))
    assert st == 0 and np.array_equal(hi, want_hi) and np.array_equal(f.view(np.uint32), want_f.view(np.uint32))
    cnt, st = ctx.histogram_counts(rgb=rgb, yuv=yuv, width=w, components=0x70)  # another plane: its own pass
    assert_same(cnt, oracle.histogram_counts(rgb, yuv, w, 0x70), "dock histogram YUV")
    # the side table belongs to the geometry it was counted on: the same plane pointer with another width is NOT served
    # from it (it used to be, with SCOPE_OK and stale counts) but refused like any geometry change inside begin/end
    try:
        _, st = ctx.histogram_counts(rgb=rgb, yuv=yuv, width=w // 2, components=0x07)
    except Exception as e:  # noqa: BLE001
        st = getattr(e, "status", None)
    assert st == -22, f"histogram of another geometry inside scope_surface_begin/end: status {st}"
    assert ctx.surface_end() == (2, 4)  # RGB and YUV uploaded once; four histogram calls served without an upload


REGIONS = ((1, 0, 0, 0, 0), (2, 0, 0, 0, 0), (3, 0, 0, 0, 0), (1, 5, 3, 17, 9), (2, 1, 2, 7, 5), (4, 0, 1, 3, 2), (128, 0, 0, 0, 0))


def check_stage_region(ctx, oracle, w, h, linesize=None, regions=REGIONS):
    """target_scale + ROI crop (scope_stage_region) against the oracle, host planes in and out."""
    import numpy as np
    import scope_oracle

    linesize = linesize or 4 * w
    plane = oracle.fill_frame(w, h, 1, 4242, linesize)
    for scale, x, y, cx, cy in regions:
        sw, sh = w // scale, h // scale
        if sw == 0 or sh == 0 or (cx and (x + cx > sw or y + cy > sh)):
            with pytest.raises(Exception):
                ctx.stage_region(plane, w, scale, x, y, cx, cy)
            continue
        want = oracle.stage_region(plane, w, scale, x, y, cx, cy)
        assert np.array_equal(want, scope_oracle.np_stage_region(plane, w, scale, x, y, cx, cy))
        assert ctx.region_size(w, h, scale, x, y, cx, cy) == (want.shape[1] // 4, want.shape[0])
        got = ctx.stage_region(plane, w, scale, x, y, cx, cy)
        assert_same(got, want, f"stage_region {w}x{h} scale {scale} roi {(x, y, cx, cy)}")
        ow = want.shape[1] // 4
        padded = ctx.stage_region(plane, w, scale, x, y, cx, cy, out_linesize=ow * 4 + 12)
        assert_same(padded[:, :ow * 4], want, "stage_region padded pitch")
        assert (padded[:, ow * 4:] == 0xA5).all(), "stage_region wrote into the pitch padding"


def check_pipeline(ctx, oracle, w, h, n_frames=7, depth=3, linesize=None, region=(0, 0, 0, 0, 0)):
    """scope_pipe: frames in flight over three streams; every frame's three tables bit-exact, in order.
    With a region the frames are scaled / cropped on the GPU first (cm_render_target's pre-steps)."""
    import obs_color_monitor_amd as ocm

    linesize = linesize or 4 * w
    frames_ = [oracle.fill_frame(w, h, 1 + i % 5, 900 + i, linesize) for i in range(n_frames)]
    pipe = ocm.ScopePipe(ctx, w, h, depth=depth, colorspace=2, wvs_components=0x07, his_components=0x70, region=region)
    full_w = w
    try:
        assert pipe.collect() is None
        done = 0

        def check(i, got):
            rgb, w = frames_[i], full_w
            if region[0]:
                rgb = oracle.stage_region(rgb, full_w, *region)  # tight pitch
                w = rgb.shape[1] // 4
            assert (pipe.w, pipe.h) == (w, rgb.shape[0])
            yuv = oracle.rgb_to_yuv(rgb, w, 2)
            assert_same(got[0], oracle.vectorscope(yuv, w), f"pipe frame {i} vectorscope")
            assert_same(got[1], oracle.waveform(rgb, yuv, w, 0x07), f"pipe frame {i} waveform")
            assert_same(got[2], oracle.histogram_counts(rgb, yuv, w, 0x70), f"pipe frame {i} histogram")

        for i, f in enumerate(frames_):
            while not pipe.submit(f.ctypes.data, f.shape[1]):
                check(done, pipe.collect())
                done += 1
        while done < n_frames:
            check(done, pipe.collect())
            done += 1
        assert pipe.collect() is None
    finally:
        pipe.close()


def check_pipeline_tables_stay_valid(ctx, oracle, w, h, depth=2):
    """The tables scope_pipe_collect hands out stay untouched until the NEXT scope_pipe_collect, however many frames are
    submitted in between -- also when the ring was full and the very next submit re-uses the collected frame's ring
    slot (the pipe rotates depth + 1 pinned table buffers)."""
    import obs_color_monitor_amd as ocm

    frames_ = [oracle.fill_frame(w, h, 1 + i % 5, 40 + i) for i in range(2 * depth + 2)]

    def same_as_frame(held, i, what):
        f = frames_[i]
        yuv = oracle.rgb_to_yuv(f, w, 2)
        assert_same(held[0], oracle.vectorscope(yuv, w), f"{what}: vectorscope of frame {i}")
        assert_same(held[1], oracle.waveform(f, yuv, w, 0x07), f"{what}: waveform of frame {i}")
        assert_same(held[2], oracle.histogram_counts(f, yuv, w, 0x07), f"{what}: counts of frame {i}")

    def submit(i):
        return pipe.submit(frames_[i].ctypes.data, 4 * w)

    pipe = ocm.ScopePipe(ctx, w, h, depth=depth, colorspace=2, wvs_components=0x07, his_components=0x07)
    try:
        # a full ring: the oldest frame is collected and the next submit takes its slot at once
        for i in range(depth):
            assert submit(i)
        assert not submit(depth)
        held = pipe.collect()  # views into the pipe's pinned memory, NOT copied
        assert submit(depth) and not submit(depth + 1)
        ctx.sync()  # the new frame's tables have reached the host
        same_as_frame(held, 0, "full ring, one more frame submitted")
        while pipe.collect() is not None:
            pass
        # one frame in flight: after its collect the whole ring refills before the next collect
        assert submit(depth + 1)
        held = pipe.collect()
        for i in range(depth):
            assert submit(i)
        ctx.sync()
        same_as_frame(held, depth + 1, "ring refilled completely")
    finally:
        pipe.close()


LABEL_STRIP_SHA256 = "5147834c7b7b22bb771bf178679aa7ca44ec2ee5e476e09dbca203072ebcea5d"  # the build's own glyphs are part of the pinned picture


def check_graticule(ctx, quick=False):
    """scope_display_graticule against oracle/graticule_oracle.py's rasteriser on the reference's geometry (vectorscope
    labels + brackets + skin-tone line / I-Q cross, waveform and histogram scale lines in every display mode), over a
    random picture so that every blended byte counts; plus the edge cases of the pinned rule."""
    import hashlib

    import graticule_oracle as go
    import obs_color_monitor_amd as ocm

    strip = ctx.graticule_label_strip()
    assert strip.shape == (16, 96) and set(np.unique(strip)) == {0, 1}
    assert hashlib.sha256(strip.tobytes()).hexdigest() == LABEL_STRIP_SHA256, "the label glyphs changed: the pinned picture changed"
    rng = np.random.default_rng(5)

    def one(w, h, labels, lines, argb, what):
        pic = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        got = ctx.display_graticule(pic, labels, lines, argb)
        want = go.rasterise(pic, labels if labels is not None else [], lines if lines is not None else [], argb, strip)
        assert_same(got, want, what)
        return got, pic

    for cs in (1, 2):
        for flags, argb in ((1, 0x80FFBF00), (2 | 256, 0x8000FF00)):
            labels, lines = go.vectorscope(cs, flags, 0x0054FF)
            got, pic = one(256, 256, labels, lines, argb, f"vectorscope graticule cs{cs} flags{flags}")
            assert (got != pic).any()
            for zoom in ((1.7,) if quick else (1.02, 1.7, 3.3)):  # scaled about 127.5 like the sprite: parts leave the picture
                z, ofst = np.float32(zoom), np.float32(127.5) * (np.float32(1) - np.float32(zoom))
                zl = labels.copy()
                zl[:, 0:2] = zl[:, 0:2] * z + ofst
                zl[:, 2:4] = zl[:, 2:4] * z
                one(256, 256, zl, lines * z + ofst, argb, f"vectorscope graticule cs{cs} zoom {zoom}")
    for display, ncomp in ((0, 3), (1, 3), (2, 3), (1, 2)) if not quick else ((1, 3), (2, 2)):
        w = 70
        ow, oh = ctx.display_size(w, 256, {3: 0x07, 2: 0x50, 1: 0x20}[ncomp], display)
        for n in ((5,) if quick else (1, 4, 5, 10)):
            one(ow, oh, None, go.waveform(n, display, ncomp, w), 0x80FFBF00, f"waveform graticule {n} d{display} n{ncomp}")
        ow, oh = ctx.display_size(256, 120, {3: 0x07, 2: 0x50, 1: 0x20}[ncomp], display)
        one(ow, oh, None, go.histogram(5, 25.0, 0, 100, 0, display, ncomp, 120), 0x80FFBF00, f"histogram graticule d{display} n{ncomp}")
    # the rule's corners: nothing to draw, a point, a diagonal (x-major on a tie), steep and shallow, reversed end points,
    # segments that start and end outside the picture, two segments over one pixel (blended twice), opaque colour
    edge = np.array([[3, 3], [3, 3], [0, 0], [9, 9], [9, 0], [0, 9], [2.25, 1.5], [2.75, 18.5], [18.5, 7.75], [1.5, 7.25],
                     [-5, 4.5], [40, 4.5], [4.5, -3], [4.5, 30], [6, 6.5], [12, 6.5], [12, 6.5], [6, 6.5]], dtype=np.float32)
    one(20, 20, None, edge, 0x80FFBF00, "edge segments")
    one(20, 20, None, edge, 0xFF102030, "edge segments, opaque")
    one(20, 20, None, None, 0x80FFBF00, "nothing to draw")
    with pytest.raises(ocm.ScopeError):
        ctx.display_graticule(np.zeros((4, 4, 4), np.uint8), None, edge[:3], 0x80FFBF00)  # odd number of vertices


def check_display(ctx, oracle, w=70, h=40, quick=False):
    """D1-D3 post-passes against the oracle's restatement of the display shaders, every technique.
    quick: the subset the CPU emulation runs (every technique once); the GPU tests run everything."""
    rgb, yuv = frames(oracle, w, h, 2)
    vss = oracle.vectorscope(yuv, w)
    for color_type, cs, intensity in ((1, 2, 25), (1, 1, 3), (0, 2, 255), (1, 2, 0)):
        want = oracle.display_vectorscope(vss, max(intensity, 1), color_type, cs)
        assert_same(ctx.display_vectorscope(vss, intensity, color_type, cs), want, f"D1 ct{color_type} cs{cs} i{intensity}")
    for zoom in ((1.005, 1.5, 3.7) if quick else (1.0, 1.005, 1.02, 1.5, 2.0, 3.7, 40.0)):  # mouse-wheel zoom about the centre; <= 1.01 is the identity
        want = oracle.display_vectorscope(vss, 25, 1, 2, zoom)
        assert_same(ctx.display_vectorscope(vss, 25, 1, 2, zoom), want, f"D1 zoom {zoom}")
    assert_same(oracle.display_vectorscope(vss, 25, 1, 2, 1.005), oracle.display_vectorscope(vss, 25, 1, 2), "zoom <= 1.01")
    for comp in ((0x07, 0x50, 0x20) if quick else (0x07, 0x70, 0x50, 0x20, 0x04)):
        wv = oracle.waveform(rgb, yuv, w, comp)
        flt, hi = oracle.histogram(rgb, yuv, w, comp)
        flt_log, hi_log = oracle.histogram(rgb, yuv, w, comp, logscale=1)
        for display in (0, 1, 2):
            assert ctx.display_size(w, 256, comp, display) == oracle.display_size(w, 256, comp, display)
            for intensity in ((25,) if quick else (1, 25)):
                assert_same(ctx.display_waveform(wv, comp, display, intensity),
                            oracle.display_waveform(wv, comp, display, intensity), f"D2 c{comp:#x} d{display} i{intensity}")
            for lh in ((50,) if quick else (50, 200)):
                assert_same(ctx.display_histogram(flt, hi, comp, display, lh),
                            oracle.display_histogram(flt, hi, comp, display, lh), f"D3 c{comp:#x} d{display} h{lh}")
            assert_same(ctx.display_histogram(flt_log, hi_log, comp, display, 77),
                        oracle.display_histogram(flt_log, hi_log, comp, display, 77), f"D3 log c{comp:#x} d{display}")


def check_three_scopes_host(ctx, oracle, w, h, dist, wc=0x07, hc=0x07, linesize=None, colorspace=2):
    """scope_three_scopes_batch with host planes and tables (the two-pass form): the three tables of the separate scopes."""
    rgb, yuv = frames(oracle, w, h, dist, linesize, colorspace=colorspace)
    vss, wvs, his, st = ctx.three_scopes(rgb, w, wc, hc, colorspace)
    assert st == 0
    assert_same(vss, oracle.vectorscope(yuv, w), f"three scopes: vectorscope {w}x{h} d{dist}")
    assert_same(wvs, oracle.waveform(rgb, yuv, w, wc), f"three scopes: waveform {wc:#x} {w}x{h} d{dist}")
    assert_same(his, oracle.histogram_counts(rgb, yuv, w, hc), f"three scopes: histogram {hc:#x} {w}x{h} d{dist}")


def check_three_scopes_dev(ctx, oracle, to_dev, to_host, w, h, dists, wc=0x07, hc=0x07, linesize=None, colorspace=2, reps=2):
    """Device-resident planes and tables (to_dev / to_host as in check_device_path): small launches go through
    scope_dock_kernel (one read, the strip's texels in registers); every frame's three tables against the oracle, `reps`
    times (tickets and accumulators come back to zero)."""
    import numpy as np

    def pair_of(i, d):
        if isinstance(d, dict):  # screen-like content (screen_frame's keywords): flat areas, rectangles, sprinkled detail
            rgb = screen_frame(w, h, 500 + i, linesize=linesize, **d)
            return rgb, oracle.rgb_to_yuv(rgb, w, colorspace)
        return frames(oracle, w, h, d, linesize, seed=300 + d, colorspace=colorspace)

    pairs = [pair_of(i, d) for i, d in enumerate(dists)]
    n = len(pairs)
    ls = pairs[0][0].shape[1]
    hr, addr = to_dev(np.concatenate([p[0].reshape(-1) for p in pairs]))
    b = ctx._batch([addr + i * h * ls for i in range(n)], ls, w, h, True, colorspace)
    hv, av = to_dev(np.full(n * 65536, 0xEE, np.uint8))
    hw, aw = to_dev(np.full(n * 256 * w * 4, 0xEE, np.uint8))
    hh, ah = to_dev(np.full(n * 4096, 0xEE, np.uint8))
    for rep in range(reps):
        ctx.three_scopes_batch_dev(b, wc, hc, av, aw, ah)
        vss, wvs, his = to_host(hv).reshape(n, 256, 256), to_host(hw).reshape(n, 256, w, 4), to_host(hh).view(np.uint32).reshape(n, 256, 4)
        for f, (rgb, yuv) in enumerate(pairs):
            tag = f"{w}x{h} d{dists[f]} f{f} rep{rep}"
            assert_same(vss[f], oracle.vectorscope(yuv, w), "dock vectorscope " + tag)
            assert_same(wvs[f], oracle.waveform(rgb, yuv, w, wc), f"dock waveform {wc:#x} " + tag)
            assert_same(his[f], oracle.histogram_counts(rgb, yuv, w, hc), f"dock histogram {hc:#x} " + tag)


def check_vectorscope_fold(ctx, oracle, to_dev, to_host, w, h, dists, launches=3, linesize=None):
    """The in-launch fold of the vectorscope's partial tables (vss_write_out, round 6): launches of 2, 3 and 4 slices per frame --
    staged planes, fused K0, the packed wire format, the three-scope kernel -- against the oracle AND against the same launch with
    vss.fold = 0 (partials + the reduction kernel), several launches in a row into fresh tables (the arrival tickets come back to
    zero; a stale partial of the launch before would show), one slice count after the other on ONE context (the partial buffers are
    reused with another layout)."""
    import numpy as np

    pairs = [frames(oracle, w, h, d, linesize, seed=600 + i) if not isinstance(d, dict) else None for i, d in enumerate(dists)]
    for i, d in enumerate(dists):
        if isinstance(d, dict):
            rgb = screen_frame(w, h, 700 + i, linesize=linesize, **d)
            pairs[i] = (rgb, oracle.rgb_to_yuv(rgb, w, 2))
    n = len(pairs)
    ls = pairs[0][0].shape[1]
    want = [oracle.vectorscope(yuv, w) for _, yuv in pairs]
    want_wvs = [oracle.waveform(r, y, w, 0x07) for r, y in pairs] if w % 4 == 0 else None
    keep_rgb, a_rgb = to_dev(np.concatenate([p[0].reshape(-1) for p in pairs]))  # (the handles own the memory)
    keep_yuv, a_yuv = to_dev(np.concatenate([p[1].reshape(-1) for p in pairs]))
    b_yuv = ctx._batch([a_yuv + i * h * ls for i in range(n)], ls, w, h, False, 2)
    b_rgb = ctx._batch([a_rgb + i * h * ls for i in range(n)], ls, w, h, True, 2)
    for slices in (2, 4, 3):
        for fold in (1, 0):
            with options(ctx, **{"slices": slices, "vss.fold": fold}):
                before = ctx.get_stat("vss.fold_launches"), ctx.get_stat("trio.fold_launches")
                for rep in range(launches):
                    tag = f"{w}x{h} slices {slices} fold {fold} launch {rep}"
                    for name, b in (("staged", b_yuv), ("fused", b_rgb)):
                        ho, ao = to_dev(np.full(n * 65536, 0xEE, np.uint8))
                        ctx.vectorscope_batch_dev(b, ao)
                        got = to_host(ho).reshape(n, 256, 256)
                        for f in range(n):
                            assert_same(got[f], want[f], f"vectorscope fold {name} frame {f} {tag}")
                    hp, ap = to_dev(np.full(n * 32768 * 4, 0xEE, np.uint8))
                    ctx.vectorscope_batch_dev(b_yuv, ap, packed16=True)
                    wire = to_host(hp).view(np.uint32).reshape(n, 32768)
                    for f in range(n):
                        flat = want[f].reshape(-1).astype(np.uint32)
                        assert_same(wire[f], flat[0::2] | (flat[1::2] << 16), f"vectorscope fold packed frame {f} {tag}")
                    if w % 4 == 0:
                        hv, av = to_dev(np.full(n * 65536, 0xEE, np.uint8))
                        hw, aw = to_dev(np.full(n * 256 * w * 4, 0xEE, np.uint8))
                        hh, ah = to_dev(np.full(n * 4096, 0xEE, np.uint8))
                        ctx.three_scopes_batch_dev(b_rgb, 0x07, 0x07, av, aw, ah)
                        got = to_host(hv).reshape(n, 256, 256)
                        wv = to_host(hw).reshape(n, 256, w, 4)
                        for f in range(n):
                            assert_same(got[f], want[f], f"three scopes: vectorscope fold frame {f} {tag}")
                            assert_same(wv[f], want_wvs[f], f"three scopes: waveform beside the fold frame {f} {tag}")
                # the form the test means to cover is the form that ran (no silent fall-back to the reduction kernel, or the other way)
                assert ctx.get_stat("vss.fold_launches") - before[0] == (3 * launches if fold else 0), (slices, fold)
                # (the three-scope launch picks its own slices per frame -- 2 .. 4, the fold's range, only where frames are many or CUs few)
                assert ctx.get_stat("trio.fold_launches") - before[1] in ((0, launches) if fold and w % 4 == 0 else (0,)), (slices, fold)
    del keep_rgb, keep_yuv


def check_vectorscope_side_reduce(ctx, oracle, to_dev, to_host, w, h, dists, calls=12, fuse=False):
    """SCOPE_SIDE_REDUCE: a train of one-frame calls whose reductions run on the context's second stream (two sets of
    partial tables used in turn), every call into a table of its own; after scope_ctx_fence + sync every table equals the
    oracle's.  Then a call WITHOUT the flag right behind flagged ones: it fences by itself."""
    import numpy as np

    pairs = [frames(oracle, w, h, d, seed=900 + d) for d in dists]
    ls = pairs[0][0].shape[1]
    planes = []
    for rgb, yuv in pairs:
        hnd, addr = to_dev((rgb if fuse else yuv).reshape(-1))
        planes.append((hnd, addr))
    batches = [ctx._batch([addr], ls, w, h, fuse, 2) for _, addr in planes]
    outs = [to_dev(np.full(65536, 0xEE, np.uint8)) for _ in range(calls)]
    for i in range(calls):
        ctx.vectorscope_batch_dev(batches[i % len(batches)], outs[i][1], sync=False, side_reduce=True)
    ctx.fence()
    ctx.sync()
    for i in range(calls):
        assert_same(to_host(outs[i][0]).reshape(256, 256), oracle.vectorscope(pairs[i % len(pairs)][1], w), f"side reduce call {i}")
    # flagged, flagged, then plain: the plain call fences, and its own table is right as well
    extra = [to_dev(np.full(65536, 0xEE, np.uint8)) for _ in range(3)]
    ctx.vectorscope_batch_dev(batches[0], extra[0][1], sync=False, side_reduce=True)
    ctx.vectorscope_batch_dev(batches[-1], extra[1][1], sync=False, side_reduce=True)
    ctx.vectorscope_batch_dev(batches[0], extra[2][1], sync=True)
    for k, which in enumerate((0, -1, 0)):
        assert_same(to_host(extra[k][0]).reshape(256, 256), oracle.vectorscope(pairs[which][1], w), f"side reduce then plain, table {k}")
    # every OTHER entry point fences as well (include/scope_hip.h: "the next call on the context without this flag"): a flagged
    # call into a packed table with the table's unpack enqueued right behind it, and -- for RGB planes -- the one-read
    # three-scope call of the same frame into the same vectorscope table behind a flagged call
    for rep in range(4):
        hp, ap = to_dev(np.full(32768 * 4, 0xEE, np.uint8))
        ho, ao = to_dev(np.full(65536, 0xEE, np.uint8))
        ctx.vectorscope_batch_dev(batches[rep % len(batches)], ap, sync=False, packed16=True, side_reduce=True)
        ctx.unpack16_clamp_dev(ap, ao, 32768)
        ctx.sync()
        assert_same(to_host(ho).reshape(256, 256), oracle.vectorscope(pairs[rep % len(pairs)][1], w), f"side reduce then unpack, rep {rep}")
    if fuse:
        hv, av = to_dev(np.full(65536, 0xEE, np.uint8))
        hw, aw = to_dev(np.full(256 * w * 4, 0xEE, np.uint8))
        hh, ah = to_dev(np.full(4096, 0xEE, np.uint8))
        ctx.vectorscope_batch_dev(batches[0], av, sync=False, side_reduce=True)
        ctx.three_scopes_batch_dev(batches[-1], 0x07, 0x07, av, aw, ah)
        assert_same(to_host(hv).reshape(256, 256), oracle.vectorscope(pairs[-1][1], w), "side reduce then three scopes: vectorscope")
        assert_same(to_host(hw).reshape(256, w, 4), oracle.waveform(pairs[-1][0], pairs[-1][1], w, 0x07), "side reduce then three scopes: waveform")


def check_pipeline_without_histogram(ctx, oracle, w, h, depth=2):
    """scope_pipe_* with vectorscope + waveform only: the one-read kernel in its form WITHOUT the histogram hand-off
    (scope_trio_kernel<CH, false>); the histogram pointer comes back NULL."""
    import obs_color_monitor_amd as ocm

    frames_ = [oracle.fill_frame(w, h, d, 40 + d) for d in (1, 5, 2)]
    pipe = ocm.ScopePipe(ctx, w, h, depth=depth, colorspace=2, wvs_components=0x05, his_components=0)
    try:
        for f in frames_:
            while not pipe.submit(f.ctypes.data, 4 * w):
                pipe.collect()
            vss, wvs, his = pipe.collect()
            yuv = oracle.rgb_to_yuv(f, w, 2)
            assert his is None
            assert_same(vss, oracle.vectorscope(yuv, w), "pipe without histogram: vectorscope")
            assert_same(wvs, oracle.waveform(f, yuv, w, 0x05), "pipe without histogram: waveform 0x05")
    finally:
        pipe.close()
