"""Parity of the product sources against the oracle WITHOUT a GPU: csrc/*.hip compiled for the CPU
by tests/emul (fiber-per-GPU-thread executor).  Small frames only; the -m gpu tests repeat these
on the real kernels and add the full BASELINE sizes."""
import numpy as np
import pytest

import parity_cases as pc

SMALL = [(64, 36, None), (70, 9, None), (33, 5, 140), (64, 8, 320), (1, 1, None), (130, 3, 524)]


@pytest.mark.parametrize("dist", [1, 2, 3, 5])
def test_vectorscope_distributions(emul_ctx, oracle, dist):
    pc.check_vectorscope(emul_ctx, oracle, 64, 36, dist)


@pytest.mark.parametrize("w,h,ls", SMALL)
def test_vectorscope_geometry(emul_ctx, oracle, w, h, ls):
    pc.check_vectorscope(emul_ctx, oracle, w, h, 1, ls)


@pytest.mark.parametrize("merge", [0, 1, 2, 3])
@pytest.mark.parametrize("hash_", [0, 1])
def test_vectorscope_merge_and_hash_variants(emul_ctx, oracle, merge, hash_):
    with pc.options(emul_ctx, **{"vss.merge": merge, "vss.hash": hash_}):
        pc.check_vectorscope(emul_ctx, oracle, 70, 9, 3)
        pc.check_vectorscope(emul_ctx, oracle, 70, 9, 1)


@pytest.mark.parametrize("groups", [0, 1, 2, 3])
@pytest.mark.parametrize("unroll", [1, 2, 4])
def test_vectorscope_lane_groups(emul_ctx, oracle, groups, unroll):
    """Lane groups walk separate streams of the slice; every split must cover each pixel once."""
    with pc.options(emul_ctx, **{"vss.groups": groups, "vss.unroll": unroll, "nt": groups & 1}):
        pc.check_vectorscope(emul_ctx, oracle, 70, 9, 2)
        pc.check_vectorscope(emul_ctx, oracle, 33, 5, 1, 140)
        pc.check_vectorscope(emul_ctx, oracle, 3, 2, 1)


def test_vectorscope_saturation_and_epochs(emul_ctx, oracle):
    """One workgroup (slices=1) sees 262144 flat pixels: counters must survive four 65280-pixel
    epochs with clamp passes in between and still report exactly 255."""
    for merge, hash_, groups in ((0, 1, 2), (1, 0, 0), (2, 1, 3)):
        with pc.options(emul_ctx, **{"slices": 1, "vss.merge": merge, "vss.hash": hash_, "vss.groups": groups}):
            pc.check_vectorscope(emul_ctx, oracle, 512, 512, 3)
    with pc.options(emul_ctx, slices=1):
        pc.check_vectorscope(emul_ctx, oracle, 511, 300, 2, 2048)  # tails + epochs + pitch


@pytest.mark.parametrize("cs", [1, 2])
@pytest.mark.parametrize("hash_", [0, 1, 2])
def test_vectorscope_fused_k0(emul_ctx, oracle, cs, hash_):
    """vss.fused_hash 2 (the default): rows of 136 words, two pixels' addresses per register (vss_add_pair)."""
    with pc.options(emul_ctx, **{"vss.fused_hash": hash_}):
        pc.check_vectorscope(emul_ctx, oracle, 70, 9, 1, None, fuse=True, colorspace=cs)
        pc.check_vectorscope(emul_ctx, oracle, 64, 8, 4, None, fuse=True, colorspace=cs)
        pc.check_vectorscope(emul_ctx, oracle, 33, 5, 2, 140, fuse=True, colorspace=cs)


@pytest.mark.parametrize("merge", [0, 2, 3])
@pytest.mark.parametrize("hash_", [0, 2])
def test_vectorscope_fused_k0_long_streams(emul_ctx, oracle, merge, hash_):
    """One workgroup walks 131072 pixels: scope_stream's unchecked main loop and its checked tail, two clamp passes, the
    absorb mode on flat and banded content (dist 3, 4) and the plain adds on noise, in both table layouts."""
    for groups, unroll in ((3, 2), (0, 4), (2, 1)):
        with pc.options(emul_ctx, **{"slices": 1, "vss.merge": merge, "vss.fused_hash": hash_, "vss.groups": groups,
                                     "vss.fused_unroll": unroll}):
            for dist in (1, 3, 4):
                pc.check_vectorscope(emul_ctx, oracle, 512, 256, dist, None, fuse=True, colorspace=2)
    with pc.options(emul_ctx, **{"slices": 1, "vss.merge": merge, "vss.fused_hash": hash_}):
        pc.check_vectorscope(emul_ctx, oracle, 509, 130, 2, 2048, fuse=True, colorspace=1)  # tails + pitch


@pytest.mark.parametrize("groups,w,h,slices", [(0, 256, 80, 1), (0, 256, 80, 4), (3, 64, 300, 2), (2, 128, 200, 3), (1, 384, 70, 2),
                                               (0, 256, 530, 1), (3, 64, 2100, 1),  # these two: eight batches of the unchecked main loop
                                               (3, 32, 259, 2)])  # ONE lane group per row (64 >> 3 cells), shares of 129 and 130 rows
def test_padded_planes_walk_whole_rows(emul_ctx, oracle, groups, w, h, slices):
    """scope_stream's ROWS form: a padded DEVICE plane (linesize > 4 W, src/vectorscope.c:225,236; host planes are packed on
    their way to the GPU) whose rows are whole lane groups is walked in whole rows with a wave-uniform column cursor -- the plain
    and the fused vectorscope (unchecked main loop and checked tail, lane groups whose shares differ by a row, several slices)
    and the histogram's kernels.  The widths here are multiples of 64 >> groups cells and every lane group has sixteen rows or
    more, so this form runs (seen with a probe build); the geometry tests (odd widths, few rows) keep the general form covered."""
    ls = 4 * w + 64
    rows0 = emul_ctx.get_stat("vss.rows_launches"), emul_ctx.get_stat("his.rows_launches")
    for unroll in (0, 4):
        with pc.options(emul_ctx, **{"slices": slices, "vss.groups": groups, "vss.unroll": unroll, "vss.fused_unroll": unroll}):
            pc.check_device_path(emul_ctx, oracle, _np_dev, lambda hnd: hnd, w, h, ls, n=2)
    # the ROWS form is the form that ran (the kernels' own predicate, asked on the host: scope_ctx_get_stat) -- a change of the
    # predicate, of the sixteen-rows threshold or -DSCOPE_ROWS=0 would otherwise leave these tests green on the general form
    assert emul_ctx.get_stat("vss.rows_launches") - rows0[0] >= 4, "the padded vectorscope launches did not take the ROWS form"
    if (w // 4) % 64 == 0 and h >= 16 * slices:  # the histogram's kernels walk a slice as ONE stream: a row must be whole waves
        assert emul_ctx.get_stat("his.rows_launches") - rows0[1] >= 2, "the padded histogram launches did not take the ROWS form"
    rows1 = emul_ctx.get_stat("vss.rows_launches")
    with pc.options(emul_ctx, **{"slices": slices, "vss.groups": groups, "vss.merge": 0}):
        pc.check_device_path(emul_ctx, oracle, _np_dev, lambda hnd: hnd, w, h, ls, n=1, offset=16)
    assert emul_ctx.get_stat("vss.rows_launches") > rows1  # (planes that start 16 bytes into their buffer are as aligned as before)


def test_packed_planes_never_take_the_rows_form(emul_ctx, oracle):
    before = emul_ctx.get_stat("vss.rows_launches"), emul_ctx.get_stat("his.rows_launches")
    pc.check_device_path(emul_ctx, oracle, _np_dev, lambda hnd: hnd, 256, 80, 4 * 256, n=2)   # no padding
    pc.check_device_path(emul_ctx, oracle, _np_dev, lambda hnd: hnd, 100, 80, 4 * 100 + 64, n=1)  # rows that are not whole lane groups
    with pc.options(emul_ctx, slices=8):
        pc.check_device_path(emul_ctx, oracle, _np_dev, lambda hnd: hnd, 256, 80, 4 * 256 + 64, n=1)  # ten rows per slice: below sixteen
    assert (emul_ctx.get_stat("vss.rows_launches"), emul_ctx.get_stat("his.rows_launches")) == before


def test_three_scope_chroma_guard(emul_ctx):
    """scope_trio_kernel's launch-time guard on the compact vectorscope table (trio_chroma_fits): both matrices the library knows
    fit, their FULL-RANGE forms -- whose chroma reaches 0 and 255 -- are refused (the two-pass path would run: `dock` 0 is that
    path, covered by test_three_scopes_one_read[0])."""
    assert emul_ctx.get_stat("trio.chroma_guard_selftest") == 15
    with pytest.raises(Exception):
        emul_ctx.get_stat("no.such.counter")


@pytest.mark.parametrize("comp", pc.COMPONENTS)
def test_waveform_components(emul_ctx, oracle, comp):
    pc.check_waveform(emul_ctx, oracle, 70, 9, 5, comp)


@pytest.mark.parametrize("w,h,ls", SMALL)
def test_waveform_geometry(emul_ctx, oracle, w, h, ls):
    pc.check_waveform(emul_ctx, oracle, w, h, 1, 0x07, ls)


def test_waveform_saturation_and_bands(emul_ctx, oracle):
    pc.check_waveform(emul_ctx, oracle, 64, 600, 3, 0x07)  # every column saturates at 255
    with pc.options(emul_ctx, slices=5, **{"wvs.strip": 64}):  # five row bands -> partial tables + saturating reduce
        pc.check_waveform(emul_ctx, oracle, 64, 600, 2, 0x70)
        pc.check_waveform(emul_ctx, oracle, 66, 40, 1, 0x07)  # width % 4 != 0: single band fallback
    with pc.options(emul_ctx, **{"wvs.wide": 0}):
        pc.check_waveform(emul_ctx, oracle, 70, 100, 5, 0x07)
    pc.check_waveform(emul_ctx, oracle, 70, 101, 5, 0x07)  # odd row count: the single-row tail
    pc.check_waveform(emul_ctx, oracle, 71, 33, 1, 0x50)  # odd width: 4-byte loads
    pc.check_waveform(emul_ctx, oracle, 70, 34, 1, 0x20, 70 * 4 + 4)  # pitch % 8 == 4: 4-byte loads
    pc.check_waveform(emul_ctx, oracle, 1, 1, 1, 0x07)


def test_waveform_taller_than_16_bit_counters(emul_ctx, oracle):
    """More than 65535 rows: the 16-bit per-band counters force at least two row bands (flat frame: every row hits
    the same counter of its column)."""
    pc.check_waveform(emul_ctx, oracle, 2, 70000, 3, 0x07)
    with pc.options(emul_ctx, slices=1):
        pc.check_waveform(emul_ctx, oracle, 1, 65546, 3, 0x04)  # one band would wrap to 10


@pytest.mark.parametrize("persist", [0, 1])
def test_waveform_unit_walk(emul_ctx, oracle, persist):
    """More units than resident workgroups: a workgroup walks several (frame, strip) units with one load stream."""
    with pc.options(emul_ctx, **{"wvs.persist": persist}):
        pc.check_waveform(emul_ctx, oracle, 130, 21, 9, 0x20)  # 27 units, odd rows, ragged last strip
        pc.check_waveform(emul_ctx, oracle, 64, 150, 14, 0x07)  # more batches than one slot pair per wave
        with pc.options(emul_ctx, slices=3):
            pc.check_waveform(emul_ctx, oracle, 66, 50, 7, 0x05)  # bands of different heights


@pytest.mark.parametrize("w,h,ls", [(64, 36, None), (72, 9, None), (20, 7, 96), (4, 1, None), (132, 8, None), (320, 17, 1296), (1284, 19, None)])
def test_waveform_narrow_strips_geometry(emul_ctx, oracle, w, h, ls):
    """16-column strips (the shape of one- and two-frame launches): eight rows per wave instruction into four table
    replicas, ragged last strips, 1..7 rows left over, more than 16 units (the XCD pairing of neighbouring strips)."""
    with pc.options(emul_ctx, **{"wvs.strip": 16}):
        pc.check_waveform(emul_ctx, oracle, w, h, 1, 0x07, ls)


@pytest.mark.parametrize("comp", pc.COMPONENTS)
def test_waveform_narrow_strips_components(emul_ctx, oracle, comp):
    with pc.options(emul_ctx, **{"wvs.strip": 16}):
        pc.check_waveform(emul_ctx, oracle, 72, 19, 5, comp)


def test_waveform_narrow_strips_saturation_walk_fused(emul_ctx, oracle):
    with pc.options(emul_ctx, **{"wvs.strip": 16}):
        pc.check_waveform(emul_ctx, oracle, 64, 600, 3, 0x07)  # every column saturates; replicas add up beyond 255
        pc.check_waveform(emul_ctx, oracle, 16, 150, 14, 0x05)  # more batches than one slot pair per wave
        pc.check_waveform(emul_ctx, oracle, 72, 9, 1, 0x20, None, fuse=True, colorspace=1)
        pc.check_waveform(emul_ctx, oracle, 72, 9, 1, 0x50, None, fuse=True, colorspace=2)
        with pc.options(emul_ctx, **{"wvs.persist": 0}):
            pc.check_waveform(emul_ctx, oracle, 132, 21, 9, 0x20)
    with pc.options(emul_ctx, slices=4):  # what would be four row bands becomes narrow strips, on its own
        pc.check_waveform(emul_ctx, oracle, 64, 100, 2, 0x70)
        pc.check_waveform_histogram(emul_ctx, oracle, 72, 36, 5)


@pytest.mark.parametrize("w,h,dist", [(72, 36, 5), (132, 21, 2)])
def test_waveform_histogram_one_pass_narrow_strips(emul_ctx, oracle, w, h, dist):
    with pc.options(emul_ctx, **{"wvs.strip": 16}):
        pc.check_waveform_histogram(emul_ctx, oracle, w, h, dist)


def test_waveform_deep_queue_across_flushes(emul_ctx, oracle):
    """Batches whose workgroups walk several units keep extra loads in flight across each flush (register sets of
    their own, binned first thing after the flush): more units than workgroups, at least six slots per wave."""
    pc.check_waveform(emul_ctx, oracle, 512, 265, 5, 0x07)  # 8 units on 6 workgroups, an odd last row
    pc.check_waveform(emul_ctx, oracle, 448, 300, 1, 0x20, 448 * 4 + 64)  # 7 units: one workgroup walks two
    pc.check_waveform_histogram(emul_ctx, oracle, 512, 264, 2)
    for cs in (1, 2):  # the fused one-row (luma) form takes the deep queue as well
        pc.check_waveform(emul_ctx, oracle, 512, 265, 1, 0x20, None, fuse=True, colorspace=cs)
    with pc.options(emul_ctx, **{"wvs.deep": 0}):
        pc.check_waveform(emul_ctx, oracle, 512, 265, 5, 0x07)
        pc.check_waveform(emul_ctx, oracle, 512, 265, 1, 0x20, None, fuse=True, colorspace=2)


@pytest.mark.parametrize("forced", [0, 1, 2])
def test_fused_vectorscope_screen_content(emul_ctx, oracle, forced):
    """Screen-like content (flat areas, rectangles, sprinkled detail, greys, random alpha) and the oracle's distributions through
    the fused vectorscope with the absorb mode voted / forced; several unroll factors, one workgroup per frame (clamp passes between
    runs) and the default slicing, a padded pitch and a ragged width."""
    for slices, unroll in ((0, 0), (1, 4), (2, 2), (1, 1)):
        with pc.options(emul_ctx, **{"slices": slices, "vss.fused_unroll": unroll}):
            pc.check_fused_screen_content(emul_ctx, oracle, [(256, 96, None), (132, 50, 132 * 4 + 16)] if slices == 0 else [(160, 64, None)], forced)
    with pc.options(emul_ctx, slices=1):
        rgb = pc.screen_frame(512, 300, 5, detail=0.0, rects=0)   # one colour, 153 600 texels in one workgroup: counts past 65 535
        pc.check_vectorscope_rgb_frame(emul_ctx, oracle, rgb, 512, 2, "one flat colour")


@pytest.mark.parametrize("dock", [1, 2, 0])
def test_three_scopes_one_read(emul_ctx, oracle, dock):
    """One read of the plane for waveform + histogram + vectorscope.  dock=1: scope_trio_kernel (both tables resident in LDS,
    any number of frames, slices of strips); dock=2: round 2's scope_dock_kernel (the emulated chip has 6 CUs, so up to six
    16-column strips qualify, the rest takes two passes); dock=0: always the two-pass form of the same entry point."""
    with pc.options(emul_ctx, dock=dock):
        dev = lambda *a, **k: pc.check_three_scopes_dev(emul_ctx, oracle, _np_dev, lambda hnd: hnd, *a, **k)  # noqa: E731
        dev(72, 19, [5])                                  # 5 strips, the last one 8 columns wide, row ranges of 3 rows
        dev(64, 150, [3], reps=1)                         # flat: waveform columns saturate, absorb mode in the vectorscope
        dev(36, 7, [1], 0x05, 0x04, 36 * 4 + 8, 1)        # fewer rows than row ranges, padded pitch, BT.601
        dev(16, 9, [1, 2], 0x02, 0x02, reps=1)            # two frames in one launch
        dev(100, 5, [1], reps=1)                          # 7 strips on 6 CUs: two passes
        dev(80, 2170, [2], reps=1)                        # 17 loads per lane, the last ones partly past the end
        dev(16, 2300, [1], reps=1)                        # more rows than the dock kernel's registers hold
        if dock == 1:
            dev(32, 2100, [3, 1, 2, 5, 1, 3], reps=1)     # six frames on six CUs: one slice of TWO strips per frame, final tables
                                                          # straight from the kernel, a clamp pass between the strips (flat: saturates)
            dev(200, 40, [1, 5], 0x06, 0x02, reps=2)      # 13 strips (the last 8 columns wide) in 4 paired slices per frame
            dev(16, 4090, [3], reps=1)                    # a tall flat strip: clamp passes inside the strip


@pytest.mark.parametrize("forced", [0, 1])
@pytest.mark.parametrize("strip", [0, 16, 32])
def test_three_scopes_screen_content(emul_ctx, oracle, strip, forced):
    """scope_trio_kernel's absorb mode (the emulated wave's vote is its one lane's: a lane with a flat texel pair absorbs): screen-like
    content, where a lane keeps its colour for long, meets single texels of detail and crosses into other flat areas -- colour changes
    in the vote's slots, plain adds between them.  forced 1: the vectorscope half's run-length form (trio_rle_pair, round 6) on every
    slot of any content (vss.rle = 1); by vote it runs where a wave is flat in most lanes but not in all -- never in a one-lane wave,
    under the 64-lane model (tests/test_wave64_emul.py) on the screen-like frames here."""
    dev = lambda *a, **k: pc.check_three_scopes_dev(emul_ctx, oracle, _np_dev, lambda hnd: hnd, *a, **k)  # noqa: E731
    kinds = [{}, {"detail": 0.0}, {"detail": 0.3}, {"grey": True}, {"alpha_noise": True, "detail": 0.01}, {"rects": 40, "detail": 0.05}]
    with pc.options(emul_ctx, **{"trio.strip": strip, "vss.rle": forced}):
        dev(96, 260, kinds, reps=1)
        dev(72, 130, kinds[:2], 0x05, 0x04, 72 * 4 + 8, 1, reps=1)
        dev(32, 2100, [kinds[5], 3], reps=1)              # clamp passes with lanes holding counts
        dev(200, 40, kinds[3:6], 0x02, 0x02, reps=2)
        if forced:
            dev(100, 37, [1, 2, 5], 0x07, 0x03, 100 * 4 + 24, 1, reps=1)  # noise, a tight cluster, alpha holes: every pair rough or a short run; a ragged last strip
            dev(36, 3, [3, 4], reps=1)                    # three rows: most lanes of a wave hold no texel and break every run


@pytest.mark.parametrize("strip", [16, 32])
def test_three_scopes_strip_widths(emul_ctx, oracle, strip):
    """scope_trio_kernel's two shapes: 16-column strips with waveform replicas (two; four for a one-channel mask) and 32-column
    strips with one -- ragged last strips, fewer rows than a wave's row ranges, clamp passes, every mask width."""
    with pc.options(emul_ctx, **{"trio.strip": strip}):
        dev = lambda *a, **k: pc.check_three_scopes_dev(emul_ctx, oracle, _np_dev, lambda hnd: hnd, *a, **k)  # noqa: E731
        dev(72, 19, [5])                                  # the last strip 8 columns wide
        dev(100, 37, [1, 2], 0x07, 0x03, 100 * 4 + 24, 1)  # ragged (100 = 3 * 32 + 4), padded pitch, BT.601, two frames
        dev(64, 150, [3], 0x04, 0x04, reps=1)             # one channel, flat: four replicas of one counter
        dev(36, 3, [1], 0x06, 0x02, reps=2)               # three rows: most row ranges of a wave are empty
        dev(32, 2100, [3, 1, 5], reps=1)                  # clamp passes between and inside strips
        dev(200, 40, [1, 5, 2, 4], 0x05, 0x05, reps=1)


def test_three_scopes_host_planes(emul_ctx, oracle):
    pc.check_three_scopes_host(emul_ctx, oracle, 72, 19, 5)
    pc.check_three_scopes_host(emul_ctx, oracle, 70, 9, 1, 0x05, 0x04, colorspace=1)
    pc.check_three_scopes_host(emul_ctx, oracle, 64, 36, 2, 0x07, 0x70)  # histogram of another plane kind: its own pass


def test_waveform_fused_k0(emul_ctx, oracle):
    pc.check_waveform(emul_ctx, oracle, 70, 9, 1, 0x20, None, fuse=True, colorspace=1)
    pc.check_waveform(emul_ctx, oracle, 70, 9, 1, 0x50, None, fuse=True, colorspace=2)


@pytest.mark.parametrize("comp", pc.COMPONENTS)
def test_histogram_components(emul_ctx, oracle, comp):
    pc.check_histogram_counts(emul_ctx, oracle, 70, 9, 5, comp)


@pytest.mark.parametrize("w,h,ls", SMALL)
def test_histogram_geometry(emul_ctx, oracle, w, h, ls):
    pc.check_histogram_counts(emul_ctx, oracle, w, h, 2, 0x07, ls)


def test_histogram_fused_k0(emul_ctx, oracle):
    pc.check_histogram_counts(emul_ctx, oracle, 70, 9, 1, 0x70, None, fuse=True, colorspace=2)


@pytest.mark.parametrize("logscale,fixed,ratio", [(0, 0, 0), (1, 0, 0), (0, 100, 0), (1, 0, 250), (0, 0, 1), (1, 5, 9)])
def test_histogram_level_pass(emul_ctx, oracle, logscale, fixed, ratio):
    pc.check_histogram_full(emul_ctx, oracle, 64, 36, 2, 0x07, logscale, fixed, ratio)
    pc.check_histogram_full(emul_ctx, oracle, 64, 36, 1, 0x20, logscale, fixed, ratio)


@pytest.mark.parametrize("cs", [1, 2])
def test_k0(emul_ctx, oracle, cs):
    pc.check_k0(emul_ctx, oracle, 70, 9, 1, cs)
    pc.check_k0(emul_ctx, oracle, 33, 5, 4, cs, 140)


def test_batches(emul_ctx, oracle):
    pc.check_batches(emul_ctx, oracle, 64, 12, 3)   # 3 frames x 2 slices on the 6 emulated CUs
    pc.check_batches(emul_ctx, oracle, 66, 5, 7)    # more frames than CUs


def test_error_behaviour(emul_ctx, oracle):
    pc.check_error_behaviour(emul_ctx, oracle)


def _np_dev(a):
    """In the emulation "device memory" is host memory, so a numpy buffer is its own handle."""
    import numpy as np

    buf = np.ascontiguousarray(a).copy()
    return buf, buf.ctypes.data


@pytest.mark.parametrize("w,h,ls,off", [(64, 12, 256, 0), (35, 7, 140, 0), (64, 12, 256, 4), (35, 7, 148, 8)])
def test_device_path_and_unaligned_rows(emul_ctx, oracle, w, h, ls, off):
    pc.check_device_path(emul_ctx, oracle, _np_dev, lambda hnd: hnd, w, h, ls, n=2, offset=off)


def test_surface_fanout_uploads_each_plane_once(emul_ctx, oracle):
    pc.check_surface_fanout(emul_ctx, oracle, 70, 20, 4 * 70 + 24)
    pc.check_surface_fanout(emul_ctx, oracle, 64, 9)


def test_context_is_safe_to_share_between_threads(emul_ctx, oracle):
    """The plugin runs one pipeline thread per stand-alone source plus the render thread on ONE module context."""
    import threading

    w, h = 66, 30
    rgb, yuv = pc.frames(oracle, w, h, 1)
    want_v, want_w = oracle.vectorscope(yuv, w), oracle.waveform(rgb, yuv, w, 0x07)
    want_h = oracle.histogram_counts(rgb, yuv, w, 0x07)
    errors = []

    def worker(kind):
        try:
            for i in range(6):
                if kind == 0:
                    got, _ = emul_ctx.vectorscope(yuv=yuv, width=w)
                    assert np.array_equal(got, want_v)
                elif kind == 1:
                    got, _ = emul_ctx.waveform(rgb=rgb, yuv=yuv, width=w, components=0x07)
                    assert np.array_equal(got, want_w)
                elif kind == 2:
                    got, _ = emul_ctx.histogram_counts(rgb=rgb, yuv=yuv, width=w, components=0x07)
                    assert np.array_equal(got, want_h)
                else:  # a dock fan-out holding the context across several calls
                    emul_ctx.surface_begin()
                    a, _ = emul_ctx.vectorscope(rgb=rgb, yuv=yuv, width=w)
                    b, _ = emul_ctx.waveform(rgb=rgb, yuv=yuv, width=w, components=0x07)
                    assert emul_ctx.surface_end() == (2, 0)
                    assert np.array_equal(a, want_v) and np.array_equal(b, want_w)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(k,)) for k in (0, 1, 2, 3, 3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_frame_pipeline(emul_ctx, oracle):
    pc.check_pipeline(emul_ctx, oracle, 70, 20, n_frames=5, depth=2, linesize=4 * 70 + 24)


@pytest.mark.parametrize("depth", [1, 3])
def test_frame_pipeline_tables_stay_valid(emul_ctx, oracle, depth):
    pc.check_pipeline_tables_stay_valid(emul_ctx, oracle, 70, 20, depth)


@pytest.mark.parametrize("region", [(2, 0, 0, 0, 0), (1, 3, 2, 33, 11), (3, 1, 1, 20, 4)])
def test_frame_pipeline_scaled_and_cropped(emul_ctx, oracle, region):
    """target_scale and the ROI rectangle applied on the device before the scopes."""
    pc.check_pipeline(emul_ctx, oracle, 70, 20, n_frames=4, depth=2, linesize=4 * 70 + 24, region=region)


@pytest.mark.parametrize("w,h,ls", [(70, 36, None), (64, 33, 64 * 4 + 20), (129, 5, None), (1, 1, None)])
def test_stage_region(emul_ctx, oracle, w, h, ls):
    pc.check_stage_region(emul_ctx, oracle, w, h, ls)


def test_stage_region_rejects_bad_arguments(emul_ctx, oracle):
    import obs_color_monitor_amd as ocm

    plane = oracle.fill_frame(16, 8, 1, 1)
    for bad in ((0, 0, 0, 0, 0), (2, 8, 0, 1, 1), (2, 0, 0, 9, 1), (2, 7, 3, 2, 2), (32, 0, 0, 0, 0)):
        with pytest.raises(ocm.ScopeError):
            emul_ctx.stage_region(plane, 16, *bad)


@pytest.mark.parametrize("w,h,dist,bands", [(70, 36, 5, 3), (130, 21, 2, 0), (1, 1, 1, 0)])
def test_waveform_histogram_one_pass(emul_ctx, oracle, w, h, dist, bands):
    with pc.options(emul_ctx, slices=bands):  # row bands: partial waveform tables, the histogram still complete
        pc.check_waveform_histogram(emul_ctx, oracle, w, h, dist)


def test_surface_waveform_feeds_histogram(emul_ctx, oracle):
    pc.check_surface_waveform_feeds_histogram(emul_ctx, oracle, 70, 36)


def test_display_post_passes(emul_ctx, oracle):
    pc.check_display(emul_ctx, oracle, 35, 20, quick=True)


def test_graticule_drawn_over_the_picture(emul_ctx):
    pc.check_graticule(emul_ctx, quick=True)


@pytest.mark.parametrize("fold", [0, 1])
def test_partial_tables_combined_in_kernel_or_by_reduce_kernel(emul_ctx, oracle, fold):
    """slices > 1: partial tables are combined by a reduce kernel or (histogram, few partials) by the last workgroup."""
    with pc.options(emul_ctx, fold=fold, slices=5):
        for _ in range(3):  # the arrival tickets must be back at zero after every launch
            pc.check_vectorscope(emul_ctx, oracle, 70, 100, 2)  # saturated bins: saturating add of clamped partials
            pc.check_histogram_counts(emul_ctx, oracle, 70, 100, 5, 0x07)
    with pc.options(emul_ctx, fold=fold):
        pc.check_batches(emul_ctx, oracle, 33, 9, 3)


@pytest.mark.parametrize("cs", [1, 2])
def test_k0_all_colours_fixed_point_path(emul_ctx, oracle, cs):
    """Every one of the 2^24 colours through K0.  In the emulator a wave is one lane, so each texel decides on its
    own between the fixed-point result and the fp32 redo: every colour the guard does not flag is checked on the
    fast path alone (on the GPU a flagged neighbour would pull the whole wave onto the fp32 path)."""
    v = np.arange(1 << 24, dtype=np.uint32) | np.uint32(0xFF000000)
    rgb = v.view(np.uint8).reshape(4096, 4096 * 4)
    pc.assert_same(emul_ctx.convert_rgb_yuv(rgb, 4096, cs), oracle.rgb_to_yuv(rgb, 4096, cs), f"K0 exhaustive cs{cs}")


@pytest.mark.parametrize("block", [256, 512, 1024])
def test_vectorscope_workgroup_sizes(emul_ctx, oracle, block):
    """vss_write_out for every workgroup size: 16 / 8 / 4 table groups per thread -- dense partials below 512 threads, a mask byte of
    eight groups at 512, the in-launch fold only at 1024 (round 6: the loop over a thread's groups once stopped at eight)."""
    for slices in (1, 3):
        with pc.options(emul_ctx, **{"vss.block": block, "slices": slices}):
            pc.check_vectorscope(emul_ctx, oracle, 160, 90, 1)
            pc.check_vectorscope(emul_ctx, oracle, 160, 90, 2, fuse=True)


def test_vectorscope_partials_meet_inside_the_launch(emul_ctx, oracle):
    """vss_write_out's fold (round 6): 2 / 3 / 4 slices per frame, every producer of vectorscope partials, against the oracle and
    against the reduction kernel's path."""
    dev = lambda *a, **k: pc.check_vectorscope_fold(emul_ctx, oracle, _np_dev, lambda hnd: hnd, *a, **k)  # noqa: E731
    dev(128, 72, [1, 2, 5], launches=2)
    dev(100, 37, [3, {"detail": 0.05}], launches=2, linesize=100 * 4 + 16)  # flat (one bin saturates in every slice), screen-like
    dev(64, 2100, [3], launches=1)                                            # tall and flat: clamp passes before the fold


def test_frame_pool_alloc_and_free(emul_ctx):
    """scope_frame_pool_alloc: several candidate allocations, each read and timed, the fastest kept and the rest freed;
    the kept pool is ordinary device memory a scope reads from."""
    import ctypes as C

    pool, rates = emul_ctx.frame_pool_alloc(64 * 36 * 4, candidates=3)
    assert pool.ptr and len(rates) == 3 and rates[0] == max(rates)
    w, h = 64, 36
    import scope_oracle

    o = scope_oracle.COracle()
    rgb = o.fill_frame(w, h, 1, 9)
    yuv = o.rgb_to_yuv(rgb, w, 2)
    C.memmove(pool.ptr, yuv.ctypes.data, yuv.nbytes)
    b = emul_ctx._batch([pool.ptr], 4 * w, w, h, False, 2)
    out = np.zeros(65536, dtype=np.uint8)
    emul_ctx.vectorscope_batch_dev(b, out.ctypes.data)
    assert np.array_equal(out.reshape(256, 256), o.vectorscope(yuv, w))
    pool.free()
    one, rates = emul_ctx.frame_pool_alloc(4096, candidates=1)  # a plain allocation: nothing to compare
    assert one.ptr and rates == [0.0]
    one.free()
    with pytest.raises(Exception):
        emul_ctx.frame_pool_alloc(0, 2)


def test_vectorscope_side_reduce(emul_ctx, oracle):
    pc.check_vectorscope_side_reduce(emul_ctx, oracle, _np_dev, lambda hnd: hnd, 96, 40, [1, 5, 2], calls=7)
    pc.check_vectorscope_side_reduce(emul_ctx, oracle, _np_dev, lambda hnd: hnd, 64, 30, [5, 3], calls=4, fuse=True)


def test_pipeline_without_histogram(emul_ctx, oracle):
    pc.check_pipeline_without_histogram(emul_ctx, oracle, 72, 20)


@pytest.mark.parametrize("forced", [0, 1])
def test_plain_vectorscope_screen_content(emul_ctx, oracle, forced):
    """The staged-plane vectorscope on screen-like content: the run-length form of its absorb mode (mixed windows), with the
    batch's unroll (the form that carries it), the default, and one workgroup per frame."""
    for slices, unroll in ((0, 0), (1, 4), (0, 4), (7, 1)):
        with pc.options(emul_ctx, **{"slices": slices, "vss.unroll": unroll}):
            pc.check_plain_screen_content(emul_ctx, oracle, [(256, 96, None), (132, 50, 132 * 4 + 16), (1024, 40, None)], forced)


def test_reference_colour_targets(emul_ctx):
    """The kernels (compiled for the CPU) against the graticule's colour targets, the expected chroma the reference itself holds."""
    pc.check_reference_colour_targets(emul_ctx)
