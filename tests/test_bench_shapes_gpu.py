"""-m gpu: the EXACT launches bench.py times -- 64 distinct 3840x2160 frames per call, every workload and variant of
its line -- with ALL 64 tables of every call compared with the oracle, and the HIP tables of the seeded full-size frames
hashed against tests/golden/fullsize_hashes.json directly (committed data, no oracle call in between).

The reference loops these tables restate: src/vectorscope.c:217-238, src/waveform.c:220-257, src/histogram.c:357-395;
K0: data/common.effect:23-43.  bench.py's own check looks at a sample of a step's frames; this file is where the whole
launch is verified on hardware.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H, N = 3840, 2160, 64
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def fnv1a64(a) -> str:
    """FNV-1a, 64 bit, over the array's bytes: the hash tests/golden/make_golden.py stored (written out here so that the
    comparison with the fixture does not pass through oracle/)."""
    h = 0xCBF29CE484222325
    for b in np.ascontiguousarray(a).tobytes():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"


@pytest.fixture(scope="module")
def pool64(hip_ctx, oracle):
    """64 distinct 4K frames, SURVEY.md 8(d)'s distributions D1..D5 in turn (uniform, concentrated, flat, bars, alpha holes),
    resident in HBM as bench.py holds them: one contiguous [64][H][4W] tensor per plane kind."""
    import torch

    rgbs = [oracle.fill_frame(W, H, 1 + f % 5, 1000 + f) for f in range(N)]
    d_rgb = torch.empty((N, H, 4 * W), dtype=torch.uint8, device="cuda")
    for f, r in enumerate(rgbs):
        d_rgb[f].copy_(torch.from_numpy(r))
    # the staged Y'CbCr planes of both matrices: made by the oracle's K0 (the reference's K1 input is OBS's shader output)
    d_yuv = {}
    yuvs = {}
    for cs in (1, 2):
        d_yuv[cs] = torch.empty((N, H, 4 * W), dtype=torch.uint8, device="cuda")
        # host copies of the 709 planes only (the luma waveform's check needs them); 601 planes are remade per frame
        for f, r in enumerate(rgbs):
            y = oracle.rgb_to_yuv(r, W, cs)
            d_yuv[cs][f].copy_(torch.from_numpy(y))
            if cs == 2:
                yuvs[f] = y
    torch.cuda.synchronize()
    yield {"rgb": rgbs, "yuv709": yuvs, "d_rgb": d_rgb, "d_yuv": d_yuv}
    del d_rgb, d_yuv


def _all_frames(got, want_of, what):
    bad = [f for f in range(N) if not np.array_equal(got[f], want_of(f))]
    assert not bad, f"{what}: frames {bad} of {N} differ from the oracle"


def test_bench_shapes_64x4k_vectorscope(hip_ctx, oracle, pool64):
    """bench.py's headline launch (staged planes) and its vectorscope_rgb variant (SCOPE_FUSE_K0), both matrices, all 64 tables."""
    import torch

    out = torch.empty((N, 65536), dtype=torch.uint8, device="cuda")
    want = {}
    for cs in (2, 1):
        yuv_of = (lambda f: pool64["yuv709"][f]) if cs == 2 else (lambda f, cs=cs: oracle.rgb_to_yuv(pool64["rgb"][f], W, cs))
        want[cs] = [oracle.vectorscope(yuv_of(f), W).reshape(-1) for f in range(N)]
        b = hip_ctx.make_device_batch(pool64["d_yuv"][cs], 4 * W, W, H, planes_are_rgb=False, colorspace=cs)
        out.fill_(0xEE)
        hip_ctx.vectorscope_batch_dev(b, out, sync=False)  # bench.py's call: SCOPE_ASYNC
        hip_ctx.sync()
        _all_frames(out.cpu().numpy(), lambda f: want[cs][f], f"vectorscope x64, staged planes, colorspace {cs}")
        b = hip_ctx.make_device_batch(pool64["d_rgb"], 4 * W, W, H, planes_are_rgb=True, colorspace=cs)
        out.fill_(0xEE)
        hip_ctx.vectorscope_batch_dev(b, out, sync=False)
        hip_ctx.sync()
        _all_frames(out.cpu().numpy(), lambda f: want[cs][f], f"vectorscope x64, SCOPE_FUSE_K0, colorspace {cs}")
    # the packed wire format of the multi-GPU sum (bench.py --gpus N, config 4): two bins per word
    b = hip_ctx.make_device_batch(pool64["d_yuv"][2], 4 * W, W, H, planes_are_rgb=False)
    packed = torch.empty((N, 32768), dtype=torch.int32, device="cuda")
    hip_ctx.vectorscope_batch_dev(b, packed, sync=False, packed16=True)
    hip_ctx.sync()
    p = packed.cpu().numpy().view(np.uint32)
    got = np.stack([p & 0xFFFF, p >> 16], axis=-1).reshape(N, 65536)
    _all_frames(got, lambda f: want[2][f], "vectorscope x64, SCOPE_OUT_PACKED16")


def test_bench_shapes_64x4k_padded_pitch(hip_ctx, oracle, pool64):
    """bench.py's padded-pitch leg at its real size (linesize = 4 W + 256, src/vectorscope.c:225,236): the
    same 64 frames behind a pitch, so the tables must be the ones of the packed planes -- the staged and the fused vectorscope
    (scope_stream's ROWS form: 2160 rows in 32 lane-group shares of 67 and 68) and the histogram, all 64 tables each; then one
    frame per call and the ring of three, which keep the general form (8.4 rows per workgroup are no whole-row shares), and
    launches of 16 and 24 frames (other slice counts of the ROWS form)."""
    import torch

    from obs_color_monitor_amd.synth import pad_pitch

    ls = 4 * W + 256
    out = torch.empty((N, 65536), dtype=torch.uint8, device="cuda")
    want = [oracle.vectorscope(pool64["yuv709"][f], W).reshape(-1) for f in range(N)]
    for key, is_rgb in (("d_yuv", False), ("d_rgb", True)):
        padded = pad_pitch(torch, pool64[key] if is_rgb else pool64[key][2], W)
        padded[:, :, 4 * W:] = 0x5A  # whatever lies in the padding is nobody's texel
        b = hip_ctx.make_device_batch(padded, ls, W, H, planes_are_rgb=is_rgb, colorspace=2)
        out.fill_(0xEE)
        torch.cuda.synchronize()  # the planes were made on torch's stream, the library launches on its context's
        hip_ctx.vectorscope_batch_dev(b, out, sync=False)
        hip_ctx.sync()
        _all_frames(out.cpu().numpy(), lambda f: want[f], f"vectorscope x64, pitch {ls}, {'SCOPE_FUSE_K0' if is_rgb else 'staged planes'}")
        for nf in (1, 3, 16, 24):  # 16 / 24 frames: 16 / 10 slices per frame, lane-group shares of 16-17 / 27 rows (ROWS form)
            b = hip_ctx.make_device_batch(padded[:nf], ls, W, H, planes_are_rgb=is_rgb, colorspace=2)
            out[:nf].fill_(0xEE)
            torch.cuda.synchronize()
            hip_ctx.vectorscope_batch_dev(b, out[:nf], sync=False)
            hip_ctx.sync()
            got = out[:nf].cpu().numpy()
            assert all(np.array_equal(got[f], want[f]) for f in range(nf)), f"vectorscope x{nf}, pitch {ls}, rgb {is_rgb}"
        if is_rgb:
            his = torch.empty((N, 1024), dtype=torch.int32, device="cuda")
            b = hip_ctx.make_device_batch(padded, ls, W, H, planes_are_rgb=True, colorspace=2)
            for comp in (0x07, 0x20):
                his.fill_(-1)
                torch.cuda.synchronize()
                hip_ctx.histogram_counts_batch_dev(b, comp, his, sync=False)
                hip_ctx.sync()
                got = his.cpu().numpy().view(np.uint32)
                _all_frames(got, lambda f: oracle.histogram_counts(pool64["rgb"][f], pool64["yuv709"][f], W, comp).reshape(-1),
                            f"histogram {comp:#04x} x64, pitch {ls}")
        del padded


@pytest.mark.parametrize("comp", [0x07, 0x20, 0x04, 0x02, 0x01])
def test_bench_shapes_64x4k_waveform(hip_ctx, oracle, pool64, comp):
    """bench.py's waveform variants: RGB planes in every mode (the luma mode derives Y' through the fused K0), and the
    staged Y'CbCr planes for the luma mode as the reference's K2 reads them."""
    import torch

    out = torch.empty((N, 256 * W * 4), dtype=torch.uint8, device="cuda")
    want = [oracle.waveform(pool64["rgb"][f], pool64["yuv709"][f], W, comp).reshape(-1) for f in range(N)]
    kinds = [("d_rgb", True)] + ([("d_yuv", False)] if comp & 0x70 else [])
    for key, is_rgb in kinds:
        planes = pool64[key] if is_rgb else pool64[key][2]
        b = hip_ctx.make_device_batch(planes, 4 * W, W, H, planes_are_rgb=is_rgb, colorspace=2)
        out.fill_(0xEE)
        hip_ctx.waveform_batch_dev(b, comp, out, sync=False)
        hip_ctx.sync()
        got = out.cpu().numpy()
        _all_frames(got, lambda f: want[f], f"waveform {comp:#04x} x64, {'RGB' if is_rgb else 'staged YUV'} planes")


def test_bench_shapes_64x4k_histogram(hip_ctx, oracle, pool64):
    import torch

    out = torch.empty((N, 1024), dtype=torch.int32, device="cuda")
    b = hip_ctx.make_device_batch(pool64["d_rgb"], 4 * W, W, H, planes_are_rgb=True, colorspace=2)
    for comp in (0x07, 0x20):
        out.fill_(-1)
        hip_ctx.histogram_counts_batch_dev(b, comp, out, sync=False)
        hip_ctx.sync()
        got = out.cpu().numpy().view(np.uint32)
        _all_frames(got, lambda f: oracle.histogram_counts(pool64["rgb"][f], pool64["yuv709"][f], W, comp).reshape(-1),
                    f"histogram {comp:#04x} x64")


def test_bench_shapes_64x4k_three_scopes(hip_ctx, oracle, pool64):
    """Config 5's kernel at the bench's launch: the three tables of all 64 frames from one read (scope_trio_kernel)."""
    import torch

    vss = torch.full((N, 65536), 0xEE, dtype=torch.uint8, device="cuda")
    wvs = torch.full((N, 256 * W * 4), 0xEE, dtype=torch.uint8, device="cuda")
    his = torch.full((N, 1024), -1, dtype=torch.int32, device="cuda")
    b = hip_ctx.make_device_batch(pool64["d_rgb"], 4 * W, W, H, planes_are_rgb=True, colorspace=2)
    hip_ctx.three_scopes_batch_dev(b, 0x07, 0x07, vss, wvs, his, sync=False)
    hip_ctx.sync()
    rgb, yuv = pool64["rgb"], pool64["yuv709"]
    _all_frames(vss.cpu().numpy(), lambda f: oracle.vectorscope(yuv[f], W).reshape(-1), "three scopes x64: vectorscope")
    _all_frames(his.cpu().numpy().view(np.uint32), lambda f: oracle.histogram_counts(rgb[f], yuv[f], W, 0x07).reshape(-1),
                "three scopes x64: histogram")
    _all_frames(wvs.cpu().numpy(), lambda f: oracle.waveform(rgb[f], yuv[f], W, 0x07).reshape(-1), "three scopes x64: waveform")


# ------------------------------------------------------------------ HIP against the committed fixtures, directly ----


@pytest.mark.parametrize("key", ["d1_1280x720", "d1_3840x2160", "d2_3840x2160", "d5_3840x2160", "d2_7680x4320"])
def test_hip_tables_hash_to_the_committed_fixtures(hip_ctx, oracle, key):
    """The seeded full-size frames (the generator is the only thing taken from oracle/: its output is pinned by the "rgb"
    hash) through the HIP library, every table hashed with FNV-1a64 HERE and compared with tests/golden/fullsize_hashes.json:
    HIP against committed data with no oracle table in between."""
    want = json.load(open(os.path.join(GOLD, "fullsize_hashes.json")))[key]
    dist, geom = key.split("_")
    w, h = (int(v) for v in geom.split("x"))
    rgb = oracle.fill_frame(w, h, int(dist[1:]), 0)
    assert f"{oracle.fnv(rgb):016x}" == want["rgb"], "the frame generator drifted from the fixture"
    yuv = hip_ctx.convert_rgb_yuv(rgb, w, 2)  # the product's K0
    assert fnv1a64(yuv) == want["yuv709"]
    vss, st = hip_ctx.vectorscope(yuv=yuv, width=w)
    assert st == 0 and fnv1a64(vss) == want["vectorscope"]
    vss, st = hip_ctx.vectorscope(rgb=rgb, width=w, colorspace=2, fuse_k0=True)
    assert st == 0 and fnv1a64(vss) == want["vectorscope"]
    his, st = hip_ctx.histogram_counts(rgb=rgb, yuv=yuv, width=w, components=0x07)
    assert st == 0 and fnv1a64(his) == want["histogram_07"]
    for comp in (0x07, 0x20):
        if f"waveform_{comp:02x}" in want:
            wvs, st = hip_ctx.waveform(rgb=rgb, yuv=yuv, width=w, components=comp)
            assert st == 0 and fnv1a64(wvs) == want[f"waveform_{comp:02x}"]
