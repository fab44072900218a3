"""-m gpu: the real gfx950 kernels through the C-ABI against the oracle, bit-exact.

Small cases mirror tests/test_parity_emul.py; the rest are BASELINE.json's full sizes
(1280x720 histogram, 3840x2160 vectorscope / waveform / histogram, 7680x4320 row-tiled vectorscope),
an exhaustive K0 check over all 2^24 colours, and size-independent properties.
"""
import os

import numpy as np
import pytest

import parity_cases as pc

pytestmark = pytest.mark.gpu

SMALL = [(64, 36, None), (70, 9, None), (33, 5, 140), (64, 8, 320), (1, 1, None), (130, 3, 524), (256, 144, None)]


def _torch_dev():
    import torch

    def to_dev(a):
        t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
        return t, int(t.data_ptr())

    def to_host(t):
        torch.cuda.synchronize()
        return t.cpu().numpy()

    return to_dev, to_host


# ---------------------------------------------------------------------------------- small cases --


@pytest.mark.parametrize("dist", [1, 2, 3, 4, 5])
def test_vectorscope_distributions(hip_ctx, oracle, dist):
    pc.check_vectorscope(hip_ctx, oracle, 256, 144, dist)


@pytest.mark.parametrize("w,h,ls", SMALL)
def test_geometry_all_scopes(hip_ctx, oracle, w, h, ls):
    pc.check_vectorscope(hip_ctx, oracle, w, h, 1, ls)
    pc.check_waveform(hip_ctx, oracle, w, h, 1, 0x07, ls)
    pc.check_histogram_counts(hip_ctx, oracle, w, h, 2, 0x07, ls)
    pc.check_k0(hip_ctx, oracle, w, h, 1, 2, ls)


@pytest.mark.parametrize("merge", [0, 1, 2, 3])
@pytest.mark.parametrize("unroll", [1, 2, 4])
@pytest.mark.parametrize("hash_", [0, 1])
def test_vectorscope_variants(hip_ctx, oracle, merge, unroll, hash_):
    with pc.options(hip_ctx, **{"vss.merge": merge, "vss.unroll": unroll, "vss.hash": hash_}):
        for dist in (1, 2, 3, 4):
            pc.check_vectorscope(hip_ctx, oracle, 640, 360, dist)
        pc.check_vectorscope(hip_ctx, oracle, 70, 9, 3)


@pytest.mark.parametrize("groups", [0, 1, 2, 3])
@pytest.mark.parametrize("nt", [0, 1])
def test_vectorscope_lane_groups_and_nt_loads(hip_ctx, oracle, groups, nt):
    with pc.options(hip_ctx, **{"vss.groups": groups, "nt": nt}):
        for dist in (1, 3, 4):
            pc.check_vectorscope(hip_ctx, oracle, 640, 360, dist)
        pc.check_vectorscope(hip_ctx, oracle, 33, 5, 1, 140)
        pc.check_vectorscope(hip_ctx, oracle, 1279, 300, 2, 5120)
        pc.check_histogram_counts(hip_ctx, oracle, 640, 360, 5, 0x07)
        pc.check_waveform(hip_ctx, oracle, 640, 360, 5, 0x07)


@pytest.mark.parametrize("slices", [1, 2, 7, 256])
def test_partition_independence(hip_ctx, oracle, slices):
    """Any split of a frame over workgroups gives the same tables (saturating/plain sums commute)."""
    with pc.options(hip_ctx, slices=slices):
        pc.check_vectorscope(hip_ctx, oracle, 640, 360, 2)
        pc.check_vectorscope(hip_ctx, oracle, 640, 360, 3)
        pc.check_histogram_counts(hip_ctx, oracle, 640, 360, 5, 0x07)
        pc.check_waveform(hip_ctx, oracle, 640, 360, 3, 0x07)
        pc.check_waveform(hip_ctx, oracle, 330, 200, 5, 0x70)


def test_vectorscope_epochs_single_workgroup(hip_ctx, oracle):
    """One workgroup sees > 65280 pixels of one bin: 16-bit LDS counters + clamp passes stay exact."""
    for merge in (0, 1, 2):
        for hash_ in (0, 1):
            with pc.options(hip_ctx, **{"slices": 1, "vss.merge": merge, "vss.hash": hash_, "vss.groups": merge}):
                pc.check_vectorscope(hip_ctx, oracle, 1280, 720, 3)
                pc.check_vectorscope(hip_ctx, oracle, 1279, 300, 2, 5120)


@pytest.mark.parametrize("comp", pc.COMPONENTS)
def test_component_masks(hip_ctx, oracle, comp):
    pc.check_waveform(hip_ctx, oracle, 330, 77, 5, comp)
    pc.check_histogram_counts(hip_ctx, oracle, 330, 77, 5, comp)


@pytest.mark.parametrize("fold", [0, 1])
def test_partial_tables_combined_in_kernel_or_by_reduce_kernel(hip_ctx, oracle, fold):
    """slices > 1: the vectorscope's partial tables are combined by a reduce kernel; the histogram's slices meet
    inside the launch (fold=1: device-scope atomics + the last workgroup of the frame) or through partial tables and a
    reduce kernel (fold=0).  Repeated launches check that tickets and accumulators are back at zero each time;
    256 slices per frame is the single-frame case on 256 CUs."""
    with pc.options(hip_ctx, fold=fold):
        for _ in range(4):
            pc.check_vectorscope(hip_ctx, oracle, 1920, 1080, 2)  # one frame -> 256 slices, saturated bins
            pc.check_histogram_counts(hip_ctx, oracle, 1920, 1080, 5, 0x07)
        pc.check_batches(hip_ctx, oracle, 640, 360, 7)
        pc.check_batches(hip_ctx, oracle, 320, 180, 32)  # 8 slices per frame
        with pc.options(hip_ctx, fold=fold, slices=3):
            for _ in range(3):
                pc.check_batches(hip_ctx, oracle, 640, 360, 5)


@pytest.mark.parametrize("sparse", [0, 1])
@pytest.mark.parametrize("slices", [2, 3, 5, 16, 17, 85, 128, 255, 256])
def test_vectorscope_sparse_partials(hip_ctx, oracle, sparse, slices):
    """Partial tables that hold only the non-empty 16-bin groups (+ a mask byte per thread), in the blocked layout the
    parallel reduce kernel reads, against dense partials: every slice count that changes the reduction's shape
    (slice lanes 2..16, ragged last round), flat / concentrated / noisy / alpha-holed content, plain and fused K0."""
    with pc.options(hip_ctx, **{"vss.sparse": sparse, "slices": slices}):
        for dist in (1, 2, 3, 5):
            pc.check_vectorscope(hip_ctx, oracle, 640, 360, dist)
        pc.check_vectorscope(hip_ctx, oracle, 1279, 300, 2, 5120)
        pc.check_vectorscope(hip_ctx, oracle, 640, 360, 1, fuse=True)
        pc.check_vectorscope(hip_ctx, oracle, 33, 5, 1, 140)  # most slices are empty


@pytest.mark.parametrize("block", [256, 512, 768, 1024])
def test_vectorscope_sparse_partials_workgroup_sizes(hip_ctx, oracle, block):
    """The mask byte of a thread covers its 4096 / blockDim groups: 8 at 512 threads; below that the launch falls back
    to dense partials."""
    with pc.options(hip_ctx, **{"vss.block": block}):
        pc.check_vectorscope(hip_ctx, oracle, 1920, 1080, 1)
        pc.check_vectorscope(hip_ctx, oracle, 640, 360, 2)
        with pc.options(hip_ctx, slices=7):
            pc.check_vectorscope(hip_ctx, oracle, 640, 360, 5)


def test_one_and_three_frame_launches_4k(hip_ctx, oracle):
    """The reference's call shape (one surface per callback, ring of 3: src/common.c:366-368, src/common.h:46) at
    3840x2160 through the device-pointer batch entry points, all three scopes, tables checked for every frame."""
    import torch

    w, h = 3840, 2160
    rgbs = [oracle.fill_frame(w, h, d, 40 + d) for d in (1, 2, 5)]
    yuvs = [oracle.rgb_to_yuv(r, w, 2) for r in rgbs]
    ty, tr = torch.from_numpy(np.stack(yuvs)).cuda(), torch.from_numpy(np.stack(rgbs)).cuda()
    for n in (1, 3):
        by = hip_ctx.make_device_batch(ty[:n], 4 * w, w, h, planes_are_rgb=False)
        br = hip_ctx.make_device_batch(tr[:n], 4 * w, w, h, planes_are_rgb=True, colorspace=2)
        vss = torch.empty((n, 65536), dtype=torch.uint8, device="cuda")
        cnt = torch.empty((n, 1024), dtype=torch.int32, device="cuda")
        wvs = torch.empty((n, 256 * w * 4), dtype=torch.uint8, device="cuda")
        for rep in range(3):  # accumulators and tickets must come back to zero
            for batch in (by, br):
                vss.fill_(0xEE)
                torch.cuda.synchronize()  # torch fills on ITS stream; the context launches on its own (non-blocking) one
                hip_ctx.vectorscope_batch_dev(batch, vss)
                for f in range(n):
                    pc.assert_same(vss[f].cpu().numpy().reshape(256, 256), oracle.vectorscope(yuvs[f], w), f"vss n{n} f{f}")
            cnt.fill_(-1)
            torch.cuda.synchronize()  # torch fills on ITS stream; the context launches on its own (non-blocking) one
            hip_ctx.histogram_counts_batch_dev(br, 0x07, cnt)
            for f in range(n):
                pc.assert_same(cnt[f].cpu().numpy().view(np.uint32).reshape(256, 4),
                               oracle.histogram_counts(rgbs[f], None, w, 0x07), f"his n{n} f{f}")
        hip_ctx.waveform_batch_dev(br, 0x07, wvs)
        for f in range(n):
            pc.assert_same(wvs[f].cpu().numpy().reshape(256, w, 4), oracle.waveform(rgbs[f], None, w, 0x07), f"wvs n{n} f{f}")


def test_randomised_geometries(hip_ctx, oracle):
    """tools/fuzz_parity.py: random geometries (vector and scalar shapes, heights around the limits of the one-frame
    kernels), pitches, masks, matrices and launch sizes through every accumulation entry point."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    assert fz.run(hip_ctx, oracle, 60, 5, lambda m: None) == 60


@pytest.mark.parametrize("dock", [1, 2, 0])
def test_three_scopes_one_read(hip_ctx, oracle, dock):
    """scope_three_scopes_batch: the dock's three scopes of a frame from ONE read of the plane.  dock=1: scope_trio_kernel
    (both tables resident in LDS, any number of frames); dock=2: round 2's scope_dock_kernel for launches of one round of
    strips (the strip's texels stay in registers, waveform + histogram first, the vectorscope through the fused K0 second;
    other launches take two passes); dock=0 forces the two-pass form.  All against the oracle: 4K on every content kind, ragged last strips,
    heights that leave row ranges partly empty, padded pitches, BT.601, every RGB mask, two small frames in one launch,
    a frame too tall for the registers (falls back), host planes."""
    to_dev, to_host = _torch_dev()

    def dev(*a, **k):
        pc.check_three_scopes_dev(hip_ctx, oracle, to_dev, to_host, *a, **k)

    with pc.options(hip_ctx, dock=dock):
        dev(3840, 2160, [1])
        dev(3840, 2160, [5], 0x02, 0x02)
        for d in (2, 3, 4):
            dev(1920, 1080, [d], reps=1)
        dev(1924, 1083, [1], 0x07, 0x05, 1924 * 4 + 64, colorspace=1)
        for comp in (0x01, 0x02, 0x03, 0x04, 0x05, 0x06):
            dev(328, 77, [5], comp, comp, reps=1)
        dev(1280, 720, [1, 5])       # two frames, 160 strips
        dev(4, 5, [1])
        dev(64, 2300, [1], reps=1)   # more than 2176 rows: two passes
        dev(330, 77, [5], reps=1)    # width % 4 != 0: two passes
        dev(3840, 2160, [1, 2, 5], reps=1)  # the ring of three: 80 slices of three strips per frame (dock=2: two passes)
        pc.check_three_scopes_host(hip_ctx, oracle, 640, 360, 5)


def test_three_scopes_batches_one_read(hip_ctx, oracle):
    """scope_trio_kernel in its batch shapes: twelve 4K frames of every content kind (20 paired slices of twelve strips per
    frame, a clamp pass of the vectorscope table before every strip but the first; flat frames saturate bins and columns),
    64 small frames (four slices per frame, the launch the bench's three_scope workload makes at 4K), seven frames (252
    workgroups: the last block of the pairing is a partial one), a batch with a padded pitch and a ragged last strip, and
    more frames than one launch takes."""
    to_dev, to_host = _torch_dev()

    def dev(*a, **k):
        pc.check_three_scopes_dev(hip_ctx, oracle, to_dev, to_host, *a, **k)

    dev(3840, 2160, [1, 2, 3, 4, 5, 1, 3, 2, 5, 4, 1, 2], reps=1)
    folds = hip_ctx.get_stat("trio.fold_launches")
    dev(320, 180, [(i % 5) + 1 for i in range(64)], reps=2)
    assert hip_ctx.get_stat("trio.fold_launches") == folds + 2, "64 frames = four slices per frame: the vectorscope partials meet inside the launch"
    dev(1280, 720, [1, 2, 3, 4, 5, 1, 2], 0x05, 0x04, reps=1)
    dev(1000, 300, [5, 1, 2, 3], 0x07, 0x07, 1000 * 4 + 48, colorspace=1, reps=1)
    dev(256, 144, [(i % 5) + 1 for i in range(70)], reps=1)
    dev(7680, 4320, [1], reps=1)  # BASELINE config 4's frame: 480 strips, 240 slices of two, clamp passes inside a strip


@pytest.mark.parametrize("strip", [16, 32])
def test_three_scopes_strip_widths(hip_ctx, oracle, strip):
    """scope_trio_kernel's two shapes forced on launches of every size: 16-column strips with waveform replicas, 32-column strips
    with one replica (what batches run by default); 4K content of every kind, one- and two-channel masks, ragged widths."""
    to_dev, to_host = _torch_dev()

    def dev(*a, **k):
        pc.check_three_scopes_dev(hip_ctx, oracle, to_dev, to_host, *a, **k)

    with pc.options(hip_ctx, **{"trio.strip": strip}):
        dev(3840, 2160, [1, 2, 3, 4, 5, 1, 3, 2], reps=1)
        dev(3840, 2160, [5], 0x02, 0x02, reps=1)
        dev(3840, 2160, [1, 3], 0x05, 0x04, reps=1)
        dev(1924, 1083, [1, 2], 0x07, 0x05, 1924 * 4 + 64, colorspace=1, reps=1)
        for comp in (0x01, 0x03, 0x04, 0x06):
            dev(328, 77, [5, 3], comp, comp, reps=1)
        dev(100, 5, [1], reps=2)
        dev(320, 180, [(i % 5) + 1 for i in range(64)], reps=1)
        dev(7680, 4320, [2], reps=1)


@pytest.mark.parametrize("forced", [0, 1])
@pytest.mark.parametrize("strip", [0, 16, 32])
def test_three_scopes_screen_content(hip_ctx, oracle, strip, forced):
    """scope_trio_kernel's flat-content modes at work: screen-like content -- flat areas, rectangles, sprinkled detail, greys, random
    alpha -- where most lanes of a wave hold flat texel pairs and some do not: the run-length form (trio_rle_pair, round 6: runs of
    equal pairs across the lanes of a DPP row, one add per run) where the vote finds such a wave, the per-lane absorb mode where the
    whole wave is flat, plain adds elsewhere; batches and one-frame calls, both strip widths, both matrices, one- and two-channel
    masks.  forced 1: the run-length form on every slot of any content (vss.rle = 1)."""
    to_dev, to_host = _torch_dev()

    def dev(*a, **k):
        pc.check_three_scopes_dev(hip_ctx, oracle, to_dev, to_host, *a, **k)

    kinds = [{}, {"detail": 0.0}, {"detail": 0.3}, {"grey": True}, {"alpha_noise": True, "detail": 0.01}, {"rects": 200, "detail": 0.0},
             {"rects": 300, "detail": 0.03}, {"rects": 60, "detail": 0.1}]
    with pc.options(hip_ctx, **{"trio.strip": strip, "vss.rle": forced}):
        dev(3840, 2160, kinds, reps=1)
        dev(3840, 2160, kinds[:1], reps=1)
        dev(3840, 2160, [kinds[6], 3, kinds[0]], 0x05, 0x04, colorspace=1, reps=1)
        dev(1924, 1083, kinds[:3], 0x07, 0x05, 1924 * 4 + 64, reps=1)
        dev(328, 77, kinds[4:7], 0x02, 0x02, reps=2)
        if forced:
            dev(3840, 2160, [1, 2, 5, 4], reps=1)        # noise, a tight cluster, alpha holes, bars: rough pairs and short runs
            dev(100, 37, [1, 2, 5], 0x07, 0x03, 100 * 4 + 24, 1, reps=1)  # a ragged last strip, lanes without texels


@pytest.mark.parametrize("strip", [16, 64])
def test_waveform_strip_widths(hip_ctx, oracle, strip):
    """Both shapes a small launch can take -- 16-column strips (eight rows per wave instruction into four table
    replicas, no partial tables) and 64-column strips cut into row bands (partial tables + saturating reduce) -- on
    ragged last strips, left-over rows, padded pitches, saturating columns, fused K0, every component mask, with the
    histogram riding on the pass, and for one, two and three 4K frames per launch."""
    import torch

    with pc.options(hip_ctx, **{"wvs.strip": strip}):
        for comp in pc.COMPONENTS:
            pc.check_waveform(hip_ctx, oracle, 328, 77, 5, comp)
        pc.check_waveform(hip_ctx, oracle, 1924, 1083, 1, 0x07, 1924 * 4 + 64)
        pc.check_waveform(hip_ctx, oracle, 640, 1500, 3, 0x07)  # flat: every column saturates
        pc.check_waveform(hip_ctx, oracle, 4, 5, 1, 0x70)
        pc.check_waveform(hip_ctx, oracle, 1280, 720, 2, 0x70, None, fuse=True, colorspace=1)
        pc.check_waveform_histogram(hip_ctx, oracle, 1280, 723, 5)
        pc.check_waveform(hip_ctx, oracle, 3840, 2160, 5, 0x07)
        pc.check_waveform(hip_ctx, oracle, 3840, 2160, 2, 0x20)
        w, h = 3840, 2160
        rgbs = [oracle.fill_frame(w, h, d, 60 + d) for d in (1, 5, 2)]
        tr = torch.from_numpy(np.stack(rgbs)).cuda()
        for n in (1, 2, 3):
            br = hip_ctx.make_device_batch(tr[:n], 4 * w, w, h, planes_are_rgb=True, colorspace=2)
            wvs = torch.empty((n, 256 * w * 4), dtype=torch.uint8, device="cuda")
            cnt = torch.empty((n, 1024), dtype=torch.int32, device="cuda")
            for rep in range(2):
                wvs.fill_(0xEE)
                cnt.fill_(-1)
                torch.cuda.synchronize()  # torch fills on ITS stream; the context launches on its own (non-blocking) one
                hip_ctx.waveform_histogram_batch_dev(br, 0x07, 0x07, wvs, cnt)
                for f in range(n):
                    pc.assert_same(wvs[f].cpu().numpy().reshape(256, w, 4), oracle.waveform(rgbs[f], None, w, 0x07), f"wvs strip{strip} n{n} f{f}")
                    pc.assert_same(cnt[f].cpu().numpy().view(np.uint32).reshape(256, 4),
                                   oracle.histogram_counts(rgbs[f], None, w, 0x07), f"his strip{strip} n{n} f{f}")


@pytest.mark.parametrize("w,h,dist", [(640, 360, 5), (70, 9, 1), (1279, 300, 2), (330, 77, 3), (3840, 2160, 5)])
def test_waveform_histogram_one_pass(hip_ctx, oracle, w, h, dist):
    """K2 + K3 from one read of the plane (the histogram is added up from the waveform's per-column counters at
    flush time), every mask pairing incl. the ones that cannot share a pass, single frames (row bands) and batches."""
    pc.check_waveform_histogram(hip_ctx, oracle, w, h, dist)
    if w < 3000:
        with pc.options(hip_ctx, slices=1):
            pc.check_waveform_histogram(hip_ctx, oracle, w, h, dist, None, 1)


def test_waveform_histogram_batch_device(hip_ctx, oracle):
    import torch

    w, h, n = 1280, 720, 5
    rgbs = [oracle.fill_frame(w, h, 1 + i % 5, 500 + i) for i in range(n)]
    tr = torch.from_numpy(np.stack(rgbs)).cuda()
    b = hip_ctx.make_device_batch(tr, 4 * w, w, h, planes_are_rgb=True, colorspace=1)
    wv = torch.empty((n, 256 * w * 4), dtype=torch.uint8, device="cuda")
    cnt = torch.empty((n, 1024), dtype=torch.int32, device="cuda")
    for wc, hc in ((0x07, 0x07), (0x70, 0x20), (0x07, 0x70)):
        for rep in range(3):
            cnt.fill_(-1)
            torch.cuda.synchronize()  # torch fills on ITS stream; the context launches on its own (non-blocking) one
            hip_ctx.waveform_histogram_batch_dev(b, wc, hc, wv, cnt)
            for f in range(n):
                yuv = oracle.rgb_to_yuv(rgbs[f], w, 1)
                pc.assert_same(wv[f].cpu().numpy().reshape(256, w, 4), oracle.waveform(rgbs[f], yuv, w, wc), f"batch waveform {wc:#x} f{f}")
                pc.assert_same(cnt[f].cpu().numpy().view(np.uint32).reshape(256, 4), oracle.histogram_counts(rgbs[f], yuv, w, hc),
                               f"batch counts {hc:#x} f{f}")


def test_surface_waveform_feeds_histogram(hip_ctx, oracle):
    pc.check_surface_waveform_feeds_histogram(hip_ctx, oracle, 640, 360)
    pc.check_surface_waveform_feeds_histogram(hip_ctx, oracle, 1279, 301)


def test_histogram_in_kernel_combine_stress(hip_ctx, oracle):
    """The in-launch combine of the histogram hands data between workgroups with device-scope atomics and an arrival
    ticket (scope_last_arrival).  4000 back-to-back launches over rotating frames, 1..256 slices per frame, every count
    of every launch compared on the device with the tables of the two-kernel path (fold=0)."""
    import torch

    w, h, pool = 1280, 720, 8
    frames = torch.from_numpy(np.stack([oracle.fill_frame(w, h, 1 + i % 5, 900 + i) for i in range(pool)])).cuda()
    for n, slices in ((1, 0), (3, 0), (8, 0), (1, 37)):
        groups = pool // n
        batches = [hip_ctx.make_device_batch(frames[g * n:(g + 1) * n], 4 * w, w, h, planes_are_rgb=True) for g in range(groups)]
        want = torch.empty((groups, n, 1024), dtype=torch.int32, device="cuda")
        with pc.options(hip_ctx, fold=0, slices=slices):
            for g in range(groups):
                hip_ctx.histogram_counts_batch_dev(batches[g], 0x07, want[g])
        reps = 1000
        got = torch.empty((reps, n, 1024), dtype=torch.int32, device="cuda")
        with pc.options(hip_ctx, fold=1, slices=slices):
            for i in range(reps):
                hip_ctx.histogram_counts_batch_dev(batches[i % groups], 0x07, got[i], sync=False)
            hip_ctx.sync()
        idx = torch.arange(reps, device="cuda") % groups
        assert torch.equal(got, want[idx]), f"n={n} slices={slices}"
    pc.assert_same(want[0, 0].cpu().numpy().view(np.uint32).reshape(256, 4),
                   oracle.histogram_counts(frames[0].cpu().numpy(), None, w, 0x07), "fold=0 reference")


def test_waveform_taller_than_16_bit_counters(hip_ctx, oracle):
    """More than 65535 rows: the 16-bit per-band counters force at least two row bands."""
    pc.check_waveform(hip_ctx, oracle, 8, 70000, 3, 0x07)
    with pc.options(hip_ctx, slices=1):
        pc.check_waveform(hip_ctx, oracle, 300, 65546, 3, 0x20)  # one band would wrap to 10


@pytest.mark.parametrize("persist", [0, 1])
def test_waveform_unit_walk(hip_ctx, oracle, persist):
    """More units than resident workgroups (256 CUs): a workgroup walks several units with one load stream."""
    with pc.options(hip_ctx, **{"wvs.persist": persist}):
        pc.check_waveform(hip_ctx, oracle, 3840, 35, 9, 0x07)  # 540 units of 3 channels on 256 CUs
        pc.check_waveform(hip_ctx, oracle, 3838, 34, 12, 0x20)  # 720 one-channel units, ragged last strip
        with pc.options(hip_ctx, slices=7):
            pc.check_waveform(hip_ctx, oracle, 1920, 100, 3, 0x50)  # 630 units, bands of different heights


@pytest.mark.parametrize("wide", [0, 1])
def test_waveform_load_shapes(hip_ctx, oracle, wide):
    """8-byte texel-pair loads vs 4-byte loads; odd widths / 4-byte pitches take the 4-byte path either way."""
    with pc.options(hip_ctx, **{"wvs.wide": wide}):
        pc.check_waveform(hip_ctx, oracle, 640, 360, 1, 0x07)
        pc.check_waveform(hip_ctx, oracle, 64, 701, 3, 0x07)  # odd row count: the single-row tail
        pc.check_waveform(hip_ctx, oracle, 331, 77, 5, 0x50)  # odd width
        pc.check_waveform(hip_ctx, oracle, 330, 78, 5, 0x20, 330 * 4 + 4)  # pitch % 8 == 4
        pc.check_waveform(hip_ctx, oracle, 2, 1, 1, 0x07)
        pc.check_waveform(hip_ctx, oracle, 1, 3, 1, 0x07)


def test_waveform_deep_queue_batches(hip_ctx, oracle):
    """32 frames of 640x360 = 320 units on 256 persistent workgroups, six slots per wave: the deep queue across flushes, for
    the plain RGB form and for the fused one-row (luma) form; the same with it switched off."""
    pc.check_batches(hip_ctx, oracle, 640, 360, 32)
    with pc.options(hip_ctx, **{"wvs.deep": 0}):
        pc.check_batches(hip_ctx, oracle, 640, 360, 32)


@pytest.mark.parametrize("cs", [1, 2])
def test_fused_k0(hip_ctx, oracle, cs):
    pc.check_vectorscope(hip_ctx, oracle, 640, 360, 1, None, fuse=True, colorspace=cs)
    pc.check_vectorscope(hip_ctx, oracle, 640, 360, 4, None, fuse=True, colorspace=cs)
    pc.check_waveform(hip_ctx, oracle, 640, 360, 1, 0x70, None, fuse=True, colorspace=cs)
    pc.check_histogram_counts(hip_ctx, oracle, 640, 360, 1, 0x70, None, fuse=True, colorspace=cs)


@pytest.mark.parametrize("merge", [0, 2, 3])
@pytest.mark.parametrize("hash_", [0, 2])
def test_fused_vectorscope_layouts_and_modes(hip_ctx, oracle, merge, hash_):
    """vss.fused_hash 2 (the default: rows of 136 words, two pixels' addresses per register) and 0 (the plain kernel's
    layout), with the absorb mode off, forced and voted, on noise, flat and banded content, one workgroup per frame (clamp
    passes, scope_stream's unchecked main loop and its tail) and the default slicing, tails and a padded pitch."""
    for slices, groups, unroll in ((1, 3, 2), (1, 0, 4), (0, -1, 2), (3, 2, 1)):
        with pc.options(hip_ctx, **{"slices": slices, "vss.merge": merge, "vss.fused_hash": hash_, "vss.groups": groups,
                                    "vss.fused_unroll": unroll}):
            for dist in (1, 3, 4):
                pc.check_vectorscope(hip_ctx, oracle, 1280, 720, dist, None, fuse=True, colorspace=2)
            pc.check_vectorscope(hip_ctx, oracle, 509, 130, 2, 2048, fuse=True, colorspace=1)


@pytest.mark.parametrize("forced", [0, 1, 2])
def test_fused_vectorscope_screen_content(hip_ctx, oracle, forced):
    """The fused vectorscope on screen-like content -- flat areas, rectangles, sprinkled detail, greys, random alpha -- and the
    oracle's distributions, with the absorb mode voted and forced, at 4K, on small ragged frames, with one workgroup per frame
    and with the batch's unroll; then a batch of 4K screen frames as bench.py --dist screen launches them."""
    pc.check_fused_screen_content(hip_ctx, oracle, [(3840, 2160, None)], forced)
    for slices, unroll in ((0, 0), (1, 4), (1, 2), (7, 1)):
        with pc.options(hip_ctx, **{"slices": slices, "vss.fused_unroll": unroll}):
            pc.check_fused_screen_content(hip_ctx, oracle, [(640, 360, None), (509, 130, 2048)], forced)
    # a batch of eight 4K frames of screen content, as bench.py --dist screen launches them (device planes)
    import torch

    w, h, n = 3840, 2160, 8
    frames_ = [pc.screen_frame(w, h, 900 + i, detail=0.03, rects=300) for i in range(n)]
    d = torch.from_numpy(np.stack(frames_)).cuda()
    b = hip_ctx.make_device_batch(d, 4 * w, w, h, planes_are_rgb=True, colorspace=2)
    out = torch.empty((n, 256, 256), dtype=torch.uint8, device="cuda")
    # vss.fused_unroll 4: the batch form of the kernel (the one a 64-frame launch runs, and the one that carries the run-length mode)
    with pc.options(hip_ctx, **dict({"vss.fused_unroll": 4}, **({1: {"vss.rle": 1}, 2: {"vss.merge": 2}}.get(forced, {})))):
        hip_ctx.vectorscope_batch_dev(b, out)
    got = out.cpu().numpy()
    for i in range(n):
        pc.assert_same(got[i], oracle.vectorscope(oracle.rgb_to_yuv(frames_[i], w, 2), w), f"screen batch frame {i} forced={forced}")


def test_fused_vectorscope_run_length_mode_stress(hip_ctx, oracle):
    """The run-length mode keeps nothing in registers between cells but packs rough cells through a scratch row of the wave in the
    LDS: 400 back-to-back launches of the batch form with the mode forced, on noise (every cell rough: three packed rounds per
    wave-cell), screen-like content and bars, every table of every launch compared (the tables of one input must never change)."""
    import torch

    w, h, n = 1280, 720, 6
    frames_ = [oracle.fill_frame(w, h, 1, 700 + i) for i in range(2)] + [pc.screen_frame(w, h, 800 + i, detail=0.05, rects=40) for i in range(2)] + \
              [oracle.fill_frame(w, h, 4, 3), pc.screen_frame(w, h, 9, detail=0.5, rects=5, alpha_noise=True)]
    want = np.stack([oracle.vectorscope(oracle.rgb_to_yuv(f, w, 2), w) for f in frames_])
    d = torch.from_numpy(np.stack(frames_)).cuda()
    b = hip_ctx.make_device_batch(d, 4 * w, w, h, planes_are_rgb=True, colorspace=2)
    outs = [torch.empty((n, 256, 256), dtype=torch.uint8, device="cuda") for _ in range(8)]
    with pc.options(hip_ctx, **{"vss.rle": 1, "vss.fused_unroll": 4, "slices": 2}):
        for rep in range(50):
            for o in outs:
                hip_ctx.vectorscope_batch_dev(b, o, sync=False)
            hip_ctx.sync()
            for k, o in enumerate(outs):
                got = o.cpu().numpy()
                assert np.array_equal(got, want), f"launch {rep * 8 + k}: run-length mode changed a table"


@pytest.mark.parametrize("cs", [1, 2])
@pytest.mark.parametrize("hash_", [0, 1, 2])
def test_fused_vectorscope_all_colours(hip_ctx, oracle, cs, hash_):
    """All 2^24 colours through the vectorscope's own fused K0 (fixed point with its fp32 fallback): 64 slices see 262144 colours each, so no bin of a partial saturates early."""
    v = np.arange(1 << 24, dtype=np.uint32) | np.uint32(0xFF000000)
    rgb = v.view(np.uint8).reshape(4096, 4096 * 4)
    want = oracle.vectorscope(oracle.rgb_to_yuv(rgb, 4096, cs), 4096)
    with pc.options(hip_ctx, **{"vss.fused_hash": hash_}):
        got, st = hip_ctx.vectorscope(rgb=rgb, width=4096, colorspace=cs, fuse_k0=True)
    assert st == 0
    pc.assert_same(got, want, f"fused vectorscope, all colours, cs{cs} hash{hash_}")


@pytest.mark.parametrize("logscale,fixed,ratio", [(0, 0, 0), (1, 0, 0), (0, 100, 0), (1, 0, 250), (0, 0, 1), (1, 5, 9)])
def test_histogram_level_pass(hip_ctx, oracle, logscale, fixed, ratio):
    pc.check_histogram_full(hip_ctx, oracle, 640, 360, 2, 0x07, logscale, fixed, ratio)
    pc.check_histogram_full(hip_ctx, oracle, 640, 360, 1, 0x20, logscale, fixed, ratio)


def test_batches_and_error_behaviour(hip_ctx, oracle):
    pc.check_batches(hip_ctx, oracle, 320, 180, 5)
    pc.check_batches(hip_ctx, oracle, 66, 5, 70)  # more than SCOPE_MAX_FRAMES: two launches
    pc.check_error_behaviour(hip_ctx, oracle)


@pytest.mark.parametrize("w,h,ls,off", [(640, 360, 2560, 0), (35, 7, 140, 0), (640, 36, 2560, 4), (35, 7, 148, 8)])
def test_device_path_and_unaligned_rows(hip_ctx, oracle, w, h, ls, off):
    to_dev, to_host = _torch_dev()
    pc.check_device_path(hip_ctx, oracle, to_dev, to_host, w, h, ls, n=3, offset=off)


# -------------------------------------------------------------------------------- K0 exhaustive --


@pytest.mark.parametrize("cs", [1, 2])
def test_k0_all_colours(hip_ctx, oracle, cs):
    """Every one of the 2^24 RGB triples through the K0 kernel == the oracle's fp32 evaluation
    (this also covers the division-free byte/255 used on the GPU)."""
    v = np.arange(1 << 24, dtype=np.uint32) | np.uint32(0xFF000000)
    rgb = v.view(np.uint8).reshape(4096, 4096 * 4)
    want = oracle.rgb_to_yuv(rgb, 4096, cs)
    got = hip_ctx.convert_rgb_yuv(rgb, 4096, cs)
    pc.assert_same(got, want, f"K0 exhaustive cs{cs}")


def test_hip_k0_against_what_the_reference_holds(hip_ctx):
    """The HIP path against the two pieces of expected chroma the reference itself carries, no oracle in between
    (tests/test_oracle.py holds the data and the same checks for the oracle):
      * the graticule's colour targets (src/vectorscope.c:283-316): the 100 % and 75 % colour bars through scope_convert_rgb_yuv,
        and through the fused-K0 and the staged vectorscope -- the twelve bins that light up lie on the twelve targets;
      * the integer macros RGB2U/V_601 / _709 (src/vectorscope.c:28-34): all 2^24 colours through the K0 kernel stay within 3 (U)
        and 2 (V) of them."""
    pc.check_reference_colour_targets(hip_ctx)
    v = np.arange(1 << 24, dtype=np.uint32) | np.uint32(0xFF000000)
    rgb = v.view(np.uint8).reshape(4096, 4096 * 4)
    b, g, r = (rgb.reshape(-1, 4)[:, i].astype(np.int64) for i in range(3))

    def cdiv(a, d):
        return np.sign(a) * (np.abs(a) // d)

    for cs, (mu, mv) in ((1, (cdiv(-150 * r - 296 * g + 448 * b, 1024) + 128, cdiv(448 * r - 374 * g - 72 * b, 1024) + 128)),
                         (2, (cdiv(-102 * r - 346 * g + 450 * b, 1024) + 128, cdiv(450 * r - 408 * g - 40 * b, 1024) + 128))):
        got = hip_ctx.convert_rgb_yuv(rgb, 4096, cs).reshape(-1, 4)
        assert int(np.abs(got[:, 0].astype(np.int64) - mu).max()) <= 3 and int(np.abs(got[:, 2].astype(np.int64) - mv).max()) <= 2


@pytest.mark.parametrize("cs", [1, 2])
def test_fused_k0_all_colours(hip_ctx, oracle, cs):
    """The fused kernels convert on the fly (fixed-point fast path + fp32 redo near a step): all 2^24 colours through
    the histogram (exact u32 counts of U, Y and V), the luma waveform and the vectorscope."""
    v = np.arange(1 << 24, dtype=np.uint32) | np.uint32(0xFF000000)
    rgb = v.view(np.uint8).reshape(4096, 4096 * 4)
    yuv = oracle.rgb_to_yuv(rgb, 4096, cs)
    got, st = hip_ctx.histogram_counts(rgb=rgb, width=4096, components=0x70, colorspace=cs, fuse_k0=True)
    assert st == 0
    pc.assert_same(got, oracle.histogram_counts(None, yuv, 4096, 0x70), f"fused histogram, all colours, cs{cs}")
    got, st = hip_ctx.vectorscope(rgb=rgb, width=4096, colorspace=cs, fuse_k0=True)
    pc.assert_same(got, oracle.vectorscope(yuv, 4096), f"fused vectorscope, all colours, cs{cs}")
    # a shuffled copy: the unsaturated counters of the waveform see every colour in a different column
    perm = np.random.default_rng(7).permutation(1 << 24)
    rgb2 = np.ascontiguousarray(v[perm]).view(np.uint8).reshape(4096, 4096 * 4)
    yuv2 = oracle.rgb_to_yuv(rgb2, 4096, cs)
    got, st = hip_ctx.waveform(rgb=rgb2, width=4096, components=0x20, colorspace=cs, fuse_k0=True)
    pc.assert_same(got, oracle.waveform(None, yuv2, 4096, 0x20), f"fused luma waveform, all colours, cs{cs}")


def test_dock_kernel_back_to_back_stress(hip_ctx, oracle):
    """2000 one-read launches back to back on one stream, two alternating 1080p frames (120 strips each), no host
    synchronisation in between: the histogram's accumulator and arrival tickets -- fired after the waveform's flush, waited
    for at the end of the kernel -- must be back at zero for every launch, and partial tables must not be overwritten
    before the reduce kernel has read them.  Every 100th pair of results is compared on the device with tables checked
    against the oracle."""
    import torch

    w, h = 1920, 1080
    frames_ = [pc.frames(oracle, w, h, d, seed=700 + d) for d in (1, 5)]
    tr = [torch.from_numpy(f[0]).cuda() for f in frames_]
    batches = [hip_ctx.make_device_batch(t.unsqueeze(0), 4 * w, w, h, planes_are_rgb=True, colorspace=2) for t in tr]
    ref = []
    for i, (rgb, yuv) in enumerate(frames_):
        v = torch.empty(65536, dtype=torch.uint8, device="cuda")
        wv = torch.empty(256 * w * 4, dtype=torch.uint8, device="cuda")
        hs = torch.empty(1024, dtype=torch.int32, device="cuda")
        hip_ctx.three_scopes_batch_dev(batches[i], 0x07, 0x07, v, wv, hs)
        pc.assert_same(v.cpu().numpy().reshape(256, 256), oracle.vectorscope(yuv, w), "stress reference vectorscope")
        pc.assert_same(wv.cpu().numpy().reshape(256, w, 4), oracle.waveform(rgb, yuv, w, 0x07), "stress reference waveform")
        pc.assert_same(hs.cpu().numpy().view(np.uint32).reshape(256, 4), oracle.histogram_counts(rgb, yuv, w, 0x07), "stress reference histogram")
        ref.append((v, wv, hs))
    outs = [[(torch.empty(65536, dtype=torch.uint8, device="cuda"), torch.empty(256 * w * 4, dtype=torch.uint8, device="cuda"),
              torch.empty(1024, dtype=torch.int32, device="cuda")) for _ in range(2)] for _ in range(2)]
    for it in range(1000):
        for i in range(2):
            v, wv, hs = outs[i][it & 1]
            hip_ctx.three_scopes_batch_dev(batches[i], 0x07, 0x07, v, wv, hs, sync=False)
        if it % 100 == 99:
            hip_ctx.sync()
            for i in range(2):
                for got, want, what in zip(outs[i][it & 1], ref[i], ("vectorscope", "waveform", "histogram")):
                    assert torch.equal(got, want), f"dock stress: {what} of frame {i} differs after {2 * (it + 1)} launches"


@pytest.mark.parametrize("cs", [1, 2])
def test_dock_kernel_all_colours(hip_ctx, oracle, cs):
    """All 2^24 colours through the one-read kernel (K0 on texel pairs out of registers): two device-resident frames of
    4096 x 2048 (256 strips each: exactly one round), a shuffled order so that every row range and replica sees all kinds;
    the vectorscope, the RGB waveform and the RGB histogram of each against the oracle."""
    to_dev, to_host = _torch_dev()
    perm = np.random.default_rng(11 + cs).permutation(1 << 24).astype(np.uint32)
    v = perm | np.uint32(0xFF000000)
    for half in range(2):
        rgb = np.ascontiguousarray(v[half << 23:(half + 1) << 23]).view(np.uint8).reshape(2048, 4096 * 4)
        yuv = oracle.rgb_to_yuv(rgb, 4096, cs)
        hr, ar = to_dev(rgb.reshape(-1))
        b = hip_ctx._batch([ar], 4096 * 4, 4096, 2048, True, cs)
        hv, av = to_dev(np.full(65536, 0xEE, np.uint8))
        hw, aw = to_dev(np.full(256 * 4096 * 4, 0xEE, np.uint8))
        hh, ah = to_dev(np.full(4096, 0xEE, np.uint8))
        hip_ctx.three_scopes_batch_dev(b, 0x07, 0x07, av, aw, ah)
        pc.assert_same(to_host(hv).reshape(256, 256), oracle.vectorscope(yuv, 4096), f"dock vectorscope, all colours, cs{cs} half {half}")
        pc.assert_same(to_host(hw).reshape(256, 4096, 4), oracle.waveform(rgb, yuv, 4096, 0x07), f"dock waveform, all colours, half {half}")
        pc.assert_same(to_host(hh).view(np.uint32).reshape(256, 4), oracle.histogram_counts(rgb, yuv, 4096, 0x07), f"dock histogram, half {half}")


# ---------------------------------------------------------------------------------- full sizes ---


def test_config1_histogram_720p(hip_ctx, oracle):
    """BASELINE config 1: 256-bin RGB histogram over one 1280x720 frame (counts + level pass)."""
    pc.check_histogram_counts(hip_ctx, oracle, 1280, 720, 1, 0x07)
    pc.check_histogram_full(hip_ctx, oracle, 1280, 720, 1, 0x07, 0, 0, 0)
    pc.check_histogram_full(hip_ctx, oracle, 1280, 720, 2, 0x07, 1, 0, 0)


@pytest.mark.parametrize("dist", [1, 2, 3, 4, 5])
def test_config2_vectorscope_4k(hip_ctx, oracle, dist):
    pc.check_vectorscope(hip_ctx, oracle, 3840, 2160, dist)


def test_config2_vectorscope_4k_fused_and_padded(hip_ctx, oracle):
    pc.check_vectorscope(hip_ctx, oracle, 3840, 2160, 1, None, fuse=True, colorspace=2)
    pc.check_vectorscope(hip_ctx, oracle, 3840, 2160, 2, None, fuse=True, colorspace=1)
    pc.check_vectorscope(hip_ctx, oracle, 3840, 2160, 1, 3840 * 4 + 256)


@pytest.mark.parametrize("comp", [0x20, 0x04, 0x02, 0x01, 0x07, 0x50, 0x70])
def test_config3_waveform_4k(hip_ctx, oracle, comp):
    pc.check_waveform(hip_ctx, oracle, 3840, 2160, 1 if comp != 0x07 else 5, comp)


def test_config3_waveform_4k_concentrated_and_flat(hip_ctx, oracle):
    pc.check_waveform(hip_ctx, oracle, 3840, 2160, 2, 0x07)
    pc.check_waveform(hip_ctx, oracle, 3840, 2160, 3, 0x70)


def test_histogram_4k(hip_ctx, oracle):
    for dist in (1, 2, 3, 5):
        pc.check_histogram_counts(hip_ctx, oracle, 3840, 2160, dist, 0x07)
    pc.check_histogram_counts(hip_ctx, oracle, 3840, 2160, 1, 0x70)


def test_config4_vectorscope_8k_row_tiles(hip_ctx, oracle):
    """BASELINE config 4 on one GPU: a 7680x4320 frame as 8 row tiles of 540 rows; the tiles'
    clamped tables, packed two-per-word, summed (what the RCCL all-reduce does) and clamped once
    equal the whole-frame table.  Also the whole frame in one call."""
    import torch

    w, h, tiles = 7680, 4320, 8
    rgb, yuv = pc.frames(oracle, w, h, 2)
    want = oracle.vectorscope(yuv, w)
    got, st = hip_ctx.vectorscope(yuv=yuv, width=w)
    assert st == 0
    pc.assert_same(got, want, "8K vectorscope, one call")

    acc = torch.zeros(32768, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for t in range(tiles):
        tile = yuv[t * h // tiles:(t + 1) * h // tiles]
        part, st = hip_ctx.vectorscope(yuv=tile, width=w)
        assert st == 0
        d = torch.from_numpy(part.reshape(-1)).cuda()
        packed = torch.empty(32768, dtype=torch.int32, device="cuda")
        hip_ctx.pack16_dev(d, packed, 32768)
        hip_ctx.sync()
        acc += packed
    out = torch.empty(65536, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()  # the sums ran on torch's stream, the context has its own
    hip_ctx.unpack16_clamp_dev(acc, out, 32768)
    hip_ctx.sync()
    pc.assert_same(out.cpu().numpy().reshape(256, 256), want, "8K vectorscope, 8 row tiles + packed sum + clamp")


def test_batched_device_4k(hip_ctx, oracle):
    """Eight distinct 4K frames resident in HBM, one batched launch per scope."""
    import torch

    w, h, n = 3840, 2160, 8
    rgbs, yuvs = [], []
    for i in range(n):
        r, y = pc.frames(oracle, w, h, 1 + i % 5, seed=31 + i)
        rgbs.append(r)
        yuvs.append(y)
    dy = torch.from_numpy(np.stack(yuvs)).cuda()
    dr = torch.from_numpy(np.stack(rgbs)).cuda()
    by = hip_ctx.make_device_batch(dy, 4 * w, w, h, planes_are_rgb=False)
    br = hip_ctx.make_device_batch(dr, 4 * w, w, h, planes_are_rgb=True)

    out = torch.empty((n, 256, 256), dtype=torch.uint8, device="cuda")
    hip_ctx.vectorscope_batch_dev(by, out)
    got = out.cpu().numpy()
    for i in range(n):
        pc.assert_same(got[i], oracle.vectorscope(yuvs[i], w), f"batched 4K vectorscope frame {i}")

    out = torch.empty((n, 256, 4), dtype=torch.int32, device="cuda")
    hip_ctx.histogram_counts_batch_dev(br, 0x07, out)
    got = out.cpu().numpy().view(np.uint32)
    for i in range(n):
        pc.assert_same(got[i], oracle.histogram_counts(rgbs[i], None, w, 0x07), f"batched 4K histogram frame {i}")

    out = torch.empty((n, 256, w, 4), dtype=torch.uint8, device="cuda")
    hip_ctx.waveform_batch_dev(br, 0x07, out)
    got = out.cpu().numpy()
    for i in range(0, n, 3):
        pc.assert_same(got[i], oracle.waveform(rgbs[i], None, w, 0x07), f"batched 4K waveform frame {i}")


# ---------------------------------------------------------------------------------- properties ---


def test_properties_full_size(hip_ctx, oracle):
    """Size-independent checks at 4K that do not need the oracle's tables."""
    w, h = 3840, 2160
    rgb, yuv = pc.frames(oracle, w, h, 5)
    counts, st = hip_ctx.histogram_counts(rgb=rgb, yuv=yuv, width=w, components=0x07)
    visible = int((rgb[:, : 4 * w].reshape(h, w, 4)[..., 3] != 0).sum())
    assert st == 0 and [int(counts[:, j].sum()) for j in range(4)] == [visible, visible, visible, 0]
    # idempotence / determinism: same call twice -> same table
    a, _ = hip_ctx.vectorscope(yuv=yuv, width=w)
    b, _ = hip_ctx.vectorscope(yuv=yuv, width=w)
    assert np.array_equal(a, b)
    # a vectorscope never exceeds 255 and is monotone under adding pixels: top half <= whole
    top, _ = hip_ctx.vectorscope(yuv=yuv[: h // 2], width=w)
    assert (top <= a).all()
    # waveform columns: luma table of a flat frame has exactly one non-zero row, all 255
    flat_rgb, flat_yuv = pc.frames(oracle, w, h, 3)
    wv, _ = hip_ctx.waveform(rgb=flat_rgb, yuv=flat_yuv, width=w, components=0x20)
    rows = np.nonzero(wv[..., 1].any(axis=1))[0]
    assert list(rows) == [255 - 204] and (wv[255 - 204, :, 1] == 255).all() and not wv[..., (0, 2, 3)].any()


def test_surface_fanout_uploads_each_plane_once(hip_ctx, oracle):
    """The dock case: three scopes on one staged surface, each plane crossing PCIe once."""
    pc.check_surface_fanout(hip_ctx, oracle, 1920, 1080)
    pc.check_surface_fanout(hip_ctx, oracle, 1279, 301, 1279 * 4 + 20)


def test_context_is_safe_to_share_between_threads(hip_ctx, oracle):
    import threading

    w, h = 640, 360
    rgb, yuv = pc.frames(oracle, w, h, 1)
    want_v, want_w = oracle.vectorscope(yuv, w), oracle.waveform(rgb, yuv, w, 0x07)
    errors = []

    def worker(kind):
        try:
            for _ in range(20):
                if kind == 0:
                    got, _ = hip_ctx.vectorscope(yuv=yuv, width=w)
                    assert np.array_equal(got, want_v)
                elif kind == 1:
                    got, _ = hip_ctx.waveform(rgb=rgb, yuv=yuv, width=w, components=0x07)
                    assert np.array_equal(got, want_w)
                else:
                    hip_ctx.surface_begin()
                    a, _ = hip_ctx.vectorscope(rgb=rgb, yuv=yuv, width=w)
                    b, _ = hip_ctx.waveform(rgb=rgb, yuv=yuv, width=w, components=0x07)
                    assert hip_ctx.surface_end() == (2, 0)
                    assert np.array_equal(a, want_v) and np.array_equal(b, want_w)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(k,)) for k in (0, 1, 2, 0, 2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_frame_pipeline_streams(hip_ctx, oracle):
    """BASELINE config 5 in miniature and at full size: frames in flight over upload/compute/download streams."""
    pc.check_pipeline(hip_ctx, oracle, 640, 360, n_frames=9, depth=3)
    pc.check_pipeline(hip_ctx, oracle, 3840, 2160, n_frames=5, depth=3)


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_frame_pipeline_tables_stay_valid(hip_ctx, oracle, depth):
    pc.check_pipeline_tables_stay_valid(hip_ctx, oracle, 1280, 720, depth)


@pytest.mark.parametrize("region", [(2, 0, 0, 0, 0), (1, 100, 50, 333, 111), (3, 7, 5, 200, 100)])
def test_frame_pipeline_scaled_and_cropped(hip_ctx, oracle, region):
    """target_scale (the reference's default is 2) and the ROI rectangle on the device."""
    pc.check_pipeline(hip_ctx, oracle, 1280, 720, n_frames=5, depth=3, region=region)


def test_frame_pipeline_4k_default_scale(hip_ctx, oracle):
    pc.check_pipeline(hip_ctx, oracle, 3840, 2160, n_frames=4, depth=3, region=(2, 0, 0, 0, 0))


@pytest.mark.parametrize("w,h,ls", [(640, 360, None), (1279, 301, 1279 * 4 + 20), (3840, 2160, None)])
def test_stage_region(hip_ctx, oracle, w, h, ls):
    pc.check_stage_region(hip_ctx, oracle, w, h, ls)


def test_stage_region_device_planes(hip_ctx, oracle):
    import torch

    w, h = 1920, 1080
    plane = oracle.fill_frame(w, h, 1, 77)
    src = torch.from_numpy(plane).cuda()
    for region in ((2, 0, 0, 0, 0), (5, 3, 2, 100, 60), (1, 11, 7, 640, 360)):
        want = oracle.stage_region(plane, w, *region)
        dst = torch.full((want.shape[0], want.shape[1] + 16), 0xA5, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()  # the context runs on its own stream: the fill above must have landed first
        hip_ctx.stage_region_dev(src, w * 4, w, h, region, dst, dst.shape[1])
        got = dst.cpu().numpy()
        pc.assert_same(got[:, :want.shape[1]], want, f"device stage_region {region}")
        assert (got[:, want.shape[1]:] == 0xA5).all()


def test_display_post_passes(hip_ctx, oracle):
    """tables -> scope pictures on the GPU, every display technique."""
    pc.check_display(hip_ctx, oracle, 70, 40)
    pc.check_display(hip_ctx, oracle, 1280, 72)


def test_graticule_drawn_over_the_picture(hip_ctx):
    """SURVEY.md 8f rank 4 on the GPU: label quads and line lists of the reference's graticule geometry rasterised into
    the scope pictures (pinned rule) against the oracle's rasteriser, every display mode, zoomed vectorscope included."""
    pc.check_graticule(hip_ctx)


def test_graticule_device_picture(hip_ctx, oracle):
    """The picture stays in HBM between the display pass and the graticule pass."""
    import ctypes as C

    import torch

    import graticule_oracle as go

    rgb, yuv = pc.frames(oracle, 640, 360, 2)
    vss = oracle.vectorscope(yuv, 640)
    want = go.rasterise(oracle.display_vectorscope(vss, 25, 1, 2), *go.vectorscope(2, 1 | 256, 0x0054FF), 0x80FFBF00,
                        hip_ctx.graticule_label_strip())
    tab = torch.from_numpy(vss.copy()).cuda()
    pic = torch.empty((256, 256, 4), dtype=torch.uint8, device="cuda")
    dev = 0x1 | 0x2
    st = hip_ctx.L.scope_display_vectorscope(hip_ctx.h, C.c_void_p(tab.data_ptr()), 25, 1, 2, dev, C.c_void_p(pic.data_ptr()))
    assert st == 0
    labels, lines = go.vectorscope(2, 1 | 256, 0x0054FF)
    labels, lines = np.ascontiguousarray(labels, np.float32), np.ascontiguousarray(lines, np.float32)
    st = hip_ctx.L.scope_display_graticule(hip_ctx.h, C.c_void_p(pic.data_ptr()), 256, 256, labels.ctypes.data_as(C.c_void_p), 6,
                                          lines.ctypes.data_as(C.c_void_p), lines.shape[0], 0x80FFBF00, dev)
    assert st == 0
    pc.assert_same(pic.cpu().numpy(), want, "display + graticule on a device picture")


def test_table_all_reduce_side_stream_buckets(hip_ctx, oracle):
    """The reducer bench.py uses for N > 1, on a one-rank RCCL group: tables written straight into the wire buffer,
    several steps per collective, collective + unpack on a side stream, a partly filled bucket flushed by drain()."""
    import torch
    import torch.distributed as dist

    from obs_color_monitor_amd.row_tiles import TableAllReduce

    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29591")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        w, h, n = 320, 180, 2
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            hip_ctx.set_stream(stream.cuda_stream)
            red = TableAllReduce(hip_ctx, n * 65536, "cuda", "sat_u8", bucket=3)
            red32 = TableAllReduce(hip_ctx, n * 4096, "cuda", "sum_u32", bucket=3)
            wants, wants32, keep = [], [], []
            cnt = torch.empty((n, 1024), dtype=torch.int32, device="cuda")
            for step in range(7):
                planes_y, planes_r = [], []
                for f in range(n):
                    rgb = oracle.fill_frame(w, h, 1 + (step + f) % 5, 300 + 10 * step + f)
                    planes_r.append(rgb)
                    planes_y.append(oracle.rgb_to_yuv(rgb, w, 2))
                wants.append(np.stack([oracle.vectorscope(y, w) for y in planes_y]))
                wants32.append(np.stack([oracle.histogram_counts(r, None, w, 0x07) for r in planes_r]))
                ty = torch.from_numpy(np.stack(planes_y)).cuda()
                tr = torch.from_numpy(np.stack(planes_r)).cuda()
                keep += [ty, tr]
                by = hip_ctx.make_device_batch(ty, 4 * w, w, h, planes_are_rgb=False)
                br = hip_ctx.make_device_batch(tr, 4 * w, w, h, planes_are_rgb=True, colorspace=2)
                hip_ctx.vectorscope_batch_dev(by, red.wire_buffer(), sync=False, packed16=True)
                red.submit_packed()
                hip_ctx.histogram_counts_batch_dev(br, 0x07, cnt, sync=False)
                red32.submit(cnt.view(torch.uint8))
            last = red.drain()
            last32 = red32.drain()
            torch.cuda.current_stream().synchronize()
            assert np.array_equal(last.cpu().numpy().reshape(n, 256, 256), wants[6])
            assert np.array_equal(last32.cpu().numpy().view(np.uint32).reshape(n, 256, 4), wants32[6])
            # the buckets still hold the other steps: 3,4,5 in the second bucket; 1,2 behind step 6 in the first
            for step, (s, k) in {3: (1, 0), 4: (1, 1), 5: (1, 2), 1: (0, 1), 2: (0, 2)}.items():
                got = red._final_slice(s, k).cpu().numpy().reshape(n, 256, 256)
                assert np.array_equal(got, wants[step]), f"step {step}"
                got32 = red32._final_slice(s, k).cpu().numpy().view(np.uint32).reshape(n, 256, 4)
                assert np.array_equal(got32, wants32[step]), f"histogram step {step}"
        hip_ctx.set_stream(None)
    finally:
        hip_ctx.set_stream(None)
        if created:
            dist.destroy_process_group()


def test_table_allreduce_through_the_c_abi(hip_ctx, oracle):
    """scope_table_allreduce: RCCL called from the C-ABI with a communicator made by scope_rccl_comm_create (a one-rank
    clique on this 1-GPU box), on the context's stream between the packed-output vectorscope and the unpack/clamp
    pass -- the data path a C host uses for row tiles, no torch.distributed in it."""
    import torch

    w, h, n = 640, 360, 3
    uid = hip_ctx.rccl_unique_id()
    assert len(uid) == 128 and any(uid)
    comm = hip_ctx.rccl_comm_create(uid, 1, 0)
    try:
        yuvs = [oracle.rgb_to_yuv(oracle.fill_frame(w, h, 1 + i, 70 + i), w, 2) for i in range(n)]
        ty = torch.from_numpy(np.stack(yuvs)).cuda()
        batch = hip_ctx.make_device_batch(ty, 4 * w, w, h, planes_are_rgb=False)
        wire = torch.empty((n, 32768), dtype=torch.int32, device="cuda")
        out = torch.empty((n, 65536), dtype=torch.uint8, device="cuda")
        for _ in range(3):
            hip_ctx.vectorscope_batch_dev(batch, wire, sync=False, packed16=True)
            hip_ctx.table_allreduce(comm, wire, n * 32768)
            hip_ctx.unpack16_clamp_dev(wire, out, n * 32768)
            hip_ctx.sync()
            for f in range(n):
                pc.assert_same(out[f].cpu().numpy().reshape(256, 256), oracle.vectorscope(yuvs[f], w), f"frame {f}")
        cnt = torch.empty((n, 1024), dtype=torch.int32, device="cuda")
        tr = torch.from_numpy(np.stack([oracle.fill_frame(w, h, 5, 3 + i) for i in range(n)])).cuda()
        br = hip_ctx.make_device_batch(tr, 4 * w, w, h, planes_are_rgb=True)
        hip_ctx.histogram_counts_batch_dev(br, 0x07, cnt, sync=False)
        hip_ctx.table_allreduce(comm, cnt, n * 1024)
        hip_ctx.sync()
        for f in range(n):
            pc.assert_same(cnt[f].cpu().numpy().view(np.uint32).reshape(256, 4),
                           oracle.histogram_counts(tr[f].cpu().numpy(), None, w, 0x07), f"histogram frame {f}")
    finally:
        hip_ctx.rccl_comm_destroy(comm)


def _run_bench(args, env_extra=None, timeout=900):
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=timeout, cwd=root)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"rc={out.returncode}\n{out.stdout[-2000:]}\n{out.stderr[-5000:]}"
    return out.returncode, json.loads(lines[0]), out.stderr


def test_bench_two_ranks_on_one_gpu_table_sum_overlapped_and_serial(hip_ctx):
    """`python bench.py --gpus 2` as the driver types it (no WORLD_SIZE): two self-launched ranks that SHARE this box's one
    GPU, the real kernels, the C-ABI's scope_table_allreduce bound to tools/libshm_rccl_hip.so (a stream-ordered,
    asynchronous all-reduce between processes; RCCL itself refuses two ranks on one device).  What it checks is this
    repo's side of the N > 1 path: the launcher, the gloo control plane, the two-stream overlap's event order (a table
    summed under the next step's accumulation is compared with the oracle, and in the config-4 leg the overlapped and
    the serial path must leave the same table), the row tiles of one shared 7680x4320 batch."""
    from conftest import ROOT, locked_make

    locked_make(os.path.join(ROOT, "tools"), "libshm_rccl_hip.so")
    lib = os.path.join(ROOT, "tools", "libshm_rccl_hip.so")
    rc, d, err = _run_bench(["--gpus", "2", "--oversubscribe", "--frames", "8", "--pool", "8", "--steps", "20", "--warmup", "3",
                             "--min-seconds", "0.3", "--settle-seconds", "0.1", "--placement-candidates", "1", "--no-cpu",
                             "--config4-frames", "4", "--launch-timeout", "400"], {"SCOPE_RCCL_LIBRARY": lib})
    assert rc == 0, err[-4000:]
    assert d["n_gpus"] == 2 and d["parity"].startswith("bit-exact"), d["parity"]
    assert d["config"]["gpu_shared_by_ranks"] is True and d["config"]["rccl"]["ranks_in_communicator"] == 2
    assert d["config"]["reduce"].startswith("C-ABI") and "second hipStream" in d["config"]["reduce"]
    assert len(d["launcher"]["attempts"]) == 1 and d["launcher"]["attempts"][0]["failure"] is None
    c4 = d["config4"]
    assert "error" not in c4, c4
    assert c4["parity"] is True and c4["rccl_ranks_in_communicator"] == 2 and c4["rows_of_this_rank"] == [0, 2160]
    assert c4["ms_per_step_reduce_overlapped_second_stream"] > 0
    # and the same summed on the accumulation's own stream, headline included
    rc, d2, err = _run_bench(["--gpus", "2", "--oversubscribe", "--serial-reduce", "--frames", "8", "--pool", "8", "--steps", "20",
                              "--warmup", "3", "--min-seconds", "0.2", "--settle-seconds", "0.1", "--placement-candidates", "1",
                              "--no-cpu", "--no-extras", "--launch-timeout", "400"], {"SCOPE_RCCL_LIBRARY": lib})
    assert rc == 0 and d2["parity"].startswith("bit-exact") and "same stream" in d2["config"]["reduce"], err[-3000:]


def test_bench_one_gpu_line_has_config4_and_rccl_rank_count(hip_ctx):
    """The driver's N = 1 command: the line carries the config-4 strong leg's base point (one rank owns all 4320 rows, the
    sum goes through RCCL itself with a one-rank communicator) and says how many ranks RCCL saw."""
    rc, d, err = _run_bench(["--gpus", "1", "--steps", "5", "--warmup", "2", "--min-seconds", "0.2", "--no-cpu", "--no-extras", "--strong",
                             "--workload", "vectorscope", "--frames", "8", "--pool", "8", "--placement-candidates", "1", "--config4-frames", "4"])
    assert rc == 0, err[-4000:]
    c4 = d["config4"]
    assert "error" not in c4, c4
    assert c4["parity"] is True and c4["rccl_ranks_in_communicator"] == 1 and c4["rows_of_this_rank"] == [0, 4320]
    assert d["scaling"] == "strong" and d["value"] == c4["value"]


def test_vectorscope_partials_meet_inside_the_launch(hip_ctx, oracle):
    """vss_write_out's fold (round 6: the frame's last-arriving workgroup adds the other slices' write-through partials to its
    own): 2 / 3 / 4 slices per frame through every producer (staged, fused, packed wire format, the three-scope kernel), several
    launches in a row, against the oracle and against the reduction kernel's path; 4K frames so that the workgroups of a frame
    finish at different moments on different XCDs."""
    to_dev, to_host = _torch_dev()

    def dev(*a, **k):
        pc.check_vectorscope_fold(hip_ctx, oracle, to_dev, to_host, *a, **k)

    dev(3840, 2160, [1, 3, {"detail": 0.03}, 5], launches=3)
    dev(1924, 1083, [2, 1], launches=2, linesize=1924 * 4 + 64)
    dev(640, 360, [1, 2, 3, 4, 5, 1, 2, 3], launches=4)


def test_vectorscope_fold_stress(hip_ctx, oracle):
    """The hand-off under UNEVEN load with every table word checked, many times: sixteen different 1080p frames of very different
    cost (noise, flat, screen-like) per launch, four slices per frame, 300 launches back to back into fresh tables; a stale or a
    missing partial shows as a wrong count."""
    import numpy as np
    import torch

    w, h, n = 1920, 1080, 16
    kinds = [1, 3, {"detail": 0.03}, 2, 5, {"detail": 0.3}, 3, 1] * 2
    pairs = []
    for i, d in enumerate(kinds):
        if isinstance(d, dict):
            rgb = pc.screen_frame(w, h, 900 + i, **d)
            pairs.append((rgb, oracle.rgb_to_yuv(rgb, w, 2)))
        else:
            pairs.append(pc.frames(oracle, w, h, d, seed=800 + i))
    want = torch.from_numpy(np.stack([oracle.vectorscope(y, w).reshape(-1) for _, y in pairs])).cuda()
    t_yuv = torch.from_numpy(np.stack([y for _, y in pairs])).cuda()
    t_rgb = torch.from_numpy(np.stack([r for r, _ in pairs])).cuda()
    b_yuv = hip_ctx.make_device_batch(t_yuv, 4 * w, w, h, planes_are_rgb=False)
    b_rgb = hip_ctx.make_device_batch(t_rgb, 4 * w, w, h, planes_are_rgb=True, colorspace=2)
    with pc.options(hip_ctx, **{"slices": 4}):
        outs = [torch.full((n, 65536), 0xEE, dtype=torch.uint8, device="cuda") for _ in range(8)]
        bad = 0
        for it in range(300):
            o = outs[it % 8]
            o.fill_(0xEE)
            hip_ctx.vectorscope_batch_dev(b_rgb if it % 3 == 2 else b_yuv, o, sync=False)
            if it % 8 == 7:
                hip_ctx.sync()
                bad += sum(int((x != want).sum()) for x in outs)
        hip_ctx.sync()
        assert bad == 0, f"{bad} wrong table cells over 300 folded launches"


def test_frame_pool_alloc_places_a_pool(hip_ctx, oracle):
    """scope_frame_pool_alloc on the GPU: four candidates of 256 MB each are read and timed, the fastest is kept (its rate
    comes first and is the maximum), the pool is ordinary device memory that torch can view and a scope can read, and
    freeing it gives the memory back."""
    import torch

    n_bytes = 256 << 20
    free0 = torch.cuda.mem_get_info()[0]
    pool, rates = hip_ctx.frame_pool_alloc(n_bytes, candidates=4)
    assert len(rates) == 4 and rates[0] == max(rates) and min(rates) > 500.0, rates  # GB/s of a plain streaming read
    assert torch.cuda.mem_get_info()[0] <= free0 - n_bytes + (64 << 20)  # one allocation is held, three went back
    w, h = 1920, 1080
    t = pool.tensor(torch, (4, h, 4 * w))
    frames_ = [pc.frames(oracle, w, h, d, seed=40 + d) for d in (1, 2, 5, 3)]
    for i, (rgb, yuv) in enumerate(frames_):
        t[i].copy_(torch.from_numpy(yuv))
    assert int(t.data_ptr()) == pool.ptr
    b = hip_ctx.make_device_batch(t, 4 * w, w, h, planes_are_rgb=False)
    out = torch.empty((4, 65536), dtype=torch.uint8, device="cuda")
    hip_ctx.vectorscope_batch_dev(b, out)
    for i, (rgb, yuv) in enumerate(frames_):
        pc.assert_same(out[i].cpu().numpy().reshape(256, 256), oracle.vectorscope(yuv, w), f"vectorscope of pool frame {i}")
    del t
    pool.free()
    assert torch.cuda.mem_get_info()[0] >= free0 - (64 << 20)


def test_vectorscope_side_reduce(hip_ctx, oracle):
    """SCOPE_SIDE_REDUCE on the GPU: trains of one-frame 4K calls (160 slices each) and of 1080p calls with the fused K0,
    the reductions on the second stream under the next call's accumulation; 300 calls back to back with no host
    synchronisation in between."""
    to_dev, to_host = _torch_dev()
    pc.check_vectorscope_side_reduce(hip_ctx, oracle, to_dev, to_host, 3840, 2160, [1, 5, 2, 3], calls=40)
    pc.check_vectorscope_side_reduce(hip_ctx, oracle, to_dev, to_host, 1920, 1080, [5, 1], calls=300, fuse=True)


def test_pipeline_without_histogram(hip_ctx, oracle):
    pc.check_pipeline_without_histogram(hip_ctx, oracle, 1920, 1080, depth=3)


@pytest.mark.parametrize("forced", [0, 1])
def test_plain_vectorscope_screen_content(hip_ctx, oracle, forced):
    """The staged-plane vectorscope on screen-like content: the run-length form of its absorb mode (mixed windows), with the
    batch's unroll (the form that carries it), the default, and one workgroup per frame."""
    pc.check_plain_screen_content(hip_ctx, oracle, [(3840, 2160, None)], forced)
    for slices, unroll in ((0, 0), (1, 4), (0, 4), (7, 1)):
        with pc.options(hip_ctx, **{"slices": slices, "vss.unroll": unroll}):
            pc.check_plain_screen_content(hip_ctx, oracle, [(640, 360, None), (509, 130, 2048)], forced)
