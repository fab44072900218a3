#!/usr/bin/env python3
"""Kernel-variant sweep on one MI355X (tuning aid, not the bench contract).

Every line times the accumulation kernel alone with the C-ABI's HIP-event bracket
(scope_ctx_profile / scope_ctx_kernel_ms) over distinct HBM-resident 3840x2160 frames and reports
ms per launch, algorithmic GB/s (SURVEY.md 8(d) bytes) and Gpix/s.  Output: one JSON object per line
on stdout-file, human table on stderr.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import obs_color_monitor_amd as ocm  # noqa: E402

W, H = 3840, 2160


def make_frames(ctx, n, dist, seed=1):
    from obs_color_monitor_amd.synth import make_frames as mk

    return mk(torch, ctx, n, W, H, dist, seed)


def timed(ctx, fn, reps=20, warm=5):
    for _ in range(warm):
        fn()
    ctx.sync()
    ctx.profile(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    wall = (time.perf_counter() - t0) / reps * 1e3
    n, ms = ctx.kernel_ms()
    ctx.profile(False)
    return ms / max(n, 1), wall, n // reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/sweep.jsonl")
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--sections", default="ABCDEFG", help="which lettered sections to run")
    args = ap.parse_args()
    sec = args.sections.upper()
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    out = open(args.out, "w")

    ctx = ocm.ScopeContext(0)
    stream = torch.cuda.Stream()  # explicit stream shared by torch and the C-ABI (NULL would mean ctx's own)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    info = ctx.device_info()
    print(f"# {info}", file=sys.stderr)
    n = args.frames

    def emit(rec):
        out.write(json.dumps(rec) + "\n")
        out.flush()
        print("  ".join(f"{k}={v}" for k, v in rec.items()), file=sys.stderr, flush=True)

    def report(tag, kern_ms, wall_ms, launches, nframes, table_bytes, **kw):
        algo = nframes * (4 * W * H + table_bytes) / max(launches, 1) if launches else 0
        emit(dict(tag=tag, **kw, frames=nframes, launches_per_call=launches, kernel_ms=round(kern_ms, 4),
                  wall_ms_per_call=round(wall_ms, 4),
                  gbps=round(algo / (kern_ms * 1e-3) / 1e9, 1) if kern_ms else None,
                  gpix_wall=round(nframes * W * H / (wall_ms * 1e-3) / 1e9, 1)))

    # ---- reference points: what the chip does on a plain stream of the same bytes ------------------
    a = torch.empty(n * H * W, dtype=torch.int32, device="cuda")
    b = torch.empty_like(a)
    t_end = time.perf_counter() + 0.7
    while time.perf_counter() < t_end:
        for _ in range(16):
            b.copy_(a)
        torch.cuda.synchronize()
    for name, fn, bytes_moved in (("d2d_copy(read+write)", lambda: b.copy_(a), 2 * a.numel() * 4),
                                  ("torch_sum(read)", lambda: a.sum(), a.numel() * 4)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        emit(dict(tag="ref", what=name, ms=round(ms, 4), gbps=round(bytes_moved / (ms * 1e-3) / 1e9, 1)))
    del a, b

    dists = ["uniform", "concentrated", "flat", "ramp", "natural", "screen"] if not args.quick else ["uniform", "natural", "screen"]
    data = {d: make_frames(ctx, n, d) for d in dists}
    tab = torch.empty((n, 65536), dtype=torch.uint8, device="cuda")
    # let the clocks leave idle before anything is timed
    warm_batch = ctx.make_device_batch(data[dists[0]][1], 4 * W, W, H, planes_are_rgb=False)
    t_end = time.perf_counter() + 0.7
    while time.perf_counter() < t_end:
        for _ in range(16):
            ctx.vectorscope_batch_dev(warm_batch, tab, sync=False)
        ctx.sync()

    # ---- A: vectorscope, batch launch: merge x lane groups x hash x unroll x nt x distribution ---------
    if "A" in sec:
        defaults = {"vss.merge": 3, "vss.unroll": 4, "vss.hash": 0, "vss.groups": 3, "nt": 1}

        def vss_run(batch, d, **opt):
            for k, v in {**defaults, **opt}.items():
                ctx.set_option(k, v)
            k, wall, l = timed(ctx, lambda: ctx.vectorscope_batch_dev(batch, tab, sync=False))
            report("vss_batch", k, wall, l, n, 65536, dist=d, **{kk.replace("vss.", ""): vv for kk, vv in {**defaults, **opt}.items()})

        for d in dists:
            rgb, yuv = data[d]
            batch = ctx.make_device_batch(yuv, 4 * W, W, H, planes_are_rgb=False)
            for merge in (0, 2, 3):
                for groups in (0, 2, 3):
                    for hash_ in (0, 1):
                        for nt in (0, 1):
                            vss_run(batch, d, **{"vss.merge": merge, "vss.groups": groups, "vss.hash": hash_, "nt": nt})
            for unroll in (1, 2):
                vss_run(batch, d, **{"vss.unroll": unroll})
                vss_run(batch, d, **{"vss.unroll": unroll, "vss.hash": 1, "vss.groups": 2})
            vss_run(batch, d, **{"vss.merge": 1})
        for k, v in defaults.items():
            ctx.set_option(k, v)

    # ---- B: how many workgroups share a frame (units per launch = frames * slices) ------------------
    if "B" in sec:
        rgb, yuv = data["uniform"]
        batch = ctx.make_device_batch(yuv, 4 * W, W, H, planes_are_rgb=False)
        for slices in (1, 2, 4, 8, 16, 32, 64):
            ctx.set_option("slices", slices)
            k, wall, l = timed(ctx, lambda: ctx.vectorscope_batch_dev(batch, tab, sync=False))
            report("vss_slices", k, wall, l, n, 65536, dist="uniform", slices=slices)
        ctx.set_option("slices", 0)

    # ---- C: one frame per call (the reference's per-callback situation), frames rotate ---------------
    if "C" in sec:
        rgb, yuv = data["uniform"]
        singles = [ctx.make_device_batch(yuv[i:i + 1], 4 * W, W, H, planes_are_rgb=False) for i in range(n)]
        for slices in (256, 128, 64, 32):
            ctx.set_option("slices", slices)

            def run_all():
                for i in range(n):
                    ctx.vectorscope_batch_dev(singles[i], tab[i], sync=False)
            k, wall, l = timed(ctx, run_all, reps=3, warm=1)
            report("vss_single", k, wall, l, n, 65536, dist="uniform", slices=slices)
        ctx.set_option("slices", 0)

    # ---- D: fused K0 (scopes reading RGB planes; the conversion runs inside the accumulation kernel) --------
    if "D" in sec:
        cnt_d = torch.empty((n, 1024), dtype=torch.int32, device="cuda")
        wtab_d = torch.empty((n, 256 * W * 4), dtype=torch.uint8, device="cuda")
        for d in dists:
            rgb, yuv = data[d]
            for cs in (2, 1):
                batch_rgb = ctx.make_device_batch(rgb, 4 * W, W, H, planes_are_rgb=True, colorspace=cs)
                k, wall, l = timed(ctx, lambda: ctx.vectorscope_batch_dev(batch_rgb, tab, sync=False))
                report("vss_fused_k0", k, wall, l, n, 65536, dist=d, colorspace=cs)
                for comp in (0x70, 0x20):
                    k, wall, l = timed(ctx, lambda: ctx.histogram_counts_batch_dev(batch_rgb, comp, cnt_d, sync=False))
                    report("his_fused_k0", k, wall, l, n, 4096, dist=d, colorspace=cs, comp=hex(comp))
                    k, wall, l = timed(ctx, lambda: ctx.waveform_batch_dev(batch_rgb, comp, wtab_d, sync=False), reps=10)
                    report("wvs_fused_k0", k, wall, l, n, 256 * W * 4, dist=d, colorspace=cs, comp=hex(comp))
        del cnt_d, wtab_d

    # ---- E: histogram ----------------------------------------------------------------------------------
    if "E" in sec:
        cnt = torch.empty((n, 1024), dtype=torch.int32, device="cuda")
        for d in dists:
            rgb, yuv = data[d]
            batch_rgb = ctx.make_device_batch(rgb, 4 * W, W, H, planes_are_rgb=True, colorspace=2)
            for unroll in (1, 2, 4):
                for nt in (0, 1):
                    ctx.set_option("his.unroll", unroll)
                    ctx.set_option("nt", nt)
                    for comp in (0x07, 0x02):
                        k, wall, l = timed(ctx, lambda: ctx.histogram_counts_batch_dev(batch_rgb, comp, cnt, sync=False))
                        report("his_batch", k, wall, l, n, 4096, dist=d, unroll=unroll, nt=nt, comp=hex(comp))
        ctx.set_option("his.unroll", 1)
        ctx.set_option("nt", 1)

    # ---- F: waveform ------------------------------------------------------------------------------------
    if "F" in sec:
        wtab_all = torch.empty((n, 256 * W * 4), dtype=torch.uint8, device="cuda")
        for d in dists:
            rgb, yuv = data[d]
            for nw in sorted({min(n, 16), n}):
                wtab = wtab_all[:nw]
                batch_rgb = ctx.make_device_batch(rgb[:nw], 4 * W, W, H, planes_are_rgb=True, colorspace=2)
                batch_yuv = ctx.make_device_batch(yuv[:nw], 4 * W, W, H, planes_are_rgb=False)
                for wide, persist in ((0, 1), (1, 0), (1, 1)):
                    ctx.set_option("wvs.wide", wide)
                    ctx.set_option("wvs.persist", persist)
                    for comp, bt in ((0x07, batch_rgb), (0x05, batch_rgb), (0x04, batch_rgb), (0x20, batch_yuv)):
                        k, wall, l = timed(ctx, lambda: ctx.waveform_batch_dev(bt, comp, wtab, sync=False), reps=10)
                        report("wvs_batch", k, wall, l, nw, 256 * W * 4, dist=d, wide=wide, persist=persist, comp=hex(comp))
        ctx.set_option("wvs.wide", 1)
        ctx.set_option("wvs.persist", 1)
        wtab = wtab_all
        # single-frame waveform (row bands + saturating reduce)
        rgb, yuv = data["uniform"]
        one = ctx.make_device_batch(rgb[:1], 4 * W, W, H, planes_are_rgb=True, colorspace=2)
        k, wall, l = timed(ctx, lambda: ctx.waveform_batch_dev(one, 0x07, wtab[0], sync=False))
        report("wvs_single", k, wall, l, 1, 256 * W * 4, dist="uniform")

    # ---- G: the plugin's situation: ONE pageable host plane per synchronous call (PCIe-inclusive) -----------
    if "G" in sec:
        import numpy as np

        host_yuv = data["uniform"][1][0].cpu().numpy()
        host_rgb = data["uniform"][0][0].cpu().numpy()
        for name, fn in (("vectorscope", lambda: ctx.vectorscope(yuv=host_yuv, width=W)),
                         ("waveform_rgb", lambda: ctx.waveform(rgb=host_rgb, yuv=host_yuv, width=W, components=0x07)),
                         ("histogram_rgb", lambda: ctx.histogram(rgb=host_rgb, yuv=host_yuv, width=W, components=0x07))):
            for _ in range(3):
                fn()
            t0 = time.perf_counter()
            reps = 20
            for _ in range(reps):
                fn()
            ms = (time.perf_counter() - t0) / reps * 1e3
            emit(dict(tag="host_call", scope=name, ms_per_frame=round(ms, 3), fps=round(1e3 / ms, 1),
                      gpix=round(W * H / (ms * 1e-3) / 1e9, 2), note="pageable host plane in, table out, synchronous"))

        # the dock: three scopes on ONE staged surface (RGB + YUV planes), per frame
        def dock():
            ctx.vectorscope(rgb=host_rgb, yuv=host_yuv, width=W)
            ctx.waveform(rgb=host_rgb, yuv=host_yuv, width=W, components=0x07)
            ctx.histogram(rgb=host_rgb, yuv=host_yuv, width=W, components=0x07)

        for name, begin in (("three calls, three uploads", None), ("scope_surface_begin/end: two uploads", False),
                            ("scope_surface_begin(FUSE_K0): one upload", True)):
            def run():
                if begin is not None:
                    ctx.surface_begin(fuse_k0=begin)
                dock()
                if begin is not None:
                    ctx.surface_end()
            for _ in range(3):
                run()
            t0 = time.perf_counter()
            for _ in range(20):
                run()
            ms = (time.perf_counter() - t0) / 20 * 1e3
            emit(dict(tag="host_dock", how=name, ms_per_frame=round(ms, 3), fps=round(1e3 / ms, 1)))

    # ---- H: bytes in flight: loads per lane per batch (two batches in flight) for K1 and K3 -------------------
    if "H" in sec:
        cnt_h = torch.empty((n, 1024), dtype=torch.int32, device="cuda")
        for d in dists:
            rgb, yuv = data[d]
            batch = ctx.make_device_batch(yuv, 4 * W, W, H, planes_are_rgb=False)
            batch_rgb = ctx.make_device_batch(rgb, 4 * W, W, H, planes_are_rgb=True, colorspace=2)
            for unroll in (1, 2, 4):
                for merge in (0, 3):
                    for groups in (0, 2, 3):
                        for k, v in {"vss.merge": merge, "vss.unroll": unroll, "vss.hash": 0, "vss.groups": groups, "nt": 1}.items():
                            ctx.set_option(k, v)
                        k, wall, l = timed(ctx, lambda: ctx.vectorscope_batch_dev(batch, tab, sync=False))
                        report("vss_inflight", k, wall, l, n, 65536, dist=d, unroll=unroll, merge=merge, groups=groups)
                ctx.set_option("vss.merge", 3)
                ctx.set_option("vss.groups", 3)
                ctx.set_option("vss.fused_unroll", unroll)
                k, wall, l = timed(ctx, lambda: ctx.vectorscope_batch_dev(batch_rgb, tab, sync=False))
                report("vss_fused_inflight", k, wall, l, n, 65536, dist=d, unroll=unroll)
                ctx.set_option("his.unroll", unroll)
                for comp in (0x07, 0x02):
                    k, wall, l = timed(ctx, lambda: ctx.histogram_counts_batch_dev(batch_rgb, comp, cnt_h, sync=False))
                    report("his_inflight", k, wall, l, n, 4096, dist=d, unroll=unroll, comp=hex(comp))
        ctx.set_option("vss.unroll", 4)
        ctx.set_option("vss.fused_unroll", 0)
        ctx.set_option("his.unroll", 1)

    out.close()
    ctx.close()


if __name__ == "__main__":
    main()
