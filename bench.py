#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its config: Gpix/s of scope accumulation @3840x2160.

  python bench.py --gpus N --steps K --warmup W         (N > 1: launched under torch.distributed.run)

A *step* is one pass of the hot path over one batch of synthetic input: by default the 256x256
CbCr vectorscope scatter (BASELINE configs[1]) over `--frames` (64, the most one launch takes) distinct
3840x2160 BGRA8 frames that are already resident in HBM (64 x 33 MB = 2.1 GB, far beyond the 256 MB
Infinity Cache, so every step streams from HBM; a launch costs ~20 us on top of its streaming time,
which is why the batch is deep: 32 frames read 6-7 % slower, 128 and 256 -- two and four launches -- no faster).  Frames are uniform-random RGB pushed through the K0 kernel, i.e. the staged Y'CbCr
planes the reference's vss_draw_vectorscope reads.  Every step goes through the C-ABI of
libscope_hip.so (no CPU path exists in the product).

N > 1 ("weak"): every rank owns one 3840x2160 row tile of a frame N times as tall; the ranks'
clamped 256x256 tables are summed over RCCL (two bins per int32 word) and clamped once, overlapped
with the next step's accumulation.  value = all ranks' pixels / max-over-ranks time.

Rank 0 prints ONE JSON line (see README/DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
W4K, H4K = 3840, 2160


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle-seconds", type=float, default=0.5,
                    help="untimed run of the same step before the warm-up so the GPU clocks (DVFS) leave idle")
    ap.add_argument("--frames", type=int, default=0, help="distinct HBM-resident frames per step (default 64 at 4k, 16 at 8k)")
    ap.add_argument("--size", default="4k", choices=["4k", "8k"],
                    help="frame size: 3840x2160 (the metric's configuration) or 7680x4320 (BASELINE config 4's frame)")
    ap.add_argument("--workload", default="vectorscope", choices=["vectorscope", "vectorscope_rgb", "waveform", "histogram"])
    ap.add_argument("--components", type=lambda s: int(s, 0), default=0x07)
    ap.add_argument("--dist", default="uniform", choices=["uniform", "concentrated", "flat", "ramp", "natural", "screen"])
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="bound of the CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--force-reduce", action="store_true",
                    help="run the RCCL table reduction even with one rank (exercises the N > 1 code path on a 1-GPU box)")
    ap.add_argument("--option", action="append", default=[], help="key=value passed to scope_ctx_set_option")
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner through C
    # stdio when a process group exists), so everything else in this process goes to stderr: file descriptor 1 is
    # pointed at stderr for the run and the JSON line is written to the saved, real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    import obs_color_monitor_amd as ocm

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_reduce
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    ctx = ocm.ScopeContext(local_rank)  # raises without libscope_hip.so / a GPU: no fallback
    # one explicit (non-default) stream for torch AND the C-ABI: generation, kernels, RCCL and the HIP
    # events that time the kernels are all ordered on it (a NULL handle would mean "ctx's own stream")
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    for kv in args.option:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    info = ctx.device_info()

    # ---- synthetic input, resident in HBM ------------------------------------------------------
    w, h = (W4K, H4K) if args.size == "4k" else (2 * W4K, 2 * H4K)
    n = args.frames or (64 if args.size == "4k" else 16)
    from obs_color_monitor_amd.synth import make_frames

    rgb, yuv = make_frames(torch, ctx, n, w, h, args.dist, 0x9E3779B9 + rank)
    torch.cuda.synchronize()

    wl = args.workload
    if wl == "vectorscope":
        batch = ctx.make_device_batch(yuv, 4 * w, w, h, planes_are_rgb=False)
        table_bytes = 65536
    elif wl == "vectorscope_rgb":
        batch = ctx.make_device_batch(rgb, 4 * w, w, h, planes_are_rgb=True, colorspace=2)
        table_bytes = 65536
    elif wl == "waveform":
        batch = ctx.make_device_batch(rgb, 4 * w, w, h, planes_are_rgb=True, colorspace=2)
        table_bytes = 256 * w * 4
    else:
        batch = ctx.make_device_batch(rgb, 4 * w, w, h, planes_are_rgb=True, colorspace=2)
        table_bytes = 4096
    tables = [torch.empty((n, table_bytes), dtype=torch.uint8, device="cuda") for _ in range(2)]
    reducer = None
    if use_dist and wl != "waveform":  # the waveform shards by column strip: every rank's slab is final, no collective
        from obs_color_monitor_amd.row_tiles import TableAllReduce

        # one collective per 8 steps: a collective per step costs more host time than a step takes on the GPU
        reducer = TableAllReduce(ctx, n * table_bytes, "cuda", "sum_u32" if wl == "histogram" else "sat_u8", bucket=8)
    result_tables = [None]

    def step(i):
        buf = i & 1
        if wl in ("vectorscope", "vectorscope_rgb") and reducer is not None:
            # tables go straight into the reducer's wire buffer in its packed format: no packing pass
            ctx.vectorscope_batch_dev(batch, reducer.wire_buffer(), sync=False, packed16=True)
            reducer.submit_packed()
            return
        if wl in ("vectorscope", "vectorscope_rgb"):
            ctx.vectorscope_batch_dev(batch, tables[buf], sync=False)
        elif wl == "waveform":
            ctx.waveform_batch_dev(batch, args.components, tables[buf], sync=False)
        else:
            ctx.histogram_counts_batch_dev(batch, args.components, tables[buf], sync=False)
        if reducer is not None:
            reducer.submit(tables[buf])  # the previous step's collective overlapped this accumulation
        else:
            result_tables[0] = tables[buf]

    def drain():
        if reducer is not None:
            result_tables[0] = reducer.drain()

    # ---- clock settle (setup), warm-up, then the timed region ----------------------------------------
    t_settle = time.perf_counter() + args.settle_seconds
    while time.perf_counter() < t_settle:
        for i in range(8):
            step(i)
        drain()
        torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    drain()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    ctx.profile(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    launches, kernel_ms = ctx.kernel_ms()
    ctx.profile(False)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- parity of what was just computed (untimed) ---------------------------------------------
    import scope_oracle

    o = scope_oracle.COracle()
    final = result_tables[0].view(torch.uint8).view(n, table_bytes)
    check_frames = sorted({0, n - 1})
    parity = True
    host_planes = {}
    for f in check_frames:
        src = yuv if wl == "vectorscope" else rgb
        host_planes[f] = src[f].cpu().numpy()
        if wl in ("vectorscope", "vectorscope_rgb"):
            y = host_planes[f] if wl == "vectorscope" else o.rgb_to_yuv(host_planes[f], w, 2)
            mine = o.vectorscope(y, w).astype(np.int32)
        elif wl == "waveform":
            c = args.components
            mine = o.waveform(host_planes[f], o.rgb_to_yuv(host_planes[f], w, 2), w, c).astype(np.int32)
        else:
            mine = o.histogram_counts(host_planes[f], o.rgb_to_yuv(host_planes[f], w, 2), w, args.components)
            mine = mine.astype(np.int64)
        want = torch.from_numpy(mine.reshape(-1).astype(np.int64)).cuda()
        if reducer is not None:
            dist.all_reduce(want)
        if wl != "histogram":
            want = want.clamp(max=255)
            got = final[f].to(torch.int64)
        else:
            got = final[f].view(torch.int32).to(torch.int64) & 0xFFFFFFFF
        parity = parity and bool(torch.equal(got, want))

    if use_dist:  # the line rank 0 prints speaks for every rank
        ok = torch.tensor([1 if parity else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        parity = bool(ok.item())

    # ---- the numbers ------------------------------------------------------------------------------
    pix_per_step = world * n * w * h
    value = pix_per_step * args.steps / elapsed / 1e9
    algo_bytes = n * (4 * w * h + (table_bytes if wl != "histogram" else 4096))  # SURVEY.md 8(d), per launch
    # HBM bytes per launch from the PMC counters: measured once with rocprofv3 on this exact workload and
    # committed (bench.py cannot run under a counter pass itself); null for any other configuration
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "round1", "pmc_traffic.json")))
        for e in pmc.get("entries", [pmc]):
            if world == 1 and (e["workload"], e["frames_per_step"], e["width"], e["height"]) == (wl, n, w, h) and not args.option \
                    and args.dist == "uniform" and args.components == 0x07:
                traffic = e["bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    kern_avg_ms = kernel_ms / launches if launches else float("nan")
    achieved = algo_bytes / (kern_avg_ms * 1e-3) / 1e9 if launches else float("nan")
    result = {
        "metric": "Gpix/s scope accumulation @3840×2160; achieved HBM GB/s vs peak",
        "value": round(value, 3),
        "unit": "Gpix/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8" if wl != "histogram" else "u32",
        "data": "synthetic",
        "parity": "bit-exact vs oracle" if parity else "MISMATCH vs oracle",
        "config": {
            "workload": {
                "vectorscope": "vectorscope 256x256 CbCr scatter (K1) over staged Y'CbCr planes",
                "vectorscope_rgb": "vectorscope with fused K0 over RGB planes",
                "waveform": f"waveform per-column 256-bin (K2), components {args.components:#04x}",
                "histogram": f"histogram 3x256 u32 (K3), components {args.components:#04x}",
            }[wl] + f", {w}x{h} BGRA8, {n} distinct HBM-resident frames per step, {args.dist} RGB through K0",
            "frames_per_step": n,
            "width": w,
            "height": h,
            "input_distribution": args.dist,
            "partition": "one frame batch per GPU" if world == 1 else
                         (f"{world} column slabs of {w}x{h}, no collective" if wl == "waveform" else
                          f"{world} row tiles of {w}x{h} + RCCL sum of tables"),
            "device": info["name"],
            "compute_units": info["compute_units"],
            "options": args.option,
        },
        "roofline": {
            "bound": "hbm",
            "kernel": {"vectorscope": "scope_vss_kernel", "vectorscope_rgb": "scope_vss_kernel", "waveform": "scope_wvs_kernel",
                       "histogram": "scope_his_kernel"}[wl],
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "traffic": traffic,
            "algorithmic_bytes_per_launch": algo_bytes,
            "avg_launch_ms": round(kern_avg_ms, 5),
            "launches_timed": launches,
        },
    }

    # ---- CPU baseline: the oracle (port of the reference loop) on this host, bounded sample --------
    if rank == 0 and world == 1 and not args.no_cpu:
        plane = host_planes[0]
        plane_rgb = rgb[0].cpu().numpy()
        plane_yuv = yuv[0].cpu().numpy()
        t_end = time.perf_counter() + args.cpu_seconds
        reps, t_start = 0, time.perf_counter()
        while time.perf_counter() < t_end or reps < 3:
            if wl in ("vectorscope", "vectorscope_rgb"):
                o.vectorscope(plane_yuv, w)
            elif wl == "waveform":
                o.waveform(plane_rgb, plane_yuv, w, args.components)
            else:
                o.histogram_counts(plane_rgb, plane_yuv, w, args.components)
            reps += 1
        cpu_t = time.perf_counter() - t_start
        result["cpu_baseline"] = {
            "value": round(reps * w * h / cpu_t / 1e9, 4),
            "unit": "Gpix/s",
            "cores": 1,
            "kind": "port",
            "sample": f"{reps} passes of oracle {wl} (scope_oracle.c, gcc -O2, one thread as the reference's "
                      f"single color-monitor thread) over frame 0 of the same batch, {cpu_t:.1f} s; "
                      f"host has {os.cpu_count()} cores; excludes the reference's GPU->CPU readback",
        }
        # courtesy: every host core, row/column tiles with private tables; same tables, checked here
        threads = os.cpu_count() or 1
        cpu_model = "?"
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
        except OSError:
            pass

        def all_cores(nthreads, reps=1):
            if wl in ("vectorscope", "vectorscope_rgb"):
                return o.vectorscope_mt(plane_yuv, w, nthreads, reps)
            if wl == "waveform":
                return o.waveform_mt(plane_rgb, plane_yuv, w, args.components, nthreads, reps)
            return o.histogram_counts_mt(plane_rgb, plane_yuv, w, args.components, nthreads, reps)

        one = (o.vectorscope(plane_yuv, w) if wl in ("vectorscope", "vectorscope_rgb") else
               o.waveform(plane_rgb, plane_yuv, w, args.components) if wl == "waveform" else
               o.histogram_counts(plane_rgb, plane_yuv, w, args.components))
        mt_same = bool(np.array_equal(all_cores(threads).astype(np.int64).reshape(-1), one.astype(np.int64).reshape(-1)))
        mt_reps = 20  # per call; thread start-up is paid once per call
        best = None
        for nthreads in sorted({t for t in (16, 32, 64, 128, 256, threads) if t <= threads}):
            all_cores(nthreads, 2)
            t_start, calls = time.perf_counter(), 0
            while time.perf_counter() - t_start < min(args.cpu_seconds, 10.0) / 8 or calls < 1:
                all_cores(nthreads, mt_reps)
                calls += 1
            rate = calls * mt_reps * w * h / (time.perf_counter() - t_start) / 1e9
            if best is None or rate > best[0]:
                best = (rate, nthreads, calls)
        result["cpu_baseline"]["all_cores"] = {
            "value": round(best[0], 3), "unit": "Gpix/s", "cores": best[1], "host_cores": threads, "cpu": cpu_model,
            "same_table": mt_same,
            "sample": f"best of 16..{threads} POSIX threads: {best[2]} calls x {mt_reps} passes over frame 0 (row tiles with "
                      "private tables; column tiles for the waveform; thread start-up once per call)",
        }
        del plane

    if rank == 0:
        os.write(real_stdout, (json.dumps(result) + "\n").encode())
    ctx.close()
    if use_dist:
        dist.destroy_process_group()
    if not parity:
        sys.exit(3)


if __name__ == "__main__":
    main()
