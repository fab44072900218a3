#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its config: Gpix/s of scope accumulation @3840x2160.

  python bench.py --gpus N --steps K --warmup W

N > 1 without WORLD_SIZE in the environment: bench.py STARTS ITS OWN RANKS (one process per GPU, RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* set for each, rendezvous on 127.0.0.1), watches them against a hard deadline, and prints rank 0's
line with a `launcher` object added.  A collective that hangs or fails does not take the line with it: the launcher
kills the ranks it started and tries the next reduce path (C-ABI RCCL -> torch.distributed -> replicas without a table
sum, which says so in every field that matters).  Under `python -m torch.distributed.run ... bench.py --gpus N` (the
environment has WORLD_SIZE) the process is one rank, as before.

A *step* is one pass of the hot path over one batch of synthetic input: by default the 256x256 CbCr vectorscope
scatter (BASELINE configs[1]) over `--frames` (64, the most one launch takes) distinct 3840x2160 BGRA8 frames that are
already resident in HBM (64 x 33 MB = 2.1 GB, far beyond the 256 MB Infinity Cache, so every step streams from HBM).
Frames are uniform-random RGB pushed through the K0 kernel, i.e. the staged Y'CbCr planes the reference's
vss_draw_vectorscope reads.  Every step goes through the C-ABI of libscope_hip.so (no CPU path exists in the product).

The timed region is EXACTLY K steps between barrier + synchronize (-> `value`).  Because K steps of 0.3 ms are over
before a clock or a utilisation sampler notices, the same steps then run on for at least `--min-seconds` (1 s) more
(-> `sustained`).  The accumulation kernel's average duration (-> `roofline.avg_launch_ms`, `frac_kernel`) comes from a third
leg of the same steps in which every launch carries its own begin/end HIP events on its stream (hipExtLaunchKernelGGL: the
figure rocprofv3 --kernel-trace --stats reports for the same command).  `roofline.frac` itself is the WHOLE STEP's, from the K
timed steps, on the frame pool as this process first allocated it; the same launch on a pool the library placed
(scope_frame_pool_alloc) is a leg of its own (`config.frame_pool_placement`, `roofline.frac_placed` / `frac_kernel_placed`).
The line on stdout is SHORT (bench_core.compact_line: under 6 KB -- the contract's fields, `config`, `roofline`, `cpu_baseline`
and at most 40 flat scalars of the other legs, once); every leg in full is written to bench_detail.json beside this script.

Besides the headline the default one-GPU run measures, briefly and in the same process (`--no-extras` skips them):
  variants       the other workloads of BASELINE.json at the same batch: vectorscope with K0 fused (config 2, variant ii),
                 waveform for every mode config 3 names (RGB, Y, R, G, B), histogram, the dock's three scopes (config 5's
                 kernels; algorithmic bytes = ONE read of the frames)
  distributions  SURVEY.md 8(d)'s inputs other than uniform noise -- natural, concentrated, flat, screen, ramp, padded
                 pitch -- through the headline kernel and the fused-K0 one; alpha holes through histogram and waveform
  latency        the reference's real call shape: ONE frame per call and the 3-slot ring (src/common.c:366-368,
                 src/common.h:46) -- microseconds of stream work per call, all kernels of the call included
  config1        BASELINE configs[0]: histogram over one 1280x720 frame on the CPU path (the oracle, one thread), ms per
                 frame, and whether the GPU's tables of the same frame are bit-identical
  stream         BASELINE config 5 under this clock: pinned host frames -> scope_pipe_* (upload / compute / download
                 streams, depth 3), flat out and paced at 240 fps (PCIe-inclusive; never `value`)
  config4        BASELINE configs[3] as written, at EVERY N including 1: ONE batch of 7680x4320 frames (the same frames on
                 every rank count: shared seed), rank r owns rows [4320 r/N, 4320 (r+1)/N), tables summed over RCCL
                 through the C-ABI ("strong"), with the reduce of a step overlapped and waited for
`--workload` makes any of them the headline instead (e.g. `--workload vectorscope --frames 1`).

N > 1 headline: every rank owns one 3840x2160 row tile of a frame N times as tall ("weak"); the ranks' clamped 256x256
tables are summed over RCCL through the C-ABI (scope_table_allreduce: two bins per uint32 word) and clamped once.
torch.distributed (gloo, CPU tensors) only carries the communicator id, the barriers and the max-over-ranks time: no
control-plane collective ever shares a GPU queue with the data path's.

Rank 0 prints ONE short JSON line (see README/DESIGN.md for the fields) and writes bench_detail.json.
"""
import argparse
import faulthandler
import json
import os
import sys

import bench_core
from bench_core import (HBM_PEAK_GBPS, H4K, KERNEL, METRIC, ROOT, UNPINNED, W4K, WORKLOAD_TEXT, Ctl, Dev, Overlap, Workload, compact_line, flat_summary,  # noqa: F401
                        kernel_fingerprint, kernel_us, log, make_comm, place_pool, pmc_traffic, roofline, spread, steps_ms, table_bytes_of, timed_run, write_detail)
from bench_launch import launcher, supervisor
from bench_legs import Legs, config1_leg, config4_leg, cpu_baseline


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle-seconds", type=float, default=0.5,
                    help="untimed run of the same step before the warm-up so the GPU clocks (DVFS) leave idle")
    ap.add_argument("--min-seconds", type=float, default=1.0, help="length of the sustained leg behind the K timed steps")
    ap.add_argument("--frames", type=int, default=0, help="frames per step = per launch (default 64 at 4k, 16 at 8k)")
    ap.add_argument("--pool", type=int, default=0, help="distinct HBM-resident frames rotated through (default: max(frames, 24) for small launches)")
    ap.add_argument("--size", default="4k", choices=["4k", "8k"],
                    help="frame size: 3840x2160 (the metric's configuration) or 7680x4320 (BASELINE config 4's frame)")
    ap.add_argument("--geometry", default="", help="WxH instead of --size (emulation runs and experiments; not the metric's configuration)")
    ap.add_argument("--workload", default="vectorscope",
                    choices=["vectorscope", "vectorscope_rgb", "waveform", "histogram", "three_scope", "stream"])
    ap.add_argument("--components", type=lambda s: int(s, 0), default=0x07)
    ap.add_argument("--placement-candidates", type=int, default=6,
                    help="frame pools allocated and probed before the fastest is kept (1 = take the first, as round 1 did)")
    ap.add_argument("--dist", default="uniform", choices=["uniform", "concentrated", "flat", "ramp", "natural", "screen"])
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="bound of the CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only: no variants / distributions / latency / stream / config1 / config4 legs")
    ap.add_argument("--config4-frames", type=int, default=16, help="7680x4320 frames in BASELINE config 4's one batch")
    ap.add_argument("--config4-geometry", default="7680x4320")
    ap.add_argument("--strong", action="store_true",
                    help="make BASELINE config 4 (one 7680x4320 batch, row tiles, strong scaling) the headline")
    ap.add_argument("--force-reduce", action="store_true",
                    help="run the RCCL table reduction even with one rank (exercises the N > 1 code path on a 1-GPU box)")
    ap.add_argument("--serial-reduce", action="store_true",
                    help="enqueue the RCCL table sum on the accumulation's own stream instead of overlapping step i's sum with "
                         "step i+1's accumulation on a second stream")
    ap.add_argument("--torch-reduce", action="store_true", help="N > 1: sum the tables with torch.distributed (TableAllReduce) "
                    "instead of the C-ABI's scope_table_allreduce")
    ap.add_argument("--no-collective", action="store_true",
                    help="N > 1: independent replicas, NO table sum (the launcher's last resort; every field says so)")
    ap.add_argument("--launch-timeout", type=float, default=480.0, help="self-launched ranks: seconds before an attempt is given up")
    ap.add_argument("--rank-timeout", type=float, default=0.0,
                    help="seconds after which a rank dumps its stacks and exits (default: the launcher's deadline minus 15 s, or none)")
    ap.add_argument("--backend", default="hip", choices=["hip", "emul"],
                    help="emul = tests/emul/libscope_emul.so on CPU tensors with gloo: exercises this file's N-rank logic without a GPU "
                         "(tests only; its numbers mean nothing)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="let ranks share a GPU when there are fewer GPUs than ranks (the table sum then needs SCOPE_RCCL_LIBRARY "
                         "= tools/libshm_rccl_hip.so or --torch-reduce over gloo: RCCL refuses two ranks on one device)")
    ap.add_argument("--option", action="append", default=[], help="key=value passed to scope_ctx_set_option")
    ap.add_argument("--dry-run", action="store_true",
                    help="print the per-rank plan of this command line (rows, launches, wire bytes, reduce path, streams) as one JSON line "
                         "and exit: no device, no torch, no library")
    ap.add_argument("--lib", default=None, help="another build of libscope_hip.so (A/B of kernel variants; the line names it)")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.dry_run:
        import json

        from bench_launch import dry_run_plan

        print(json.dumps(dry_run_plan(args)), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launcher(args, sys.argv[1:]))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not os.environ.get("BENCH_WORKER") and not os.environ.get("BENCH_SELF_LAUNCHED"):
        sys.exit(supervisor(args, sys.argv[1:]))  # a rank torch.distributed.run (or anybody else) started
    run_rank(args)


def run_rank(args):
    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner through C
    # stdio when a communicator exists), so everything else in this process goes to stderr: file descriptor 1 is
    # pointed at stderr for the run and the JSON line is written to the saved, real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the environment decides, this process is rank {rank} of {world}")
    rank_timeout = args.rank_timeout or (args.launch_timeout - 15 if os.environ.get("BENCH_SELF_LAUNCHED") else 0)
    if rank_timeout > 0:
        # a rank that is stuck (a collective its peers never entered) says where and leaves: the launcher, or
        # torch.distributed.run, then ends the others instead of the whole job hanging silently
        faulthandler.dump_traceback_later(rank_timeout, exit=True)

    import numpy as np
    import torch
    import torch.distributed as dist

    import obs_color_monitor_amd as ocm
    import scope_oracle
    from obs_color_monitor_amd.synth import make_frames, pad_pitch, punch_alpha_holes

    dev = Dev(torch, args, local_rank)
    lib_path = os.path.join(ROOT, "tests", "emul", "libscope_emul.so") if dev.emul else (os.path.abspath(args.lib) if args.lib else None)
    if dev.emul:
        os.environ.setdefault("SCOPE_RCCL_LIBRARY", os.path.join(ROOT, "tests", "emul", "libshm_rccl_emul.so"))
    multi = world > 1
    use_dist = multi or args.force_reduce
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        dist.init_process_group("gloo")  # control plane only (class Ctl)
    ctl = Ctl(torch, dist if multi else None, world)
    data_group = None
    if multi and args.torch_reduce and not dev.emul and not dev.shared:
        data_group = dist.new_group(backend="nccl")  # RCCL through c10d, for TableAllReduce

    ctx = ocm.ScopeContext(dev.index, lib_path)  # raises without the library / a GPU: no fallback
    # one explicit (non-default) stream for torch AND the C-ABI: generation, kernels, RCCL and the HIP
    # events that time the kernels are all ordered on it (a NULL handle would mean "ctx's own stream")
    stream = dev.stream()
    if not dev.emul:
        torch.cuda.set_stream(stream)
        ctx.set_stream(stream.cuda_stream)
    for kv in args.option:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    info = ctx.device_info()
    o = scope_oracle.COracle()
    plain = not args.option and args.dist == "uniform" and args.components == 0x07 and not args.geometry and not args.lib

    if args.workload == "stream":
        from bench_stream import stream_leg

        res = stream_leg(torch, np, ctx, o, seconds=max(2.0, args.min_seconds), dist_kind="natural" if args.dist == "uniform" else args.dist)
        flat = res["flat_out"]
        result = {
            "metric": "sustained frames/s, 3840x2160, histogram + waveform + vectorscope per frame, pinned host frames in, "
                      "tables out (BASELINE config 5; PCIe-inclusive)",
            "value": flat["fps"], "unit": "frames/s", "n_gpus": 1, "steps": flat["frames"], "warmup": 0,
            "ms_per_step": round(1e3 / flat["fps"], 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic", "parity": res["parity"],
            "config": {"workload": "stream: scope_pipe_* (upload / compute / download hipStreams, depth 3), K0 fused, "
                                   "vectorscope + waveform 0x07 + histogram 0x07 per frame", "device": info["name"]},
            "stream": res,
        }
        os.write(real_stdout, (compact_line(result, write_detail(result)) + "\n").encode())
        ctx.close()
        sys.exit(0 if res["parity"].startswith("bit-exact") else 3)

    # ---- synthetic input, resident in HBM ------------------------------------------------------
    if args.geometry:
        w, h = (int(v) for v in args.geometry.lower().split("x"))
    else:
        w, h = (W4K, H4K) if args.size == "4k" else (2 * W4K, 2 * H4K)
    n = args.frames or (64 if w * h <= W4K * H4K else 16)
    pool = args.pool or (n if n >= 24 else (24 // n) * n)  # small launches rotate over >= 24 frames (0.8 GB > the Infinity Cache)
    wl = args.workload
    rgb, yuv = make_frames(torch, ctx, pool, w, h, args.dist, 0x9E3779B9 + rank, device=dev.device)
    dev.sync()
    placement, pools = None, []
    work = Workload(torch, ctx, wl, rgb, yuv, w, h, n, args.components)
    table_bytes = work.tbytes
    frame0 = (rgb[0].cpu().numpy(), yuv[0].cpu().numpy())  # the CPU baseline's sample

    # ---- N > 1: the table sum over RCCL -----------------------------------------------------------------------------
    reducer, comm, reduce_path, overlap, rccl_info = None, None, None, None, None
    wire = None
    sums_tables = wl in ("vectorscope", "vectorscope_rgb", "histogram")
    if use_dist and args.no_collective:
        reduce_path = ("NONE: independent replicas, no table sum (--no-collective; as the launcher's last resort this means the "
                       "collective paths failed or hung: see launcher.attempts)")
    elif use_dist and sums_tables:
        if not args.torch_reduce:
            comm, why = make_comm(ctx, ctl, world, rank)
            if comm is None:
                log(f"rank {rank}: no C-ABI communicator ({why}); every rank falls back to torch.distributed together")
        if comm:
            rccl_info = ctx.rccl_comm_info(comm)
            wire = [torch.empty((n, 32768 if wl != "histogram" else 1024), dtype=torch.int32, device=dev.device) for _ in range(2)]
            if args.serial_reduce:
                reduce_path = "C-ABI: scope_table_allreduce (ncclAllReduce uint32 sum) every step, unpack/clamp on the same stream"
            else:
                reduce_path = ("C-ABI: scope_table_allreduce (ncclAllReduce uint32 sum) + unpack/clamp of EVERY step on a second "
                               "hipStream: step i's table sum runs under step i+1's accumulation (two table buffers)")
                overlap = Overlap(dev, ocm, lib_path, stream)
        elif multi:
            from obs_color_monitor_amd.row_tiles import TableAllReduce

            # one collective per 8 steps: c10d's per-collective host cost is about a step's GPU time
            reducer = TableAllReduce(ctx, n * table_bytes, dev.device, "sum_u32" if wl == "histogram" else "sat_u8", bucket=8, group=data_group)
            reduce_path = ("torch.distributed (" + ("nccl = RCCL" if data_group is not None else "gloo") +
                           "): TableAllReduce, 8 steps per collective, side stream")
        else:
            reduce_path = "NONE: --force-reduce with one rank needs the C-ABI communicator, which could not be made"
    result_tables = [None]

    def step(i):
        if comm:
            b, buf = work.batches[i % work.groups], i & 1
            rctx = ctx
            if overlap:
                overlap.before_compute(buf)  # this buffer's previous sum must be through before it is overwritten
                rctx = overlap.ctx
            if wl == "histogram":
                ctx.histogram_counts_batch_dev(b, args.components, wire[buf], sync=False)
                if overlap:
                    overlap.after_compute(buf)
                rctx.table_allreduce(comm, wire[buf], n * 1024)
                result_tables[0] = wire[buf].view(torch.uint8).view(n, 4096)
            else:
                ctx.vectorscope_batch_dev(b, wire[buf], sync=False, packed16=True)
                if overlap:
                    overlap.after_compute(buf)
                rctx.table_allreduce(comm, wire[buf], n * 32768)
                rctx.unpack16_clamp_dev(wire[buf], work.tables[buf], n * 32768)
                result_tables[0] = work.tables[buf]
            if overlap:
                overlap.after_reduce(buf)
            work.last = (i % work.groups, buf)
            return
        if reducer is not None:
            b = work.batches[i % work.groups]
            if wl == "histogram":
                ctx.histogram_counts_batch_dev(b, args.components, work.tables[i & 1], sync=False)
                reducer.submit(work.tables[i & 1])
            else:
                # tables go straight into the reducer's wire buffer in its packed format: no packing pass
                ctx.vectorscope_batch_dev(b, reducer.wire_buffer(), sync=False, packed16=True)
                reducer.submit_packed()
            work.last = (i % work.groups, i & 1)
            return
        work.step(i)

    elapsed, sustained, launches, kernel_ms = timed_run(dev, ctl, ctx, stream, step, args.steps, args.warmup,
                                                        args.settle_seconds, args.min_seconds, multi)
    if reducer is not None:
        result_tables[0] = reducer.drain()
    dev.sync()
    elapsed = ctl.max(elapsed)

    # ---- parity of what was just computed (untimed) ---------------------------------------------
    if comm or reducer is not None:
        # the reduced table of the LAST step: every rank's oracle table of its own tile, summed, clamped once
        g, buf = work.last
        final = result_tables[0].view(torch.uint8).view(n, table_bytes)
        parity = True
        for f in sorted({0, n - 1}):
            pr, py = rgb[g * n + f].cpu().numpy(), yuv[g * n + f].cpu().numpy()
            if wl == "histogram":
                want = ctl.sum_array(o.histogram_counts(pr, py, w, args.components).astype(np.int64).reshape(-1))
                got = final[f].cpu().numpy().view(np.uint32).astype(np.int64)
            else:
                want = np.minimum(ctl.sum_array(o.vectorscope(py, w).astype(np.int64).reshape(-1)), 255)
                got = final[f].cpu().numpy().astype(np.int64)
            parity = parity and bool(np.array_equal(got, want))
    else:
        work.step(args.steps - 1)
        dev.sync()
        parity = work.check(o, rgb, yuv, np)
    parity = ctl.all_true(parity)  # the line rank 0 prints speaks for every rank

    # ---- the numbers ------------------------------------------------------------------------------
    pix_per_step = world * n * w * h
    value = pix_per_step * args.steps / elapsed / 1e9
    ms_per_step = elapsed / args.steps * 1e3
    traffic, traffic_src = pmc_traffic(wl, n, w, h, plain and world == 1)
    rf = roofline(wl, work.algo_bytes, launches, kernel_ms, ms_per_step, traffic)
    rf["traffic_source"] = traffic_src
    rf["pool"] = "as this process first allocated it (no placement search): see frame_pool_placement for scope_frame_pool_alloc's pool"
    if args.placement_candidates > 1 and not dev.emul and world == 1 and not use_dist:
        # The same launch on a pool the LIBRARY placed (scope_frame_pool_alloc: candidates held at once, each read by the
        # accumulation kernel, the fastest kept): what a host that keeps a long-lived pool can have.  A leg of its own, after
        # the headline; the headline above never saw it.
        src_name = "yuv" if wl == "vectorscope" else "rgb"
        src = yuv if wl == "vectorscope" else rgb
        placed, rates = place_pool(torch, dev, ctx, src, args.placement_candidates, pools)
        pw = Workload(torch, ctx, wl, placed if src_name == "rgb" else rgb, placed if src_name == "yuv" else yuv, w, h, n, args.components)
        # Both pools in ALTERNATING rounds of the same K steps (round 6): the chip reads 3-4 % slower after a minute of load than in
        # its first seconds (profiles/round6/placement_vmm_probe_c6.txt, section 8 against section 1), so a leg measured after the
        # headline cannot be held against the headline -- round 5's line did, and called the placed pool 6.7 % slower.
        first_ms, first_kus, p_mss, p_kus = [], [], [], []
        for _ in range(3):
            for step_fn, ms_l, kus_l in ((work.step, first_ms, first_kus), (pw.step, p_mss, p_kus)):
                ms_l.append(steps_ms(dev, ctl, step_fn, args.steps, args.warmup))
                kus_l.append(kernel_us(ctx, lambda fn=step_fn: fn(0), warm=4, reps=24))
        p_ok = pw.check(o, placed if src_name == "rgb" else rgb, placed if src_name == "yuv" else yuv, np, frames=spread(n, 3))
        med = lambda xs: sorted(xs)[len(xs) // 2]  # noqa: E731
        p_ms, f_ms = med(p_mss), med(first_ms)
        p_rf = roofline(wl, pw.algo_bytes, 1, med(p_kus) * 1e-3, p_ms, None)
        f_rf = roofline(wl, work.algo_bytes, 1, med(first_kus) * 1e-3, f_ms, None)
        placement = {"candidates": args.placement_candidates, "pool": src_name,
                     "by": "scope_frame_pool_alloc (C-ABI): `candidates` plain allocations held at once, each read by the vectorscope "
                           "accumulation kernel in three interleaved rounds (own begin/end stamps), the best median kept; untimed set-up",
                     "pool_read_gbps": rates, "value": round(n * w * h / (p_ms * 1e-3) / 1e9, 3), "ms_per_step": round(p_ms, 4),
                     "frac": p_rf["frac"], "frac_kernel": p_rf["frac_kernel"], "avg_launch_ms": p_rf["avg_launch_ms"], "parity": bool(p_ok),
                     "measured": "three alternating rounds of the K steps on the process's first pool and on the placed pool; medians",
                     "first_pool_same_rounds": {"ms_per_step": round(f_ms, 4), "frac": f_rf["frac"], "frac_kernel": f_rf["frac_kernel"],
                                                "rounds_ms": [round(x, 4) for x in first_ms]},
                     "rounds_ms": [round(x, 4) for x in p_mss]}
        rf["frac_first_pool_same_rounds"] = f_rf["frac"]
        rf["frac_kernel_first_pool_same_rounds"] = f_rf["frac_kernel"]
        rf["frac_placed"] = p_rf["frac"]
        rf["frac_kernel_placed"] = p_rf["frac_kernel"]
        # what the search is worth: placed pool over first pool INSIDE the same alternating rounds (both legs run a minute after the
        # headline, when the chip reads 3-4 % slower: neither compares with `frac`)
        rf["placed_over_first_same_rounds"] = round(p_rf["frac"] / f_rf["frac"], 4) if f_rf["frac"] else None
        rf["placed_over_first_same_rounds_kernel"] = round(p_rf["frac_kernel"] / f_rf["frac_kernel"], 4) if f_rf["frac_kernel"] else None
        parity = parity and bool(p_ok)
        del pw, placed
        dev.sync()
        pools.pop().free()
    sustained["value"] = round(pix_per_step / (sustained["ms_per_step"] * 1e-3) / 1e9, 3) if sustained["ms_per_step"] > 0 else None
    if world == 1:
        partition = "one frame batch per GPU"
    elif args.no_collective:
        partition = f"{world} independent replicas of one {w}x{h} batch each: NO table sum"
    else:
        partition = f"{world} row tiles of {w}x{h} + sum of the ranks' tables"
    result = {
        "metric": METRIC,
        "value": round(value, 3),
        "unit": "Gpix/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8" if wl != "histogram" else "u32",
        "data": "synthetic",
        "parity": ("bit-exact vs oracle" if parity else "MISMATCH vs oracle") + UNPINNED,
        "config": {
            "workload": WORKLOAD_TEXT[wl] + (f", components {args.components:#04x}" if wl in ("waveform", "histogram", "three_scope") else "")
                        + f", {w}x{h} BGRA8, {n} frame(s) per launch from {pool} distinct HBM-resident frames, {args.dist} RGB through K0",
            "frames_per_step": n,
            "frame_pool": pool,
            "width": w,
            "height": h,
            "input_distribution": args.dist,
            "partition": partition,
            "frame_pool_placement": placement,
            "reduce": reduce_path,
            "rccl": None if rccl_info is None else {
                "ranks_in_communicator": rccl_info["ranks"], "rank": rccl_info["rank"], "version": rccl_info["version"],
                "library": os.environ.get("SCOPE_RCCL_LIBRARY") or "librccl (dlopen by libscope_hip.so)",
                "what": "ncclCommCount / ncclCommUserRank / ncclGetVersion of the data path's communicator (scope_rccl_comm_info)"},
            "launch": ("supervised rank under torch.distributed.run" if os.environ.get("BENCH_WORKER") else
                       "self-launched rank" if os.environ.get("BENCH_SELF_LAUNCHED") else ("rank of a foreign launcher" if multi else "single process")),
            "backend": "libscope_hip.so on " + info["name"] if not dev.emul else "EMULATION (tests/emul, CPU): numbers are meaningless",
            "gpu_shared_by_ranks": bool(dev.shared),
            "device": info["name"],
            "compute_units": info["compute_units"],
            "options": args.option,
            "library": args.lib or "obs-color-monitor_amd/libscope_hip.so",
            "kernel_sources_sha256_16": kernel_fingerprint(),
        },
        "roofline": rf,
        "sustained": sustained,
    }

    # ---- the other legs (one-GPU run; short, >= 0.5 s each) ---------------------------------------------------------
    extras = not args.no_extras and (w, h) == (W4K, H4K) and not dev.emul
    if extras and world == 1 and not use_dist:
        del work
        legs = Legs(dev, ctl, ctx, stream, o, np, torch, w, h, n)
        variants = {}
        for v, comp in (("vectorscope_rgb", 0x07), ("waveform", 0x07), ("waveform_y", 0x20), ("waveform_r", 0x04),
                        ("waveform_g", 0x02), ("waveform_b", 0x01), ("histogram", 0x07), ("three_scope", 0x07)):
            base = v.split("_")[0] if v.startswith("waveform") else v
            if v == wl:
                continue
            variants[v] = legs.variant(base, rgb, yuv, comp, steps=max(4, min(args.steps, 40)),
                                       seconds=0.5 if v in ("vectorscope_rgb", "waveform", "histogram", "three_scope") else 0.3)
        result["variants"] = variants
        # the reference's call shape: one surface per callback, ring of three
        latency = {}
        for nf in (1, 3):
            for v in ("vectorscope", "vectorscope_rgb", "histogram", "waveform", "three_scope", "vectorscope+side", "vectorscope_rgb+side"):
                base, side = v.split("+")[0], v.endswith("+side")
                lw = Workload(torch, ctx, base, rgb[:24], yuv[:24], w, h, nf, 0x07, side_reduce=side)
                el, sus, ln, kms = timed_run(dev, ctl, ctx, stream, lw.step, 200, 50, 0.15, 0.3, finish=lw.finish)
                lw.step(0)
                lw.finish()
                dev.sync()
                ok = lw.check(o, rgb, yuv, np, frames=[nf - 1])
                us = sus["ms_per_step"] * 1e3
                latency[f"{base}_x{nf}" + ("_side_reduce" if side else "")] = {
                    "frames_per_call": nf, "us_per_call": round(us, 2), "us_host_enqueue_per_call": round(sus["host_enqueue_ms_per_step"] * 1e3, 2),
                    "us_accumulation_kernel": round(kms / ln * 1e3, 2) if ln else None,
                    "gbps": round(lw.algo_bytes / (us * 1e-6) / 1e9, 1), "frac_of_peak": round(lw.algo_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                    "parity": bool(ok)}
                if side:
                    latency[f"{base}_x{nf}_side_reduce"]["what"] = (
                        "SCOPE_SIDE_REDUCE: the partial-table reduction of call i on the context's second stream, under the accumulation "
                        "of call i + 1; tables complete after scope_ctx_fence (enqueued once, at the end of the timed calls)")
                del lw
        latency["note"] = ("us_per_call = stream time per call in steady state, calls back to back on one stream over 24 rotating "
                           "frames: every kernel of the call (accumulation + partial-table reduction) and the gaps between them, no "
                           "event brackets in between; us_accumulation_kernel comes from a separate bracketed leg and carries "
                           "~2-3 us of event overhead at this scale (rocprofv3 --stats of the same launches: profiles/)")
        result["latency"] = latency
        try:
            result["distributions"] = legs.distributions(make_frames, pad_pitch, punch_alpha_holes, rgb, yuv)
        except Exception as e:  # noqa: BLE001
            result["distributions"] = {"error": repr(e)}
        try:
            result["config1"] = config1_leg(np, ctx, o)
        except Exception as e:  # noqa: BLE001
            result["config1"] = {"error": repr(e)}
        try:
            from bench_stream import stream_leg

            result["stream"] = stream_leg(torch, np, ctx, o, seconds=1.5, dist_kind="natural")
        except Exception as e:  # noqa: BLE001
            result["stream"] = {"error": repr(e)}

    # ---- BASELINE config 4 as written: one 7680x4320 batch in N row tiles, strong scaling, at every N ---------------
    c4_wanted = (not args.no_extras and (w, h) == (W4K, H4K)) or args.strong
    if c4_wanted and wl in ("vectorscope", "vectorscope_rgb") and not args.no_collective:
        try:
            del rgb, yuv
            dev.empty_cache()
            c4_comm, c4_overlap = comm, overlap
            if c4_comm is None and not args.torch_reduce:
                c4_comm, why = make_comm(ctx, ctl, world, rank)
                if c4_comm is None:
                    raise RuntimeError(f"no C-ABI communicator for the config-4 leg: {why}")
            if c4_comm is None:
                raise RuntimeError("the config-4 leg sums through the C-ABI; this run was asked for --torch-reduce")
            if c4_overlap is None and not args.serial_reduce:
                c4_overlap = Overlap(dev, ocm, lib_path, stream)
            c4w, c4h = (int(v) for v in args.config4_geometry.lower().split("x"))
            result["config4"] = config4_leg(dev, ctl, torch, np, ctx, o, c4_comm, rank, world, c4w, c4h,
                                            frames=args.config4_frames, overlap=c4_overlap,
                                            seconds=0.6 if not dev.emul else 0.0)
            result["config4"]["rccl_ranks_in_communicator"] = ctx.rccl_comm_info(c4_comm)["ranks"]
            result["config4"]["reduce_path"] = ("C-ABI scope_table_allreduce (ncclAllReduce uint32 sum of packed u16 pairs), every step"
                                                + (", second stream" if c4_overlap is not None else ", accumulation stream"))
            if c4_comm is not comm:
                dev.sync()
                ctx.rccl_comm_destroy(c4_comm)
            if c4_overlap is not None and c4_overlap is not overlap:
                c4_overlap.ctx.close()
        except Exception as e:  # noqa: BLE001
            result["config4"] = {"error": repr(e)}
        if args.strong and "error" not in result["config4"]:
            c4 = result["config4"]
            result.update({"value": c4["value"], "ms_per_step": c4["ms_per_step"], "scaling": "strong"})
            result["config"]["partition"] = c4["partition"]

    # ---- CPU baseline: the oracle (port of the reference loop) on this host, bounded sample --------
    if rank == 0 and world == 1 and not args.no_cpu:
        result["cpu_baseline"] = cpu_baseline(np, o, "vectorscope" if wl == "three_scope" else wl, frame0[0], frame0[1],
                                              w, h, args.components, args.cpu_seconds)

    # ---- ONE line on stdout, short enough for any reader (compact_line: < 6 KB; the contract's fields, config, roofline,
    # cpu_baseline, <= 40 flat scalars of the other legs); every leg in full goes to bench_detail.json beside this script
    faulthandler.cancel_dump_traceback_later()
    if rank == 0:
        result["flat"] = flat_summary(result)
        detail = write_detail(result)
        os.write(real_stdout, (compact_line(result, detail) + "\n").encode())
    if comm:
        dev.sync()
        ctx.rccl_comm_destroy(comm)
    if overlap:
        overlap.ctx.close()
    for fp in pools:
        fp.free()
    ctx.close()
    if multi:
        ctl.barrier()
        dist.destroy_process_group()
    if not parity:
        sys.exit(3)


if __name__ == "__main__":
    main()
