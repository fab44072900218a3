#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its config: Gpix/s of scope accumulation @3840x2160.

  python bench.py --gpus N --steps K --warmup W         (N > 1: launched under torch.distributed.run)

A *step* is one pass of the hot path over one batch of synthetic input: by default the 256x256 CbCr vectorscope
scatter (BASELINE configs[1]) over `--frames` (64, the most one launch takes) distinct 3840x2160 BGRA8 frames that are
already resident in HBM (64 x 33 MB = 2.1 GB, far beyond the 256 MB Infinity Cache, so every step streams from HBM).
Frames are uniform-random RGB pushed through the K0 kernel, i.e. the staged Y'CbCr planes the reference's
vss_draw_vectorscope reads.  Every step goes through the C-ABI of libscope_hip.so (no CPU path exists in the product).

The timed region is EXACTLY K steps between barrier + synchronize (-> `value`).  Because K steps of 0.3 ms are over
before a clock or a utilisation sampler notices, the same steps then run on for at least `--min-seconds` (1 s) more
(-> `sustained`).  The accumulation kernel's average duration (-> `roofline`) comes from a third leg of the same steps
in which every launch carries its own begin/end HIP events on its stream (hipExtLaunchKernelGGL: the figure rocprofv3
--kernel-trace --stats reports for the same command); it is a separate leg because reading events costs the host a
few microseconds per launch, which a one-frame step would feel.

Besides the headline the default one-GPU run measures, briefly and in the same process (`--no-extras` skips them):
  variants   the other workloads of BASELINE.json at the same batch: vectorscope with K0 fused (config 2, variant ii),
             waveform (config 3), histogram, and the dock's three scopes on one frame set (config 5's kernels; its
             algorithmic bytes are ONE read of the frames)
  latency    the reference's real call shape: ONE frame per call and the 3-slot ring (src/common.c:366-368,
             src/common.h:46) -- microseconds of stream work per call, all kernels of the call included
  stream     BASELINE config 5 under this clock: pinned host frames -> scope_pipe_* (upload / compute / download
             streams, depth 3), flat out and paced at 240 fps (PCIe-inclusive; never `value`)
`--workload` makes any of them the headline instead (e.g. `--workload vectorscope --frames 1`).

N > 1: every rank owns one 3840x2160 row tile of a frame N times as tall ("weak"); the ranks' clamped 256x256 tables
are summed over RCCL through the C-ABI (scope_table_allreduce: two bins per uint32 word, no torch.distributed in the
data path; torch.distributed only carries the communicator id, the barriers and the max-over-ranks time) and clamped
once.  The same run then measures BASELINE config 4 as written (`config4`): ONE batch of 7680x4320 frames, rank r
owns rows [4320 r/N, 4320 (r+1)/N) of every frame ("strong"), with the reduce of every step waited for.

Rank 0 prints ONE JSON line (see README/DESIGN.md for the fields).
"""
import argparse
import hashlib
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
PROFILE_DIR = os.path.join(ROOT, "profiles", "round2")

WORKLOAD_TEXT = {
    "vectorscope": "vectorscope 256x256 CbCr scatter (K1) over staged Y'CbCr planes",
    "vectorscope_rgb": "vectorscope with fused K0 over RGB planes",
    "waveform": "waveform per-column 256-bin (K2)",
    "histogram": "histogram 3x256 u32 (K3)",
    "three_scope": "vectorscope (fused K0) + waveform + histogram of every frame (scope_three_scopes_batch): batches in two "
                   "passes with the histogram riding on the waveform's; one-frame launches read the plane once",
}
KERNEL = {"vectorscope": "scope_vss_kernel", "vectorscope_rgb": "scope_vss_kernel", "waveform": "scope_wvs_kernel",
          "histogram": "scope_his_kernel",
          "three_scope": "scope_vss_kernel + scope_wvs_kernel (batches) / scope_dock_kernel (one-frame launches: one read)"}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def table_bytes_of(wl, w):
    return {"vectorscope": 65536, "vectorscope_rgb": 65536, "waveform": 256 * w * 4, "histogram": 4096,
            "three_scope": 65536 + 256 * w * 4 + 4096}[wl]


class Workload:
    """One scope workload over a pool of HBM-resident frames, `n` frames per step, rotating through the pool."""

    def __init__(self, torch, ctx, wl, rgb, yuv, w, h, n, components=0x07):
        self.torch, self.ctx, self.wl, self.w, self.h, self.n, self.comp = torch, ctx, wl, w, h, n, components
        pool = rgb.shape[0]
        self.groups = pool // n
        src = yuv if wl == "vectorscope" else rgb
        self.batches = [ctx.make_device_batch(src[g * n:(g + 1) * n], 4 * w, w, h, planes_are_rgb=(wl != "vectorscope"), colorspace=2)
                        for g in range(self.groups)]
        self.tbytes = table_bytes_of(wl, w)
        if wl == "three_scope":
            self.vss = [torch.empty((n, 65536), dtype=torch.uint8, device="cuda") for _ in range(2)]
            self.wvs = [torch.empty((n, 256 * w * 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
            self.his = [torch.empty((n, 1024), dtype=torch.int32, device="cuda") for _ in range(2)]
        else:
            self.tables = [torch.empty((n, self.tbytes), dtype=torch.uint8, device="cuda") for _ in range(2)]
        self.last = (0, 0)
        # 4 B per pixel read once + the tables once per frame
        self.algo_bytes = n * (4 * w * h + self.tbytes)

    def step(self, i):
        ctx, b, buf = self.ctx, self.batches[i % self.groups], i & 1
        wl = self.wl
        if wl in ("vectorscope", "vectorscope_rgb"):
            ctx.vectorscope_batch_dev(b, self.tables[buf], sync=False)
        elif wl == "waveform":
            ctx.waveform_batch_dev(b, self.comp, self.tables[buf], sync=False)
        elif wl == "histogram":
            ctx.histogram_counts_batch_dev(b, self.comp, self.tables[buf], sync=False)
        else:
            # one C-ABI call: small launches read the planes once (scope_dock_kernel), batches take two passes
            ctx.three_scopes_batch_dev(b, self.comp, self.comp, self.vss[buf], self.wvs[buf], self.his[buf], sync=False)
        self.last = (i % self.groups, buf)

    def check(self, o, rgb, yuv, np, frames=None):
        """Tables of the most recent step against the oracle (frames: indices inside the step's batch)."""
        g, buf = self.last
        torch = self.torch
        ok = True
        for f in (frames if frames is not None else sorted({0, self.n - 1})):
            pr, py = rgb[g * self.n + f].cpu().numpy(), yuv[g * self.n + f].cpu().numpy()
            if self.wl in ("vectorscope", "vectorscope_rgb", "three_scope"):
                got = (self.vss if self.wl == "three_scope" else self.tables)[buf][f].cpu().numpy().reshape(256, 256)
                ok &= bool(np.array_equal(got, o.vectorscope(py, self.w)))
            if self.wl in ("waveform", "three_scope"):
                got = (self.wvs if self.wl == "three_scope" else self.tables)[buf][f].cpu().numpy().reshape(256, self.w, 4)
                ok &= bool(np.array_equal(got, o.waveform(pr, py, self.w, self.comp)))
            if self.wl in ("histogram", "three_scope"):
                t = (self.his[buf][f].view(torch.uint8) if self.wl == "three_scope" else self.tables[buf][f])
                ok &= bool(np.array_equal(t.cpu().numpy().view(np.uint32).reshape(256, 4), o.histogram_counts(pr, py, self.w, self.comp)))
        return ok


def agree(torch, dist, keep_going):
    """Loops that run 'for so many seconds' must run the SAME number of steps on every rank -- a step holds a collective, and
    a rank that enqueues one more than its peers waits for ever.  So the clock of no single rank decides: the loop goes on
    while ANY rank wants to (one tiny all-reduce per chunk of steps, which also drains the stream)."""
    if dist is None:
        return keep_going
    # nothing of the data path -- which has a communicator of its own, on a stream of its own -- is in flight while
    # torch.distributed's communicator runs: two communicators never have collectives queued at the same time
    torch.cuda.synchronize()
    t = torch.tensor([1 if keep_going else 0], dtype=torch.int32, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(t.item())


def timed_run(torch, ctx, stream, step, steps, warmup, settle_s, min_s, dist=None):
    """settle -> warm-up -> EXACTLY `steps` steps between (barrier +) synchronize -> at least min_s more of the same ->
    a short leg with the accumulation kernel bracketed by HIP events (the bracket costs a few microseconds of stream time
    per launch -- a quarter of a one-frame step -- so it stays out of the two timed legs).
    Returns (elapsed of the K steps, sustained dict, bracketed kernel launches, their summed ms)."""
    t_end = time.perf_counter() + settle_s
    i = 0
    while agree(torch, dist, time.perf_counter() < t_end):
        for _ in range(8):
            step(i)
            i += 1
        torch.cuda.synchronize()
    for j in range(warmup):
        step(j)
    torch.cuda.synchronize()  # the data path's own communicator is idle before torch.distributed's barrier runs
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(steps):
        step(j)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    # the sustained leg: the same step for >= min_s; the host only enqueues (its own pace is reported beside the GPU's)
    s_steps, e0, e1 = 0, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    chunk = max(8, steps)
    t1 = time.perf_counter()
    e0.record(stream)
    while True:
        for j in range(chunk):
            step(s_steps + j)
        s_steps += chunk
        if not agree(torch, dist, time.perf_counter() - t1 < min_s):
            break
    host_s = time.perf_counter() - t1
    e1.record(stream)
    torch.cuda.synchronize()
    sustained = {"seconds": round(time.perf_counter() - t1, 3), "steps": s_steps,
                 "ms_per_step": round(e0.elapsed_time(e1) / s_steps, 5),
                 "host_enqueue_ms_per_step": round(host_s / s_steps * 1e3, 5)}
    ctx.profile(True)
    for j in range(min(400, max(60, steps))):
        step(j)
    torch.cuda.synchronize()
    launches, kernel_ms = ctx.kernel_ms()
    ctx.profile(False)
    return elapsed, sustained, launches, kernel_ms


def place_pool(torch, ctx, frames, probe, candidates, good_us=0.0):
    """WHERE a frame pool lies in HBM decides a few per cent of every streaming kernel's rate, and the first pools a
    process allocates are regularly the slow ones (profiles/round2/placement_probe_g8.txt: 6300-6490 GB/s for the first
    against 6800-6860 for the ten that follow, same frames, same kernel, same process).  A host that
    keeps a long-lived pool can afford to look: allocate `candidates` copies, time the accumulation kernel on each
    (its own begin/end stamps), keep the fastest and give the rest back; when none of them is as fast as `good_us` (what a
    well-placed pool reads in), up to as many again are tried.  Untimed set-up, rank-local, no collective.
    Returns (the pool to use, kernel microseconds per candidate, index kept)."""
    pools, us = [frames], []
    k = 0
    while k < candidates or (good_us and k < 2 * candidates and min(us) > good_us):
        if k:
            pools.append(frames.clone())
        fn = probe(pools[k])
        for _ in range(12):
            fn()
        ctx.sync()
        ctx.profile(True)
        for _ in range(24):
            fn()
        ctx.sync()
        nl, ms = ctx.kernel_ms()
        ctx.profile(False)
        us.append(round(ms / max(nl, 1) * 1e3, 2))
        k += 1
    best = min(range(len(us)), key=lambda i: us[i])
    keep = pools[best]
    del pools
    torch.cuda.empty_cache()
    return keep, us, best


class Overlap:
    """The table sum of step i under the accumulation of step i + 1: a second scope context on a second hipStream carries
    ncclAllReduce + unpack/clamp, events order it behind the accumulation that filled the buffer and hold the accumulation
    that will overwrite the buffer (two steps later) until the sum has left it.  Nothing is skipped, bucketed or deferred
    beyond the timed region: the closing synchronize waits for both streams."""

    def __init__(self, torch, ocm, device, compute_stream):
        self.torch, self.compute = torch, compute_stream
        self.stream = torch.cuda.Stream()
        self.ctx = ocm.ScopeContext(device)
        self.ctx.set_stream(self.stream.cuda_stream)
        self.filled = [torch.cuda.Event() for _ in range(2)]
        self.summed = [torch.cuda.Event() for _ in range(2)]
        self.used = [False, False]

    def before_compute(self, buf):
        if self.used[buf]:
            self.compute.wait_event(self.summed[buf])

    def after_compute(self, buf):
        self.filled[buf].record(self.compute)
        self.stream.wait_event(self.filled[buf])

    def after_reduce(self, buf):
        self.summed[buf].record(self.stream)
        self.used[buf] = True


def roofline(wl, algo_bytes, launches, kernel_ms, ms_per_step, traffic):
    kern_avg_ms = kernel_ms / launches if launches else float("nan")
    achieved = algo_bytes / (kern_avg_ms * 1e-3) / 1e9 if launches else float("nan")
    return {
        "bound": "hbm",
        "kernel": KERNEL[wl],
        "achieved": round(achieved, 1),
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBPS, 4),
        # the whole step (every kernel of the call + launch gaps) on the same scale: what the host sees
        "frac_whole_step": round(algo_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
        "traffic": traffic,
        "algorithmic_bytes_per_launch": algo_bytes,
        "avg_launch_ms": round(kern_avg_ms, 5),
        "launches_timed": launches,
    }


def code_only(src):
    """The source text without comments and without white space: an edit of a comment is not another kernel."""
    import re

    src = re.sub(rb"/\*.*?\*/", b"", src, flags=re.S)
    src = re.sub(rb"//[^\n]*", b"", src)
    return re.sub(rb"\s+", b"", src)


def kernel_fingerprint():
    """sha256 of the kernel sources the library was built from: PMC traffic files are only believed for this build."""
    hsh = hashlib.sha256()
    for f in ("scope_kernels.hip", "scope_kernels.h", "scope_vec.h", "scope_api.hip"):
        with open(os.path.join(ROOT, "obs-color-monitor_amd", "csrc", f), "rb") as fh:
            hsh.update(code_only(fh.read()))
    return hsh.hexdigest()[:16]


def pmc_traffic(wl, n, w, h, plain):
    """HBM bytes per launch from the PMC counters (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md), measured with
    rocprofv3 on this exact workload by tools/gpu_job.sh and committed with the fingerprint of the kernel sources.
    bench.py cannot run under a counter pass itself; a file measured on OTHER kernel sources is refused, loudly."""
    path = os.path.join(PROFILE_DIR, "pmc_traffic.json")
    try:
        pmc = json.load(open(path))
    except (OSError, ValueError):
        return None, "no profiles/round2/pmc_traffic.json"
    if pmc.get("kernel_sources_sha256_16") != kernel_fingerprint():
        log(f"bench.py: {path} was measured on other kernel sources ({pmc.get('kernel_sources_sha256_16')} != "
            f"{kernel_fingerprint()}): roofline.traffic is null until tools/gpu_job.sh traffic is re-run")
        return None, "stale: measured on other kernel sources"
    if not plain:
        return None, "non-default options"
    for e in pmc.get("entries", []):
        if (e["workload"], e["frames_per_step"], e["width"], e["height"]) == (wl, n, w, h):
            return e["bytes_per_launch"], "rocprofv3 PMC, " + pmc.get("job", "?")
    return None, "not measured for this workload"


def main():
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
    ap.add_argument("--workload", default="vectorscope",
                    choices=["vectorscope", "vectorscope_rgb", "waveform", "histogram", "three_scope", "stream"])
    ap.add_argument("--components", type=lambda s: int(s, 0), default=0x07)
    ap.add_argument("--placement-candidates", type=int, default=6,
                    help="frame pools allocated and probed before the fastest is kept (1 = take the first, as round 1 did)")
    ap.add_argument("--dist", default="uniform", choices=["uniform", "concentrated", "flat", "ramp", "natural", "screen"])
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="bound of the CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only: no variants / latency / stream / config4 legs")
    ap.add_argument("--strong", action="store_true",
                    help="N > 1: make BASELINE config 4 (one 7680x4320 batch, row tiles, strong scaling) the headline")
    ap.add_argument("--force-reduce", action="store_true",
                    help="run the RCCL table reduction even with one rank (exercises the N > 1 code path on a 1-GPU box)")
    ap.add_argument("--serial-reduce", action="store_true",
                    help="N > 1: enqueue the RCCL table sum on the accumulation's own stream (round 2's first form) instead of "
                         "overlapping step i's sum with step i+1's accumulation on a second stream")
    ap.add_argument("--torch-reduce", action="store_true", help="N > 1: sum the tables with torch.distributed (TableAllReduce) "
                    "instead of the C-ABI's scope_table_allreduce")
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
    import scope_oracle
    from obs_color_monitor_amd.synth import make_frames

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
    o = scope_oracle.COracle()
    plain = not args.option and args.dist == "uniform" and args.components == 0x07

    if args.workload == "stream":
        from obs_color_monitor_amd.stream import stream_leg

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
        os.write(real_stdout, (json.dumps(result) + "\n").encode())
        ctx.close()
        sys.exit(0 if res["parity"].startswith("bit-exact") else 3)

    # ---- synthetic input, resident in HBM ------------------------------------------------------
    w, h = (W4K, H4K) if args.size == "4k" else (2 * W4K, 2 * H4K)
    n = args.frames or (64 if args.size == "4k" else 16)
    pool = args.pool or (n if n >= 24 else (24 // n) * n)  # small launches rotate over >= 24 frames (0.8 GB > the Infinity Cache)
    wl = args.workload
    rgb, yuv = make_frames(torch, ctx, pool, w, h, args.dist, 0x9E3779B9 + rank)
    torch.cuda.synchronize()
    placement = None
    if args.placement_candidates > 1:
        pn = min(pool, 64)  # the probe reads the whole pool (one launch takes at most 64 frames)
        ptab = torch.empty((pn, 65536), dtype=torch.uint8, device="cuda")

        def probe_yuv(t):
            b = ctx.make_device_batch(t[:pn], 4 * w, w, h, planes_are_rgb=False)
            return lambda: ctx.vectorscope_batch_dev(b, ptab, sync=False)

        def probe_rgb(t):
            b = ctx.make_device_batch(t[:pn], 4 * w, w, h, planes_are_rgb=True, colorspace=2)
            return lambda: ctx.histogram_counts_batch_dev(b, 0x07, ptab.view(-1)[:pn * 4096], sync=False)

        # "as fast as a well-placed pool": 0.84 of the HBM peak for the probe's bytes (the kernels reach 0.85 there)
        # (full-size launches of the headline's kernel only: a probe of fewer than 64 frames does not reach that rate anywhere)
        good = (pn * (4 * w * h + 65536) / (0.84 * HBM_PEAK_GBPS * 1e9) * 1e6) if pn == 64 else 0.0
        yuv, us_y, ky = place_pool(torch, ctx, yuv, probe_yuv, args.placement_candidates, good)
        rgb, us_r, kr = place_pool(torch, ctx, rgb, probe_rgb, args.placement_candidates)
        placement = {"candidates": args.placement_candidates,
                     "yuv_pool_vectorscope_kernel_us": us_y, "yuv_pool_kept": ky,
                     "rgb_pool_histogram_kernel_us": us_r, "rgb_pool_kept": kr,
                     "what": "copies of the frame pool allocated one after the other, the accumulation kernel timed on each "
                             "(own begin/end stamps), the fastest kept, the others freed; as many again when none reads at 0.84 of "
                             "peak: untimed set-up (DESIGN.md section 5)"}
        del ptab
    work = Workload(torch, ctx, wl, rgb, yuv, w, h, n, args.components)
    table_bytes = work.tbytes

    # ---- N > 1: the table sum over RCCL -----------------------------------------------------------------------------
    reducer, comm, reduce_path, overlap = None, None, None, None
    wire = None
    if use_dist and wl in ("vectorscope", "vectorscope_rgb", "histogram"):
        if not args.torch_reduce:
            try:  # every rank must agree on the path: the verdict is all-reduced before anybody uses it
                uid = [ctx.rccl_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                comm = ctx.rccl_comm_create(uid[0], world, rank)
                ok = 1
            except Exception as e:  # noqa: BLE001
                log(f"rank {rank}: scope_rccl_comm_create failed ({e}); falling back to torch.distributed")
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if not int(flag.item()):
                if comm:
                    ctx.rccl_comm_destroy(comm)
                comm = None
        if comm:
            wire = [torch.empty((n, 32768 if wl != "histogram" else 1024), dtype=torch.int32, device="cuda") for _ in range(2)]
            if args.serial_reduce:
                reduce_path = "C-ABI: scope_table_allreduce (ncclAllReduce uint32 sum) every step, unpack/clamp on the same stream"
                overlap = None
            else:
                reduce_path = ("C-ABI: scope_table_allreduce (ncclAllReduce uint32 sum) + unpack/clamp of EVERY step on a second "
                               "hipStream: step i's table sum runs under step i+1's accumulation (two table buffers)")
                overlap = Overlap(torch, ocm, local_rank, stream)
        else:
            from obs_color_monitor_amd.row_tiles import TableAllReduce

            # one collective per 8 steps: c10d's per-collective host cost is about a step's GPU time
            reducer = TableAllReduce(ctx, n * table_bytes, "cuda", "sum_u32" if wl == "histogram" else "sat_u8", bucket=8)
            reduce_path = "torch.distributed: TableAllReduce, 8 steps per collective, side stream"
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

    def drain():
        if reducer is not None:
            result_tables[0] = reducer.drain()

    def step_and_drain(i):
        step(i)

    elapsed, sustained, launches, kernel_ms = timed_run(torch, ctx, stream, step_and_drain, args.steps, args.warmup,
                                                        args.settle_seconds, args.min_seconds, dist if use_dist else None)
    drain()
    torch.cuda.synchronize()
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- parity of what was just computed (untimed) ---------------------------------------------
    if comm or reducer is not None:
        # the reduced table of the LAST step: every rank's oracle table of its own tile, summed, clamped once
        g, buf = work.last
        final = result_tables[0].view(torch.uint8).view(n, table_bytes)
        parity = True
        for f in sorted({0, n - 1}):
            pr, py = rgb[g * n + f].cpu().numpy(), yuv[g * n + f].cpu().numpy()
            if wl == "histogram":
                mine = o.histogram_counts(pr, py, w, args.components).astype(np.int64)
            else:
                mine = o.vectorscope(py, w).astype(np.int64)
            want = torch.from_numpy(mine.reshape(-1)).cuda()
            dist.all_reduce(want)
            if wl == "histogram":
                got = final[f].view(torch.int32).to(torch.int64) & 0xFFFFFFFF
            else:
                want = want.clamp(max=255)
                got = final[f].to(torch.int64)
            parity = parity and bool(torch.equal(got, want))
    else:
        work.step(args.steps - 1)
        torch.cuda.synchronize()
        parity = work.check(o, rgb, yuv, np)
    if use_dist:  # the line rank 0 prints speaks for every rank
        okf = torch.tensor([1 if parity else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(okf, op=dist.ReduceOp.MIN)
        parity = bool(okf.item())

    # ---- the numbers ------------------------------------------------------------------------------
    pix_per_step = world * n * w * h
    value = pix_per_step * args.steps / elapsed / 1e9
    ms_per_step = elapsed / args.steps * 1e3
    traffic, traffic_src = pmc_traffic(wl, n, w, h, plain and world == 1)
    rf = roofline(wl, work.algo_bytes, launches, kernel_ms, ms_per_step, traffic)
    rf["traffic_source"] = traffic_src
    sustained["value"] = round(pix_per_step / (sustained["ms_per_step"] * 1e-3) / 1e9, 3)
    result = {
        "metric": "Gpix/s scope accumulation @3840×2160; achieved HBM GB/s vs peak",
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
        "parity": ("bit-exact vs oracle" if parity else "MISMATCH vs oracle") + " (parity unpinned: the oracle is this build's "
                  "restatement of the reference loops, DESIGN.md section 4)",
        "config": {
            "workload": WORKLOAD_TEXT[wl] + (f", components {args.components:#04x}" if wl in ("waveform", "histogram", "three_scope") else "")
                        + f", {w}x{h} BGRA8, {n} frame(s) per launch from {pool} distinct HBM-resident frames, {args.dist} RGB through K0",
            "frames_per_step": n,
            "frame_pool": pool,
            "width": w,
            "height": h,
            "input_distribution": args.dist,
            "partition": "one frame batch per GPU" if world == 1 else f"{world} row tiles of {w}x{h} + RCCL sum of tables",
            "frame_pool_placement": placement,
            "reduce": reduce_path,
            "device": info["name"],
            "compute_units": info["compute_units"],
            "options": args.option,
            "kernel_sources_sha256_16": kernel_fingerprint(),
        },
        "roofline": rf,
        "sustained": sustained,
    }

    # ---- the other legs (rank 0 of a one-GPU run; short, >= 0.5 s each) ---------------------------------------------
    extras = not args.no_extras and w == W4K
    if extras and world == 1 and not use_dist:
        del work
        variants = {}
        for v in ("vectorscope_rgb", "waveform", "histogram", "three_scope"):
            if v == wl:
                continue
            vw = Workload(torch, ctx, v, rgb, yuv, w, h, n, 0x07)
            steps_v = max(4, min(args.steps, 40))
            el, sus, ln, kms = timed_run(torch, ctx, stream, vw.step, steps_v, 3, 0.2, 0.5)
            ok = vw.check(o, rgb, yuv, np, frames=[0])
            tr, _ = pmc_traffic(v, n, w, h, True)
            vr = roofline(v, vw.algo_bytes, ln, kms, sus["ms_per_step"], tr)
            if v == "three_scope":  # two accumulation kernels per step: the event average is per kernel, the step has both
                vr["achieved"] = round(vw.algo_bytes / (sus["ms_per_step"] * 1e-3) / 1e9, 1)
                vr["frac"] = vr["frac_whole_step"]
                vr["note"] = "achieved = SURVEY 8(d) config-5 bytes (one read of the frames + the three tables) / whole step"
            variants[v] = {"value": round(n * w * h / (sus["ms_per_step"] * 1e-3) / 1e9, 3), "unit": "Gpix/s",
                           "ms_per_step": sus["ms_per_step"], "steps": sus["steps"], "parity": bool(ok), "roofline": vr}
            del vw
        result["variants"] = variants
        # the reference's call shape: one surface per callback, ring of three
        latency = {}
        for nf in (1, 3):
            for v in ("vectorscope", "vectorscope_rgb", "histogram", "waveform", "three_scope"):
                lw = Workload(torch, ctx, v, rgb[:24], yuv[:24], w, h, nf, 0x07)
                el, sus, ln, kms = timed_run(torch, ctx, stream, lw.step, 200, 50, 0.15, 0.3)
                lw.step(0)
                torch.cuda.synchronize()
                ok = lw.check(o, rgb, yuv, np, frames=[nf - 1])
                us = sus["ms_per_step"] * 1e3
                latency[f"{v}_x{nf}"] = {
                    "frames_per_call": nf, "us_per_call": round(us, 2), "us_host_enqueue_per_call": round(sus["host_enqueue_ms_per_step"] * 1e3, 2),
                    "us_accumulation_kernel": round(kms / ln * 1e3, 2) if ln else None,
                    "gbps": round(lw.algo_bytes / (us * 1e-6) / 1e9, 1), "frac_of_peak": round(lw.algo_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                    "parity": bool(ok)}
                del lw
        latency["note"] = ("us_per_call = stream time per call in steady state, calls back to back on one stream over 24 rotating "
                           "frames: every kernel of the call (accumulation + partial-table reduction) and the gaps between them, no "
                           "event brackets in between; us_accumulation_kernel comes from a separate bracketed leg and carries "
                           "~2-3 us of event overhead at this scale (rocprofv3 --stats of the same launches: profiles/round2/)")
        result["latency"] = latency
        try:
            from obs_color_monitor_amd.stream import stream_leg

            result["stream"] = stream_leg(torch, np, ctx, o, seconds=1.5, dist_kind="natural")
        except Exception as e:  # noqa: BLE001
            result["stream"] = {"error": repr(e)}

    # ---- BASELINE config 4 as written: one 7680x4320 batch in N row tiles, strong scaling ---------------------------
    if (extras or args.strong) and use_dist and comm and wl in ("vectorscope", "vectorscope_rgb"):
        try:
            result["config4"] = config4_leg(torch, dist, np, ctx, o, comm, rank, world, stream, overlap=overlap)
        except Exception as e:  # noqa: BLE001
            result["config4"] = {"error": repr(e)}
        if args.strong and "error" not in result["config4"]:
            c4 = result["config4"]
            result.update({"value": c4["value"], "ms_per_step": c4["ms_per_step"], "scaling": "strong"})
            result["config"]["partition"] = c4["partition"]

    # ---- CPU baseline: the oracle (port of the reference loop) on this host, bounded sample --------
    if rank == 0 and world == 1 and not args.no_cpu:
        result["cpu_baseline"] = cpu_baseline(np, o, "vectorscope" if wl == "three_scope" else wl, rgb[0].cpu().numpy(),
                                              yuv[0].cpu().numpy(), w, h, args.components, args.cpu_seconds)

    if rank == 0:
        os.write(real_stdout, (json.dumps(result) + "\n").encode())
    if comm:
        torch.cuda.synchronize()
        ctx.rccl_comm_destroy(comm)
    if overlap:
        overlap.ctx.close()
    ctx.close()
    if use_dist:
        dist.destroy_process_group()
    if not parity:
        sys.exit(3)


def config4_leg(torch, dist, np, ctx, o, comm, rank, world, stream, frames=16, seconds=0.6, overlap=None):
    """ONE batch of 7680x4320 frames; rank r owns rows [4320 r/N, 4320 (r+1)/N) of every frame; per step every rank
    reduces its tiles, the packed tables are summed over RCCL and clamped.  Two figures: throughput with the steps
    enqueued back to back, and the latency of one step with its reduce waited for (nothing bucketed or deferred)."""
    from obs_color_monitor_amd.row_tiles import tile_rows
    from obs_color_monitor_amd.synth import make_frames

    w, h = 7680, 4320
    r0, r1 = tile_rows(h, rank, world)
    th = r1 - r0
    # every rank generates ITS rows of the same frames: the seed is shared, the generator is advanced past other ranks' rows
    # (cheaper than generating whole 8K frames on every GPU: a tile is all a rank ever reads)
    rgb, yuv = make_frames(torch, ctx, frames, w, th, "uniform", 0xC0FFEE + rank)
    batch = ctx.make_device_batch(yuv, 4 * w, w, th, planes_are_rgb=False)
    wires = [torch.empty((frames, 32768), dtype=torch.int32, device="cuda") for _ in range(2)]
    outs = [torch.empty((frames, 65536), dtype=torch.uint8, device="cuda") for _ in range(2)]
    wire, out = wires[0], outs[0]
    count = [0]

    def step():
        """steps enqueued back to back: step i's table sum overlaps step i + 1's accumulation (`overlap`), or follows it on
        the same stream"""
        buf = count[0] & 1
        count[0] += 1
        rctx = ctx
        if overlap:
            overlap.before_compute(buf)
            rctx = overlap.ctx
        ctx.vectorscope_batch_dev(batch, wires[buf], sync=False, packed16=True)
        if overlap:
            overlap.after_compute(buf)
        rctx.table_allreduce(comm, wires[buf], frames * 32768)
        rctx.unpack16_clamp_dev(wires[buf], outs[buf], frames * 32768)
        if overlap:
            overlap.after_reduce(buf)

    def step_waited():
        ctx.vectorscope_batch_dev(batch, wire, sync=False, packed16=True)
        ctx.table_allreduce(comm, wire, frames * 32768)
        ctx.unpack16_clamp_dev(wire, out, frames * 32768)

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    # every rank runs the SAME number of steps (a step holds a collective): fixed chunks, and the ranks agree after each
    # whether to go on; the agreement drains the stream, so the chunk is long (tens of milliseconds) against that pause
    t0 = time.perf_counter()
    steps, chunk = 0, 40
    while True:
        for _ in range(chunk):
            step()
        steps += chunk
        if not agree(torch, dist, time.perf_counter() - t0 < seconds):
            break
    torch.cuda.synchronize()
    dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    ms = float(el.item()) / steps * 1e3
    lat = []
    for _ in range(30):
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step_waited()
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t1)
    lat_t = torch.tensor([sorted(lat)[len(lat) // 2]], dtype=torch.float64, device="cuda")
    dist.all_reduce(lat_t, op=dist.ReduceOp.MAX)
    # parity of frame 0: the sum of every rank's oracle table of its tile, clamped once
    mine = torch.from_numpy(o.vectorscope(yuv[0].cpu().numpy(), w).astype(np.int64).reshape(-1)).cuda()
    dist.all_reduce(mine)
    ok = torch.tensor([1 if torch.equal(out[0].to(torch.int64), mine.clamp(max=255)) else 0], dtype=torch.int32, device="cuda")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    return {
        "value": round(frames * w * h / (ms * 1e-3) / 1e9, 3), "unit": "Gpix/s", "scaling": "strong", "ms_per_step": round(ms, 4),
        "step_latency_ms_reduce_waited": round(float(lat_t.item()) * 1e3, 4),
        "partition": f"{frames} frames of {w}x{h}, {world} row tiles of {w}x{th} per frame + RCCL sum of 256x256 tables (C-ABI)",
        "frames_per_step": frames, "parity": bool(ok.item()),
    }


def cpu_baseline(np, o, wl, plane_rgb, plane_yuv, w, h, components, seconds):
    def one():
        if wl in ("vectorscope", "vectorscope_rgb"):
            return o.vectorscope(plane_yuv, w)
        if wl == "waveform":
            return o.waveform(plane_rgb, plane_yuv, w, components)
        return o.histogram_counts(plane_rgb, plane_yuv, w, components)

    t_end = time.perf_counter() + seconds
    reps, t_start = 0, time.perf_counter()
    while time.perf_counter() < t_end or reps < 3:
        one()
        reps += 1
    cpu_t = time.perf_counter() - t_start
    res = {
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
            return o.waveform_mt(plane_rgb, plane_yuv, w, components, nthreads, reps)
        return o.histogram_counts_mt(plane_rgb, plane_yuv, w, components, nthreads, reps)

    mt_same = bool(np.array_equal(all_cores(threads).astype(np.int64).reshape(-1), one().astype(np.int64).reshape(-1)))
    mt_reps = 20  # per call; thread start-up is paid once per call
    best = None
    for nthreads in sorted({t for t in (16, 32, 64, 128, 256, threads) if t <= threads}):
        all_cores(nthreads, 2)
        t_start, calls = time.perf_counter(), 0
        while time.perf_counter() - t_start < min(seconds, 10.0) / 8 or calls < 1:
            all_cores(nthreads, mt_reps)
            calls += 1
        rate = calls * mt_reps * w * h / (time.perf_counter() - t_start) / 1e9
        if best is None or rate > best[0]:
            best = (rate, nthreads, calls)
    res["all_cores"] = {
        "value": round(best[0], 3), "unit": "Gpix/s", "cores": best[1], "host_cores": threads, "cpu": cpu_model,
        "same_table": mt_same,
        "sample": f"best of 16..{threads} POSIX threads: {best[2]} calls x {mt_reps} passes over frame 0 (row tiles with "
                  "private tables; column tiles for the waveform; thread start-up once per call)",
    }
    return res


if __name__ == "__main__":
    main()
