"""bench_core.py -- what every part of bench.py shares: the constants of the metric, the device plumbing (the real thing is
torch.cuda + libscope_hip.so; `--backend emul` is the CPU emulation build of the same sources, tests/emul, so that the N-rank
logic can run without a GPU), the gloo control plane (class Ctl), one scope workload over a pool of HBM-resident frames
(class Workload), the timed run, the roofline object, the flat summary, the sources fingerprint and the PMC traffic look-up.
Nothing here is the product; nothing here calls the oracle except Workload.check (the checker, never inside a timed region)."""
import hashlib
import json
import os
import tempfile
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
W4K, H4K = 3840, 2160
# where pmc_traffic.json is looked for: BENCH_PMC_DIR (an evidence visit points it at the directory its own PMC passes just
# wrote, tools/visit_full.sh), then the committed rounds, newest first
PROFILE_DIRS = ([os.environ["BENCH_PMC_DIR"]] if os.environ.get("BENCH_PMC_DIR") else []) + \
               [os.path.join(ROOT, "profiles", r) for r in ("round6", "round5", "round4", "round3", "round2")]
METRIC = "Gpix/s scope accumulation @3840×2160; achieved HBM GB/s vs peak"
UNPINNED = " (parity unpinned: the oracle is this build's restatement of the reference loops, DESIGN.md section 4)"

WORKLOAD_TEXT = {
    "vectorscope": "vectorscope 256x256 CbCr scatter (K1) over staged Y'CbCr planes",
    "vectorscope_rgb": "vectorscope with fused K0 over RGB planes",
    "waveform": "waveform per-column 256-bin (K2)",
    "histogram": "histogram 3x256 u32 (K3)",
    "three_scope": "vectorscope (fused K0) + waveform + histogram of every frame from ONE read of its plane (scope_three_scopes_batch)",
}
KERNEL = {"vectorscope": "scope_vss_kernel", "vectorscope_rgb": "scope_vss_kernel", "waveform": "scope_wvs_kernel",
          "histogram": "scope_his_kernel", "three_scope": "scope_trio_kernel (one read; two passes = scope_vss_kernel + scope_wvs_kernel with dock=0)"}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def table_bytes_of(wl, w):
    return {"vectorscope": 65536, "vectorscope_rgb": 65536, "waveform": 256 * w * 4, "histogram": 4096,
            "three_scope": 65536 + 256 * w * 4 + 4096}[wl]


class HostEvent:
    def __init__(self):
        self.t = 0.0

    def record(self, stream=None):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class HostStream:
    cuda_stream = None

    def wait_event(self, ev):
        pass


class Dev:
    def __init__(self, torch, args, local_rank):
        self.torch, self.emul = torch, args.backend == "emul"
        if self.emul:
            self.device, self.index, self.shared = "cpu", 0, False
            return
        have = torch.cuda.device_count()
        if have < 1:
            raise RuntimeError("no HIP device is visible: the scope path has no CPU implementation (--backend emul is the "
                               "test-only emulation build)")
        if local_rank >= have and not args.oversubscribe:
            raise RuntimeError(f"rank {local_rank} has no GPU of its own ({have} visible); --oversubscribe lets ranks share one")
        self.index = local_rank % have
        self.shared = int(os.environ.get("WORLD_SIZE", "1")) > have
        self.device = "cuda"
        torch.cuda.set_device(self.index)

    def sync(self):
        if not self.emul:
            self.torch.cuda.synchronize()

    def stream(self):
        return HostStream() if self.emul else self.torch.cuda.Stream()

    def event(self, timing=False):
        return HostEvent() if self.emul else self.torch.cuda.Event(enable_timing=timing)

    def empty_cache(self):
        if not self.emul:
            self.torch.cuda.empty_cache()


class Ctl:
    """Control plane of an N-rank run: barriers, agreement flags, max / sum over ranks -- gloo on CPU tensors, so that
    nothing of it is ever queued on a GPU beside the data path's collectives."""

    def __init__(self, torch, dist, world):
        self.torch, self.dist, self.world = torch, dist, world

    def _reduce(self, values, op, dtype):
        if self.dist is None:
            return list(values)
        t = self.torch.tensor(list(values), dtype=dtype)
        self.dist.all_reduce(t, op=op)
        return t.tolist()

    def barrier(self):
        if self.dist is not None:
            self._reduce([1], self.dist.ReduceOp.SUM, self.torch.int32)

    def max(self, x):
        return self._reduce([x], self.dist.ReduceOp.MAX if self.dist else None, self.torch.float64)[0]

    def all_true(self, flag):
        return bool(self._reduce([1 if flag else 0], self.dist.ReduceOp.MIN if self.dist else None, self.torch.int32)[0])

    def any_true(self, flag):
        return bool(self._reduce([1 if flag else 0], self.dist.ReduceOp.MAX if self.dist else None, self.torch.int32)[0])

    def sum_array(self, arr):
        """int64 numpy array summed over ranks (oracle tables of the ranks' tiles)."""
        if self.dist is None:
            return arr
        t = self.torch.from_numpy(arr.copy())
        self.dist.all_reduce(t)
        return t.numpy()

    def bcast(self, obj):
        if self.dist is None:
            return obj
        box = [obj]
        self.dist.broadcast_object_list(box, src=0)
        return box[0]


def spread(n, k):
    """k frame indices spread over a launch of n frames, first and last included."""
    return sorted({round(i * (n - 1) / max(k - 1, 1)) for i in range(min(k, n))})


class Workload:
    """One scope workload over a pool of HBM-resident frames, `n` frames per step, rotating through the pool."""

    def __init__(self, torch, ctx, wl, rgb, yuv, w, h, n, components=0x07, side_reduce=False):
        self.torch, self.ctx, self.wl, self.w, self.h, self.n, self.comp = torch, ctx, wl, w, h, n, components
        self.side = side_reduce  # vectorscope only: SCOPE_SIDE_REDUCE, the reduction of call i under the accumulation of call i + 1
        pool = rgb.shape[0]
        dev = rgb.device
        self.groups = pool // n
        src = yuv if wl == "vectorscope" else rgb
        self.linesize = int(src.shape[2])
        self.batches = [ctx.make_device_batch(src[g * n:(g + 1) * n], self.linesize, w, h, planes_are_rgb=(wl != "vectorscope"), colorspace=2)
                        for g in range(self.groups)]
        self.tbytes = table_bytes_of(wl, w)
        if wl == "three_scope":
            self.vss = [torch.empty((n, 65536), dtype=torch.uint8, device=dev) for _ in range(2)]
            self.wvs = [torch.empty((n, 256 * w * 4), dtype=torch.uint8, device=dev) for _ in range(2)]
            self.his = [torch.empty((n, 1024), dtype=torch.int32, device=dev) for _ in range(2)]
        else:
            self.tables = [torch.empty((n, self.tbytes), dtype=torch.uint8, device=dev) for _ in range(2)]
        self.last = (0, 0)
        # 4 B per pixel read once + the tables once per frame
        self.algo_bytes = n * (4 * w * h + self.tbytes)

    def step(self, i):
        ctx, b, buf = self.ctx, self.batches[i % self.groups], i & 1
        wl = self.wl
        if wl in ("vectorscope", "vectorscope_rgb"):
            ctx.vectorscope_batch_dev(b, self.tables[buf], sync=False, side_reduce=self.side)
        elif wl == "waveform":
            ctx.waveform_batch_dev(b, self.comp, self.tables[buf], sync=False)
        elif wl == "histogram":
            ctx.histogram_counts_batch_dev(b, self.comp, self.tables[buf], sync=False)
        else:
            ctx.three_scopes_batch_dev(b, self.comp, self.comp, self.vss[buf], self.wvs[buf], self.his[buf], sync=False)
        self.last = (i % self.groups, buf)

    def finish(self):
        """What a host does before it reads a table of a side_reduce call: the context's stream waits for the side stream."""
        if self.side:
            self.ctx.fence()

    def check(self, o, rgb, yuv, np, frames=None):
        """Tables of the most recent step against the oracle (frames: indices inside the step's batch)."""
        g, buf = self.last
        torch = self.torch
        ok = True
        for f in (frames if frames is not None else spread(self.n, 8)):
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


def timed_run(dev, ctl, ctx, stream, step, steps, warmup, settle_s, min_s, multi=False, finish=None):
    """settle -> warm-up -> EXACTLY `steps` steps between barrier + synchronize -> at least min_s more of the same ->
    a short leg with the accumulation kernel bracketed by its own begin/end stamps.
    Loops that run 'for so many seconds' run the SAME number of steps on every rank (a step may hold a collective, and a
    rank that enqueues one more than its peers waits for ever): no single rank's clock decides, the loop goes on while
    ANY rank wants to.  Returns (elapsed of the K steps, sustained dict, bracketed launches, their summed ms)."""
    def go_on(flag):
        if not multi:
            return flag
        dev.sync()
        return ctl.any_true(flag)

    t_end = time.perf_counter() + settle_s
    i = 0
    while go_on(time.perf_counter() < t_end):
        for _ in range(8):
            step(i)
            i += 1
        dev.sync()
    for j in range(warmup):
        step(j)
    dev.sync()
    ctl.barrier()
    dev.sync()
    t0 = time.perf_counter()
    for j in range(steps):
        step(j)
    if finish:
        finish()
    dev.sync()
    ctl.barrier()
    elapsed = time.perf_counter() - t0
    # the sustained leg: the same step for >= min_s; the host only enqueues (its own pace is reported beside the GPU's)
    s_steps, e0, e1 = 0, dev.event(True), dev.event(True)
    chunk = max(8, steps)
    t1 = time.perf_counter()
    e0.record(stream)
    while True:
        for j in range(chunk):
            step(s_steps + j)
        s_steps += chunk
        if not go_on(time.perf_counter() - t1 < min_s):
            break
    host_s = time.perf_counter() - t1
    if finish:
        finish()
    e1.record(stream)
    dev.sync()
    sustained = {"seconds": round(time.perf_counter() - t1, 3), "steps": s_steps,
                 "ms_per_step": round(e0.elapsed_time(e1) / s_steps, 5),
                 "host_enqueue_ms_per_step": round(host_s / s_steps * 1e3, 5)}
    ctx.profile(True)
    for j in range(min(400, max(60, steps)) if not dev.emul else 2):
        step(j)
    dev.sync()
    launches, kernel_ms = ctx.kernel_ms()
    ctx.profile(False)
    return elapsed, sustained, launches, kernel_ms


def kernel_us(ctx, fn, warm=12, reps=24):
    """Average duration of the accumulation kernel fn() launches: the kernel's own begin / end stamps."""
    for _ in range(warm):
        fn()
    ctx.sync()
    ctx.profile(True)
    for _ in range(reps):
        fn()
    ctx.sync()
    nl, ms = ctx.kernel_ms()
    ctx.profile(False)
    return round(ms / max(nl, 1) * 1e3, 2)


def steps_ms(dev, ctl, step, steps, warmup):
    """ms per step of exactly `steps` steps between synchronizes, after `warmup` untimed ones (one round of an interleaved comparison)."""
    for j in range(max(warmup, 2)):
        step(j)
    dev.sync()
    ctl.barrier()
    t0 = time.perf_counter()
    for j in range(steps):
        step(j)
    dev.sync()
    return (time.perf_counter() - t0) / steps * 1e3


def place_pool(torch, dev, ctx, frames, candidates, keep_alive):
    """WHERE a frame pool lies in HBM decides a few per cent of every streaming kernel's rate, and the first pools a
    process allocates are regularly the slow ones (profiles/round2/placement_probe_g8.txt; profiles/HISTORY.md section 5).  The
    product offers a host that keeps a long-lived pool a way to look before it settles -- scope_frame_pool_alloc:
    `candidates` allocations held at once, each read with the accumulation kernel timed by its own stamps, the fastest
    kept -- and this bench is such a host: the frames move into a pool the LIBRARY placed.  Untimed set-up, rank-local.
    Returns (the frames in the placed pool, read GB/s per candidate with the kept one first)."""
    pool, rates = ctx.frame_pool_alloc(frames.numel(), candidates)
    keep_alive.append(pool)
    placed = pool.tensor(torch, tuple(frames.shape))
    placed.copy_(frames)
    dev.sync()
    return placed, rates


class Overlap:
    """The table sum of step i under the accumulation of step i + 1: a second scope context on a second hipStream carries
    ncclAllReduce + unpack/clamp, events order it behind the accumulation that filled the buffer and hold the accumulation
    that will overwrite the buffer (two steps later) until the sum has left it.  Nothing is skipped, bucketed or deferred
    beyond the timed region: the closing synchronize waits for both streams."""

    def __init__(self, dev, ocm, lib_path, compute_stream):
        self.compute = compute_stream
        self.stream = dev.stream()
        self.ctx = ocm.ScopeContext(dev.index, lib_path)
        if not dev.emul:
            self.ctx.set_stream(self.stream.cuda_stream)
        self.filled = [dev.event() for _ in range(2)]
        self.summed = [dev.event() for _ in range(2)]
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
    """`achieved` / `frac` are the WHOLE STEP's: algorithmic bytes over everything the call enqueues (the accumulation kernel, the
    partial-table reduction, the gaps between launches) as the K timed steps saw it -- the conservative figure, and the one an
    outside clock around the run reproduces.  The accumulation kernel alone (HIP events on its stream around each launch, the
    figure rocprofv3's average duration for that kernel agrees with) is `achieved_kernel` / `frac_kernel`."""
    kern_avg_ms = kernel_ms / launches if launches else float("nan")
    kern = algo_bytes / (kern_avg_ms * 1e-3) / 1e9 if launches and kern_avg_ms > 0 else float("nan")
    whole = algo_bytes / (ms_per_step * 1e-3) / 1e9 if ms_per_step > 0 else float("nan")
    return {
        "bound": "hbm",
        "kernel": KERNEL[wl],
        "achieved": round(whole, 1),
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": round(whole / HBM_PEAK_GBPS, 4),
        "frac_is": "whole step (every kernel of the call + launch gaps) over the K timed steps",
        "achieved_kernel": round(kern, 1),
        "frac_kernel": round(kern / HBM_PEAK_GBPS, 4),
        "frac_whole_step": round(whole / HBM_PEAK_GBPS, 4),
        "traffic": traffic,
        "algorithmic_bytes_per_launch": algo_bytes,
        "avg_launch_ms": round(kern_avg_ms, 5),
        "launches_timed": launches,
    }


def flat_summary(result):
    """{name: scalar} of the other legs: <variant>_gpix / _frac (whole step) / _frac_kernel, <workload>_x1_us / _x3_us (the
    reference's call shape), <content>_<workload>_frac_kernel, stream / config 1 / config 4 figures, and all_parity."""
    flat, ok = {}, result["parity"].startswith("bit-exact")
    for name, v in (result.get("variants") or {}).items():
        flat[f"{name}_gpix"] = v["value"]
        flat[f"{name}_frac"] = v["roofline"]["frac"]
        flat[f"{name}_frac_kernel"] = v["roofline"]["frac_kernel"]
        ok = ok and v["parity"] is not False
    for name, v in (result.get("latency") or {}).items():
        if isinstance(v, dict):
            flat[f"{name}_us"] = v["us_per_call"]
            ok = ok and bool(v["parity"])
    for kind, v in (result.get("distributions") or {}).items():
        if not isinstance(v, dict):
            continue
        for wl, r in v.items():
            if isinstance(r, dict):
                if "frac_kernel" in r:
                    flat[f"{kind}_{wl}_frac_kernel"] = r["frac_kernel"]
                if "frac" in r:
                    flat[f"{kind}_{wl}_frac"] = r["frac"]
                flat[f"{kind}_{wl}_gpix"] = r.get("gpix_per_s")
                ok = ok and r.get("parity") is not False
    st = result.get("stream") or {}
    if "flat_out" in st:
        flat["stream_fps"] = st["flat_out"].get("fps")
        flat["stream_paced_240fps_latency_ms"] = (st.get("paced_240fps") or {}).get("latency_ms_mean")
        ok = ok and str(st.get("parity", "")).startswith("bit-exact")
    c1 = result.get("config1") or {}
    if "ms_per_frame" in c1:
        flat["config1_cpu_ms_per_frame"] = c1["ms_per_frame"]
        ok = ok and bool(c1.get("gpu_tables_bit_exact"))
    c4 = result.get("config4") or {}
    if "value" in c4:
        flat["config4_gpix"] = c4["value"]
        flat["config4_ranks_in_communicator"] = c4.get("rccl_ranks_in_communicator")
        ok = ok and bool(c4.get("parity"))
    pl = (result.get("config") or {}).get("frame_pool_placement") or {}
    if "value" in pl:
        flat["placed_pool_gpix"] = pl["value"]
    ac = (result.get("cpu_baseline") or {}).get("all_cores") or {}
    if "value" in ac:
        flat["cpu_all_cores_gpix"] = ac["value"]
    flat["all_parity"] = bool(ok)
    return flat


# the flat scalars the ONE stdout line carries (<= 40, one copy, top level); everything else is in bench_detail.json
LINE_SCALARS = (
    "vectorscope_rgb_gpix", "vectorscope_rgb_frac_kernel", "waveform_gpix", "waveform_frac_kernel", "waveform_y_frac_kernel",
    "waveform_r_frac_kernel", "waveform_g_frac_kernel", "waveform_b_frac_kernel", "histogram_gpix", "histogram_frac_kernel",
    "three_scope_gpix", "three_scope_frac_kernel",
    "vectorscope_x1_us", "vectorscope_rgb_x1_us", "histogram_x1_us", "waveform_x1_us", "three_scope_x1_us", "vectorscope_x3_us",
    "three_scope_x3_us",
    "natural_three_scope_frac", "concentrated_three_scope_frac", "screen_three_scope_frac", "ramp_three_scope_frac",
    "natural_vectorscope_frac_kernel", "natural_vectorscope_rgb_frac_kernel", "screen_vectorscope_rgb_frac_kernel",
    "ramp_vectorscope_rgb_frac_kernel", "padded_pitch_vectorscope_frac_kernel", "padded_pitch_vectorscope_rgb_frac_kernel",
    "alpha_holes_histogram_frac_kernel", "alpha_holes_waveform_frac_kernel",
    "stream_fps", "stream_paced_240fps_latency_ms", "config1_cpu_ms_per_frame", "config4_gpix", "config4_ranks_in_communicator",
    "placed_pool_gpix", "cpu_all_cores_gpix", "all_parity",
)
LINE_LIMIT = 6000  # bytes; the driver's record has been seen to lose a 21.7 KB line and to keep only the last ~8 KB of stdout
DETAIL_FILE = "bench_detail.json"


def _short(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 3] + "..."


def _pick(d, keys):
    return {k: d[k] for k in keys if k in d}


def compact_launcher(lz):
    """The launcher's object for the one line: the mode and, per attempt, what was added, how it ended and how long it took."""
    return {"mode": _short(lz.get("mode"), 80), "deadline_s_per_attempt": lz.get("deadline_s_per_attempt"),
            "attempts": [dict(_pick(a, ("extra_args", "exit_codes", "seconds")), failure=_short(a.get("failure"), 120))
                         for a in lz.get("attempts", [])]}


def compact_line(result, detail=None):
    """The ONE line that reaches stdout: the contract's scalars first, then `config`, `roofline`, `cpu_baseline` with the
    keys a reader recomputes from, at most 40 flat scalars of the other legs (once, at the top level), and -- at N > 1 or
    after --strong / --force-reduce -- `config4` and `config.rccl` with what a scaling record is judged from.  No prose, no
    nested legs: those are in the detail file this line names.  Always shorter than LINE_LIMIT bytes."""
    out = _pick(result, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                         "vs_baseline", "dtype", "data"))
    out["parity"] = _short(result.get("parity"), 48)
    if "error" in result:
        out["error"] = _short(result["error"], 200)
    cfg = result.get("config") or {}
    c = _pick(cfg, ("frames_per_step", "frame_pool", "width", "height", "input_distribution"))
    c = dict({"workload": _short(cfg.get("workload"), 160)}, **c)
    for k, n in (("partition", 120), ("reduce", 120), ("launch", 48), ("device", 48), ("library", 64)):
        if cfg.get(k) is not None:
            c[k] = _short(cfg[k], n)
    c.update(_pick(cfg, ("compute_units", "gpu_shared_by_ranks", "kernel_sources_sha256_16")))
    if cfg.get("options"):
        c["options"] = cfg["options"]
    c["rccl"] = None if not cfg.get("rccl") else _pick(cfg["rccl"], ("ranks_in_communicator", "rank", "version"))
    if str(cfg.get("backend", "")).startswith("EMULATION"):
        c["backend"] = "EMULATION (CPU): numbers are meaningless"
    out["config"] = c
    rf = result.get("roofline")
    if rf:
        r = _pick(rf, ("bound", "kernel", "achieved", "peak", "unit", "frac", "achieved_kernel", "frac_kernel", "avg_launch_ms",
                       "launches_timed", "algorithmic_bytes_per_launch", "traffic", "frac_placed", "frac_kernel_placed",
                        "frac_first_pool_same_rounds", "frac_kernel_first_pool_same_rounds", "placed_over_first_same_rounds",
                        "placed_over_first_same_rounds_kernel"))
        r["kernel"] = _short(r.get("kernel"), 40)
        r["traffic_source"] = _short(rf.get("traffic_source"), 60)
        out["roofline"] = r
    cb = result.get("cpu_baseline")
    if cb:
        b = _pick(cb, ("value", "unit", "cores", "kind"))
        ac = cb.get("all_cores") or {}
        b.update({"host_cores": ac.get("host_cores"), "cpu": _short(ac.get("cpu"), 48), "sample": _short(cb.get("sample"), 140)})
        out["cpu_baseline"] = b
    c4 = result.get("config4")
    if c4 and (result.get("n_gpus", 1) > 1 or result.get("scaling") == "strong" or "error" in c4):
        out["config4"] = (_pick(c4, ("value", "unit", "scaling", "ms_per_step", "frames_per_step", "parity", "rccl_ranks_in_communicator", "reduce_path",
                                     "ms_per_step_reduce_on_the_accumulation_stream", "ms_per_step_reduce_overlapped_second_stream",
                                     "step_latency_ms_reduce_waited", "rows_of_this_rank"))
                          if "error" not in c4 else {"error": _short(c4["error"], 160)})
        if "partition" in c4:
            out["config4"]["partition"] = _short(c4["partition"], 120)
    flat = flat_summary(result)
    for k in LINE_SCALARS:
        if flat.get(k) is not None:
            out[k] = flat[k]
    if "launcher" in result:
        out["launcher"] = compact_launcher(result["launcher"])
    if detail:
        out["detail"] = detail
    line = json.dumps(out)
    for k in reversed(LINE_SCALARS):  # never reached with today's keys; a guard, not a plan
        if len(line.encode()) < LINE_LIMIT:
            break
        if out.pop(k, None) is not None:
            line = json.dumps(out)
    return line


def write_detail(result, path=None):
    """Everything the legs measured (variants, latency, distributions, stream, config 1 / 4, the prose) as one JSON file beside
    bench.py, and its name for the line.  A directory that cannot be written costs the detail, not the line."""
    path = path or os.environ.get("BENCH_DETAIL")
    if not path:
        # an EMULATION run (tests, experiments on the CPU) never takes the name a real run's line points at
        emul = "EMULATION" in str((result.get("config") or {}).get("backend", ""))
        path = os.path.join(tempfile.gettempdir(), f"bench_detail_emul_{os.getpid()}.json") if emul else os.path.join(ROOT, DETAIL_FILE)
    try:
        # written beside and moved into place: a reader (or a second run with the same name) never sees half a file; the file
        # carries the line's own value / ms_per_step / fingerprint at its top level, so a line and a detail file that are not of
        # one run can be told apart
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "w") as fh:
            json.dump(result, fh, indent=1)
            fh.write("\n")
        os.replace(tmp, path)
        return os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
    except OSError as e:
        log(f"bench.py: detail file {path} not written ({e})")
        return None


def code_only(src):
    """The source text without comments and without white space: an edit of a comment is not another kernel."""
    import re

    src = re.sub(rb"/\*.*?\*/", b"", src, flags=re.S)
    src = re.sub(rb"//[^\n]*", b"", src)
    return re.sub(rb"\s+", b"", src)


def kernel_fingerprint():
    """sha256 of the kernel sources the library was built from: PMC traffic files are only believed for this build."""
    hsh = hashlib.sha256()
    for f in ("scope_kernels.hip", "scope_trio.hip", "scope_device.h", "scope_kernels.h", "scope_vec.h", "scope_api.hip"):
        with open(os.path.join(ROOT, "obs-color-monitor_amd", "csrc", f), "rb") as fh:
            hsh.update(code_only(fh.read()))
    return hsh.hexdigest()[:16]


def pmc_traffic(wl, n, w, h, plain):
    """HBM bytes per launch from the PMC counters (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md), measured with
    rocprofv3 on this exact workload by tools/gpu_job.sh and committed with the fingerprint of the kernel sources.
    bench.py cannot run under a counter pass itself; a file measured on OTHER kernel sources is refused, loudly."""
    pmc, path = None, None
    for d in PROFILE_DIRS:
        path = os.path.join(d, "pmc_traffic.json")
        try:
            pmc = json.load(open(path))
            break
        except (OSError, ValueError):
            continue
    if pmc is None:
        return None, "no profiles/round*/pmc_traffic.json"
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


def make_comm(ctx, ctl, world, rank):
    """A communicator for the C-ABI's table sum: rank 0's unique id travels over the control plane, every rank creates
    its end, and the verdict is agreed on (all ranks use it or none does).  Returns (comm or None, reason)."""
    uid, comm, why = None, None, ""
    if rank == 0:
        try:
            uid = ctx.rccl_unique_id()
        except Exception as e:  # noqa: BLE001
            why = repr(e)
    uid = ctl.bcast(uid)  # every rank takes part in every control-plane collective, whatever failed before it
    if uid is not None:
        try:
            comm = ctx.rccl_comm_create(uid, world, rank)
        except Exception as e:  # noqa: BLE001
            why = repr(e)
    else:
        why = why or "rank 0 could not make a unique id (no collective library)"
    if not ctl.all_true(comm is not None):
        if comm:
            ctx.rccl_comm_destroy(comm)
        return None, why or "a peer could not create its end"
    return comm, ""
