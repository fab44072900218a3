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
Every leg's headline number is repeated as a flat scalar at the top level and inside `roofline` (flat_summary()).

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

Rank 0 prints ONE JSON line (see README/DESIGN.md for the fields).
"""
import argparse
import faulthandler
import hashlib
import json
import os
import signal
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
W4K, H4K = 3840, 2160
PROFILE_DIRS = [os.path.join(ROOT, "profiles", r) for r in ("round4", "round3", "round2")]
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


# =====================================================================================================================
#  the launcher: `python bench.py --gpus N` with no WORLD_SIZE around starts N ranks itself
# =====================================================================================================================
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def kill_ranks(procs):
    """Ends exactly the processes this launcher started (each is the leader of its own process group)."""
    for p in procs:
        if p.poll() is None:
            try:
                os.killpg(p.pid, signal.SIGTERM)
            except (ProcessLookupError, PermissionError):
                pass
    t_end = time.time() + 5
    for p in procs:
        try:
            p.wait(timeout=max(0.1, t_end - time.time()))
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass
            p.wait()


def launch_once(argv, n, timeout_s):
    """N ranks of this script, one per GPU.  Returns (rank 0's last stdout line or None, outcome dict)."""
    import tempfile

    port = free_port()
    procs, t0 = [], time.time()
    out_file = tempfile.TemporaryFile()  # not a pipe: nobody has to drain it while the ranks run
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "4")
        env.setdefault("GLOO_SOCKET_IFNAME", "lo")  # one node: the control plane needs no resolvable hostname
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env, start_new_session=True,
                                      stdout=out_file if r == 0 else subprocess.DEVNULL))
    outcome = {"ranks_started": n, "rendezvous": f"127.0.0.1:{port}"}
    why = None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0, 3)]  # 3 = "ran, parity mismatch": every rank leaves with it
        if bad:
            why = f"rank {bad[0][0]} exited with code {bad[0][1]} while others were still running"
            break
        if time.time() - t0 > timeout_s:
            why = f"deadline of {timeout_s:.0f} s passed (a hung collective or rendezvous)"
            break
        time.sleep(0.05)
    if why:
        log(f"bench.py launcher: {why}; ending the ranks of this attempt")
        kill_ranks(procs)
    out_file.seek(0)
    out = out_file.read().decode(errors="replace")
    out_file.close()
    codes = [p.returncode for p in procs]
    outcome.update({"exit_codes": codes, "seconds": round(time.time() - t0, 1), "failure": why})
    line = None
    for ln in out.splitlines():
        if ln.startswith("{"):
            line = ln
    if not why and any(c not in (0, 3) for c in codes):
        outcome["failure"] = f"exit codes {codes}"
    return line, outcome


def reduce_ladder(args):
    """Rungs of extra arguments: what was asked for; the torch.distributed table sum; replicas with no table sum at all."""
    ladder = [[]]
    if not args.torch_reduce and not args.no_collective:
        ladder.append(["--torch-reduce"])
    if not args.no_collective:
        ladder.append(["--no-collective"])
    return ladder


def failure_line(args, attempts, mode):
    return json.dumps({
        "metric": METRIC, "value": 0.0, "unit": "Gpix/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "error": "no attempt of the N-rank run completed; see launcher.attempts (rank logs are on stderr)",
        "config": {"workload": "none completed"},
        "launcher": {"mode": mode, "attempts": attempts, "deadline_s_per_attempt": args.launch_timeout},
    })


def supervisor(args, argv):
    """Under `python -m torch.distributed.run ... bench.py --gpus N` every process torchrun starts is a SUPERVISOR: it runs
    the rank's work in a child process (same RANK / LOCAL_RANK / WORLD_SIZE, a rendezvous port of the attempt's own) against
    a deadline, and the supervisors -- a gloo group of their own on the rendezvous torchrun gave them, CPU only -- agree
    after every attempt whether it worked on ALL ranks.  So the fallback ladder of the self-launched form (C-ABI RCCL ->
    torch.distributed -> replicas) exists here too, and a hung collective still ends in rank 0's one JSON line."""
    import datetime
    import tempfile

    # ONE JSON line on stdout: gloo announces its connections through C stdio on file descriptor 1, so that descriptor points
    # at stderr while this process lives and the line goes to the saved, real stdout
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=args.launch_timeout * 3 + 600))
    attempts, deadline = [], args.launch_timeout
    mode = "torch.distributed.run started supervisors; each runs its rank in a child process per attempt"
    for extra in reduce_ladder(args):
        port = [free_port() if rank == 0 else None]
        dist.broadcast_object_list(port, src=0)
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port[0]), BENCH_SELF_LAUNCHED="1", BENCH_WORKER="1")
        for k in [k for k in env if k.startswith("TORCHELASTIC_")]:
            env.pop(k)  # the children rendezvous among themselves on the attempt's port: rank 0's child hosts the store, not the agent
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out_file = tempfile.TemporaryFile()
        t0 = time.time()
        child = subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv, *extra, "--launch-timeout", str(deadline)], env=env,
                                 start_new_session=True, stdout=out_file if rank == 0 else subprocess.DEVNULL)
        why = None
        while child.poll() is None:
            if time.time() - t0 > deadline:
                why = f"deadline of {deadline:.0f} s passed"
                kill_ranks([child])
                break
            time.sleep(0.05)
        rc = child.returncode
        out_file.seek(0)
        line = None
        for ln in out_file.read().decode(errors="replace").splitlines():
            if ln.startswith("{"):
                line = ln
        out_file.close()
        ok_here = why is None and rc in (0, 3) and (rank != 0 or line is not None)
        flags = torch.tensor([1 if ok_here else 0, 1 if rc == 3 else 0], dtype=torch.int32)
        dist.all_reduce(flags, op=dist.ReduceOp.MIN)  # [every rank fine, every rank reports a parity mismatch]
        codes = [None] * world
        dist.all_gather_object(codes, rc)
        attempts.append({"extra_args": extra, "exit_codes": codes, "seconds": round(time.time() - t0, 1),
                         "failure": None if int(flags[0]) else (why or f"exit codes {codes}"), "rendezvous": f"127.0.0.1:{port[0]}"})
        if int(flags[0]):
            if rank == 0:
                res = json.loads(line)
                res["launcher"] = {"mode": mode, "attempts": attempts, "deadline_s_per_attempt": args.launch_timeout}
                os.write(real_stdout, (json.dumps(res) + "\n").encode())
            dist.barrier()
            dist.destroy_process_group()
            return 3 if 3 in codes else 0
        log(f"bench.py supervisor (rank {rank}): attempt {extra or ['as asked']} failed ({attempts[-1]['failure']}); next rung")
        deadline = max(120.0, args.launch_timeout / 2)
    if rank == 0:
        os.write(real_stdout, (failure_line(args, attempts, mode) + "\n").encode())
    dist.barrier()
    dist.destroy_process_group()
    return 5


def launcher(args, argv):
    """Attempts in order: what was asked for; the torch.distributed table sum; replicas with no table sum at all.  The
    first attempt whose ranks all exit 0 and whose rank 0 printed a line wins; its line goes out with `launcher` added.
    If nothing works a line still goes out (value 0, the attempts' failures inside) and the exit code is 5."""
    ladder = reduce_ladder(args)
    attempts, deadline = [], args.launch_timeout
    for extra in ladder:
        line, outcome = launch_once(argv + extra, args.gpus, deadline)
        outcome["extra_args"] = extra
        attempts.append(outcome)
        ok = line is not None and not outcome["failure"]
        parity_exit = ok and 3 in outcome["exit_codes"]
        if ok:
            try:
                res = json.loads(line)
            except ValueError:
                outcome["failure"] = "rank 0's line is not JSON"
                continue
            res["launcher"] = {"mode": "bench.py started its own ranks (subprocess per GPU, rendezvous on 127.0.0.1)",
                               "attempts": attempts, "deadline_s_per_attempt": args.launch_timeout}
            print(json.dumps(res), flush=True)
            return 3 if parity_exit else 0
        deadline = max(120.0, args.launch_timeout / 2)  # the later rungs get less: the whole ladder stays bounded
    print(failure_line(args, attempts, "bench.py started its own ranks"), flush=True)
    return 5


# =====================================================================================================================
#  device plumbing: the real thing is torch.cuda + libscope_hip.so; `--backend emul` is the CPU emulation build of the
#  same sources (tests/emul) so that the N-rank logic of this file can run without a GPU (tests/test_bench_launcher.py)
# =====================================================================================================================
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
    flat["all_parity"] = bool(ok)
    return flat


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
    ap.add_argument("--lib", default=None, help="another build of libscope_hip.so (A/B of kernel variants; the line names it)")
    return ap.parse_args(argv)


def main():
    args = parse_args()
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
        os.write(real_stdout, (json.dumps(result) + "\n").encode())
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
        p_el, p_sus, p_ln, p_kms = timed_run(dev, ctl, ctx, stream, pw.step, args.steps, args.warmup, 0.2, 0.5)
        p_ok = pw.check(o, placed if src_name == "rgb" else rgb, placed if src_name == "yuv" else yuv, np, frames=spread(n, 3))
        p_ms = p_el / args.steps * 1e3
        p_rf = roofline(wl, pw.algo_bytes, p_ln, p_kms, p_ms, None)
        placement = {"candidates": args.placement_candidates, "pool": src_name,
                     "by": "scope_frame_pool_alloc (C-ABI): candidates held at once, each read by the vectorscope accumulation kernel "
                           "timed with its own begin/end stamps, the fastest kept, the rest freed; untimed set-up",
                     "pool_read_gbps": rates, "value": round(n * w * h * args.steps / p_el / 1e9, 3), "ms_per_step": round(p_ms, 4),
                     "frac": p_rf["frac"], "frac_kernel": p_rf["frac_kernel"], "avg_launch_ms": p_rf["avg_launch_ms"], "parity": bool(p_ok)}
        rf["frac_placed"] = p_rf["frac"]
        rf["frac_kernel_placed"] = p_rf["frac_kernel"]
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

    # ---- the legs' headline numbers once more as FLAT scalars, at the top level and inside `roofline`: a reader (the driver's
    # record keeps scalar fields of the objects it knows and only the NAMES of the others) then has every README row in one place
    flat = flat_summary(result)
    result.update(flat)
    result["roofline"].update(flat)

    faulthandler.cancel_dump_traceback_later()
    if rank == 0:
        os.write(real_stdout, (json.dumps(result) + "\n").encode())
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


class Legs:
    """The short legs of the default one-GPU run that share the headline's frame geometry."""

    def __init__(self, dev, ctl, ctx, stream, o, np, torch, w, h, n):
        self.dev, self.ctl, self.ctx, self.stream, self.o, self.np, self.torch = dev, ctl, ctx, stream, o, np, torch
        self.w, self.h, self.n = w, h, n

    def variant(self, wl, rgb, yuv, comp, steps=20, seconds=0.5, check=True):
        torch, np = self.torch, self.np
        vw = Workload(torch, self.ctx, wl, rgb, yuv, self.w, self.h, self.n, comp)
        el, sus, ln, kms = timed_run(self.dev, self.ctl, self.ctx, self.stream, vw.step, steps, 3, 0.2, seconds)
        ok = vw.check(self.o, rgb, yuv, np, frames=spread(self.n, 3)) if check else None
        tr, _ = pmc_traffic(wl, self.n, self.w, self.h, comp == 0x07 and vw.linesize == 4 * self.w)
        vr = roofline(wl, vw.algo_bytes, ln, kms, sus["ms_per_step"], tr)
        if wl == "three_scope":
            # SURVEY 8(d)'s config-5 bytes (ONE read of the frames + the three tables).  The label "hbm" would flatter this
            # workload: its ceiling is VALU issue (profiles/round4/pmc_sq_three_scope_x64_*.txt).
            vr["bound"] = "valu issue (K0 + three waveform adds + one vectorscope add per texel); hbm is the yardstick, not the ceiling"
            vr["ceiling_estimate"] = {"gpix_per_s": 1200, "frac_of_hbm_peak": 0.67,
                                      "from": "the per-texel instruction floor of this structure (K0 11.5 + waveform addresses 6 + alpha 1 + "
                                              "vectorscope 3.5 + load 1 = 23 VALU instructions) at the 73 % VALU occupancy four waves per SIMD "
                                              "reach (measured: 29.2 instructions per texel, 947 Gpix/s); DESIGN.md 3.5"}
            vr["note"] = "algorithmic bytes = SURVEY 8(d)'s config-5 figure: one read of the frames + the three tables"
        out = {"value": round(self.n * self.w * self.h / (sus["ms_per_step"] * 1e-3) / 1e9, 3), "unit": "Gpix/s", "components": f"{comp:#04x}",
               "ms_per_step": sus["ms_per_step"], "steps": sus["steps"], "parity": ok, "roofline": vr}
        del vw
        return out

    def distributions(self, make_frames, pad_pitch, punch_alpha_holes, rgb_uniform, yuv_uniform):
        """SURVEY.md 8(d) D2-D6 (+ natural / screen stand-ins) under this clock: kernel frac of the headline kernel and of
        the fused-K0 vectorscope on each; alpha holes only change what histogram and waveform do, so those run there."""
        torch = self.torch
        out = {}

        def both(rgb, yuv, trio=True):
            a = self.variant("vectorscope", rgb, yuv, 0x07, steps=10, seconds=0.5)
            b = self.variant("vectorscope_rgb", rgb, yuv, 0x07, steps=10, seconds=0.5)
            res = {"vectorscope": {"frac": a["roofline"]["frac"], "frac_kernel": a["roofline"]["frac_kernel"], "gpix_per_s": a["value"], "parity": a["parity"]},
                   "vectorscope_rgb": {"frac": b["roofline"]["frac"], "frac_kernel": b["roofline"]["frac_kernel"], "gpix_per_s": b["value"], "parity": b["parity"]}}
            if trio:  # the one-read three-scope kernel is LDS / VALU bound: same-address LDS atomics make it content dependent
                c = self.variant("three_scope", rgb, yuv, 0x07, steps=6, seconds=0.4)
                res["three_scope"] = {"gpix_per_s": c["value"], "frac": c["roofline"]["frac"], "parity": c["parity"]}
            return res

        for kind in ("natural", "concentrated", "flat", "screen", "ramp"):
            rgb, yuv = make_frames(torch, self.ctx, self.n, self.w, self.h, kind, 0xD157 + len(out), device=self.dev.device)
            out[kind] = both(rgb, yuv)
            del rgb, yuv
            self.dev.empty_cache()
        prgb, pyuv = pad_pitch(torch, rgb_uniform[:self.n], self.w), pad_pitch(torch, yuv_uniform[:self.n], self.w)
        out["padded_pitch"] = dict(both(prgb, pyuv), linesize=int(prgb.shape[2]))
        del prgb, pyuv
        holes = punch_alpha_holes(torch, rgb_uniform[:self.n].clone(), self.w)
        hy = yuv_uniform[:self.n]
        a = self.variant("histogram", holes, hy, 0x07, steps=10, seconds=0.5)
        b = self.variant("waveform", holes, hy, 0x07, steps=10, seconds=0.5)
        out["alpha_holes"] = {"histogram": {"frac": a["roofline"]["frac"], "frac_kernel": a["roofline"]["frac_kernel"], "gpix_per_s": a["value"], "parity": a["parity"]},
                              "waveform": {"frac": b["roofline"]["frac"], "frac_kernel": b["roofline"]["frac_kernel"], "gpix_per_s": b["value"], "parity": b["parity"]},
                              "note": "the vectorscope ignores alpha (src/vectorscope.c:217-238): histogram and waveform are what D5 exercises"}
        del holes
        self.dev.empty_cache()
        out["what"] = (f"{self.n} frames per launch; frac = algorithmic bytes / the whole step / 8 TB/s, frac_kernel = / the accumulation "
                       "kernel's own begin/end stamps; pools taken as allocated; uniform noise is the headline")
        return out


def config1_leg(np, ctx, o, reps=25):
    """BASELINE configs[0]: 256-bin RGB histogram over one 1280x720 frame on the CPU path (the oracle = this build's port of
    src/histogram.c:330-418, one thread, counts + level pass), and the GPU's result for the same frame."""
    w, h = 1280, 720
    rgb = o.fill_frame(w, h, 1, 0x720)
    times = []
    for _ in range(5 + reps):
        t0 = time.perf_counter()
        want, want_hi = o.histogram(rgb, None, w, 0x07)
        times.append(time.perf_counter() - t0)
    times = sorted(times[5:])
    ms = times[len(times) // 2] * 1e3
    got, hi, st = ctx.histogram(rgb=rgb, yuv=None, width=w, components=0x07)
    same = st == 0 and bool(np.array_equal(got.view(np.uint32), want.view(np.uint32))) and bool(np.array_equal(hi, want_hi))
    return {"workload": "histogram RGB (0x07), one 1280x720 BGRA8 frame, CPU path only (oracle port, one thread, -O2)",
            "ms_per_frame": round(ms, 4), "gpix_per_s": round(w * h / (ms * 1e-3) / 1e9, 3), "median_of": reps,
            "gpu_tables_bit_exact": same, "kind": "port (parity unpinned)",
            "note": "excludes the reference's GPU->CPU readback; the GPU call here is a host-buffer call (PCIe-inclusive), not timed"}


def config4_leg(dev, ctl, torch, np, ctx, o, comm, rank, world, w, h, frames=16, seconds=0.6, overlap=None):
    """ONE batch of w x h (7680x4320) frames, the same frames whatever N is (shared seed); rank r owns rows
    [h r/N, h (r+1)/N) of every frame; per step every rank reduces its tiles, the packed tables are summed over RCCL
    through the C-ABI and clamped.  Figures: throughput with the steps enqueued back to back (step i's sum under step
    i+1's accumulation when `overlap`), the same with the sum on the accumulation's stream, and the latency of one step
    with its reduce waited for (nothing bucketed or deferred)."""
    from obs_color_monitor_amd.row_tiles import tile_rows
    from obs_color_monitor_amd.synth import make_frame_tiles

    r0, r1 = tile_rows(h, rank, world)
    th = r1 - r0
    rgb, yuv = make_frame_tiles(torch, ctx, frames, w, h, r0, r1, 0xC0FFEE, device=dev.device)
    del rgb
    batch = ctx.make_device_batch(yuv, 4 * w, w, th, planes_are_rgb=False)
    wires = [torch.empty((frames, 32768), dtype=torch.int32, device=dev.device) for _ in range(2)]
    outs = [torch.empty((frames, 65536), dtype=torch.uint8, device=dev.device) for _ in range(2)]
    count = [0]

    def step(ov):
        buf = count[0] & 1
        count[0] += 1
        rctx = ctx
        if ov:
            ov.before_compute(buf)
            rctx = ov.ctx
        ctx.vectorscope_batch_dev(batch, wires[buf], sync=False, packed16=True)
        if ov:
            ov.after_compute(buf)
        rctx.table_allreduce(comm, wires[buf], frames * 32768)
        rctx.unpack16_clamp_dev(wires[buf], outs[buf], frames * 32768)
        if ov:
            ov.after_reduce(buf)
        return buf

    def throughput(ov):
        """every rank runs the SAME number of steps (a step holds a collective): fixed chunks, the ranks agree after each
        whether to go on"""
        for _ in range(6):
            step(ov)
        dev.sync()
        ctl.barrier()
        t0 = time.perf_counter()
        steps, chunk = 0, 40 if not dev.emul else 2
        while True:
            for _ in range(chunk):
                step(ov)
            steps += chunk
            dev.sync()
            if not ctl.any_true(time.perf_counter() - t0 < seconds):
                break
        dev.sync()
        ctl.barrier()
        return ctl.max(time.perf_counter() - t0) / steps * 1e3, steps

    ms_serial, steps_serial = throughput(None)
    ms_overlap, steps_overlap = throughput(overlap) if overlap else (None, 0)
    dev.sync()
    lat = []
    for _ in range(30 if not dev.emul else 2):
        ctl.barrier()
        dev.sync()
        t1 = time.perf_counter()
        last = step(None)
        dev.sync()
        lat.append(time.perf_counter() - t1)
    lat_ms = ctl.max(sorted(lat)[len(lat) // 2]) * 1e3
    # parity of frames 0 and last of the batch: the sum of every rank's oracle table of its tile, clamped once; checked on
    # the table the serial path left AND on one the overlapped path left (the two must be the same table)
    ok = True
    for f in sorted({0, frames - 1}):
        want = np.minimum(ctl.sum_array(o.vectorscope(yuv[f].cpu().numpy(), w).astype(np.int64).reshape(-1)), 255)
        ok &= bool(np.array_equal(outs[last][f].cpu().numpy().astype(np.int64), want))
        if overlap:
            b2 = step(overlap)
            dev.sync()
            ok &= bool(np.array_equal(outs[b2][f].cpu().numpy().astype(np.int64), want))
    ok = ctl.all_true(ok)
    best = min(ms_serial, ms_overlap) if ms_overlap else ms_serial
    return {
        "value": round(frames * w * h / (best * 1e-3) / 1e9, 3), "unit": "Gpix/s", "scaling": "strong", "ms_per_step": round(best, 4),
        "ms_per_step_reduce_on_the_accumulation_stream": round(ms_serial, 4),
        "ms_per_step_reduce_overlapped_second_stream": round(ms_overlap, 4) if ms_overlap else None,
        "step_latency_ms_reduce_waited": round(lat_ms, 4),
        "steps_timed": [steps_serial, steps_overlap],
        "partition": f"{frames} frames of {w}x{h} (shared seed: the same frames at every N), {world} row tile(s) of {w}x{th} per frame "
                     f"+ RCCL sum of the 256x256 tables through the C-ABI, every step",
        "frames_per_step": frames, "rows_of_this_rank": [r0, r1], "parity": bool(ok),
    }


def cpu_baseline(np, o, wl, plane_rgb, plane_yuv, w, h, components, seconds):
    def one(kind=wl, comp=components):
        if kind in ("vectorscope", "vectorscope_rgb"):
            return o.vectorscope(plane_yuv, w)
        if kind == "waveform":
            return o.waveform(plane_rgb, plane_yuv, w, comp)
        return o.histogram_counts(plane_rgb, plane_yuv, w, comp)

    def rate(kind, comp, secs):
        t_end = time.perf_counter() + secs
        reps, t_start = 0, time.perf_counter()
        while time.perf_counter() < t_end or reps < 3:
            one(kind, comp)
            reps += 1
        dt = time.perf_counter() - t_start
        return reps, dt, reps * w * h / dt / 1e9

    reps, cpu_t, gpix = rate(wl, components, seconds)
    res = {
        "value": round(gpix, 4),
        "unit": "Gpix/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{reps} passes of oracle {wl} (scope_oracle.c, gcc -O2, one thread as the reference's "
                  f"single color-monitor thread) over frame 0 of the same batch, {cpu_t:.1f} s; "
                  f"host has {os.cpu_count()} cores; excludes the reference's GPU->CPU readback",
    }
    # the other two loops of the path on the same frame, one thread, short samples
    others = {}
    for kind, comp in (("vectorscope", 0x07), ("waveform", 0x07), ("waveform", 0x20), ("histogram", 0x07)):
        if kind == wl and comp == components:
            continue
        r, dt, g = rate(kind, comp, min(2.0, seconds / 4))
        others[f"{kind}_{comp:#04x}"] = {"value": round(g, 4), "unit": "Gpix/s", "cores": 1, "passes": r, "seconds": round(dt, 2)}
    res["other_loops_one_thread"] = others
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
        r = calls * mt_reps * w * h / (time.perf_counter() - t_start) / 1e9
        if best is None or r > best[0]:
            best = (r, nthreads, calls)
    res["all_cores"] = {
        "value": round(best[0], 3), "unit": "Gpix/s", "cores": best[1], "host_cores": threads, "cpu": cpu_model,
        "same_table": mt_same,
        "sample": f"best of 16..{threads} POSIX threads: {best[2]} calls x {mt_reps} passes over frame 0 (row tiles with "
                  "private tables; column tiles for the waveform; thread start-up once per call)",
    }
    return res


if __name__ == "__main__":
    main()
