"""bench_launch.py -- `python bench.py --gpus N` with no WORLD_SIZE around starts N ranks itself (launcher); under
`python -m torch.distributed.run ... bench.py --gpus N` every process torchrun starts supervises its rank (supervisor).  Every
attempt has a deadline; a hung or failed collective ends the attempt and the next rung of the ladder runs (the C-ABI's RCCL sum
-> torch.distributed -> replicas without a table sum, which says so in every field that matters); exactly ONE JSON line reaches
stdout in every case.  tests/test_bench_launcher.py runs all of it on the emulation build at N = 2, 4 and 8."""
import json
import os
import signal
import socket
import subprocess
import sys
import time

from bench_core import LINE_LIMIT, METRIC, ROOT, compact_launcher, log

BENCH = os.path.join(ROOT, "bench.py")


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
        procs.append(subprocess.Popen([sys.executable, BENCH, *argv], env=env, start_new_session=True,
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
        "launcher": compact_launcher({"mode": mode, "attempts": attempts, "deadline_s_per_attempt": args.launch_timeout}),
    })


def with_launcher(res, mode, attempts, args):
    """Rank 0's (already compact) line with the launcher's object added, still under LINE_LIMIT.  A line the replicas rung
    printed (no table sum took place) can never call itself strong scaling or carry a config-4 value."""
    res["launcher"] = compact_launcher({"mode": mode, "attempts": attempts, "deadline_s_per_attempt": args.launch_timeout})
    if "--no-collective" in (attempts[-1].get("extra_args") or []) or args.no_collective:
        res["scaling"] = "weak"
        res.pop("config4", None)
        res.pop("config4_gpix", None)
    line = json.dumps(res)
    if len(line.encode()) >= LINE_LIMIT:  # eight ranks, three attempts, long failure texts: the attempts keep their outcome only
        for a in res["launcher"]["attempts"]:
            a["failure"] = a["failure"] and a["failure"][:40]
        line = json.dumps(res)
    return line


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
        child = subprocess.Popen([sys.executable, BENCH, *argv, *extra, "--launch-timeout", str(deadline)], env=env,
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
                os.write(real_stdout, (with_launcher(json.loads(line), mode, attempts, args) + "\n").encode())
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
        # the ranks' own watchdog follows the deadline of THEIR attempt (bench.py: rank_timeout = launch_timeout - 15 s)
        line, outcome = launch_once(argv + extra + ["--launch-timeout", str(deadline)], args.gpus, deadline)
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
            print(with_launcher(res, "bench.py started its own ranks (subprocess per GPU, rendezvous on 127.0.0.1)", attempts, args), flush=True)
            return 3 if parity_exit else 0
        deadline = max(120.0, args.launch_timeout / 2)  # the later rungs get less: the whole ladder stays bounded
    print(failure_line(args, attempts, "bench.py started its own ranks"), flush=True)
    return 5


def dry_run_plan(args, num_cu=256):
    """`bench.py --gpus N --dry-run`: what an N-rank run WOULD do, rank by rank -- rows, launches, wire bytes, reduce path, streams,
    the launcher's ladder -- computed from the same rules the run uses (row_tiles.tile_rows, the C-ABI's slices-per-frame rule,
    the in-launch fold's bound) without touching a device, torch or the library.  The first real 8-GPU run can be held against
    it field by field (`config.partition`, `config4.rows_of_this_rank`, `config.reduce`, `rccl.ranks_in_communicator`)."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("row_tiles", os.path.join(os.path.dirname(os.path.abspath(__file__)), "obs-color-monitor_amd", "row_tiles.py"))
    rt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rt)
    n = args.gpus
    if args.geometry:
        w, h = (int(x) for x in args.geometry.lower().split("x"))
    else:
        w, h = (3840, 2160) if args.size == "4k" else (7680, 4320)
    frames = args.frames or (64 if w * h <= 3840 * 2160 else 16)
    table_bytes = {"vectorscope": 65536, "vectorscope_rgb": 65536, "histogram": 4096, "waveform": 256 * w * 4, "three_scope": 65536 + 4096 + 256 * w * 4}

    def launch_shape(nf, units=1):
        slices = max(1, min(256, num_cu // max(1, nf * units)))  # scope_api.hip: pick_slices
        fold = 2 <= slices <= 4  # scope_device.h: VSS_FOLD_MAX
        return {"workgroups": nf * slices, "slices_per_frame": slices, "threads_per_workgroup": 1024,
                "partial_tables": "none (one slice per frame)" if slices == 1 else
                                  ("sparse, added up INSIDE the launch by the frame's last-arriving workgroup" if fold else
                                   "sparse + scope_sat_reduce_sparse_kernel")}

    c4w, c4h = (int(x) for x in args.config4_geometry.lower().split("x"))
    wl = args.workload
    sums = wl in ("vectorscope", "vectorscope_rgb", "histogram") and n > 1 and not args.no_collective
    wire_words = frames * (32768 if wl.startswith("vectorscope") else 1024)
    ranks = []
    for r in range(n):
        r0, r1 = rt.tile_rows(c4h, r, n)
        ranks.append({
            "rank": r, "device": f"cuda:{r}" if not args.oversubscribe else "cuda:(rank mod GPUs)",
            "headline": {"frames": frames, "tile": f"{w}x{h} (rows [{h * r}, {h * (r + 1)}) of a frame {n} times as tall: weak)",
                         "bytes_read_per_step": frames * w * h * 4, **launch_shape(frames)},
            "config4": {"rows": [r0, r1], "tile": f"{c4w}x{r1 - r0}", "frames": args.config4_frames,
                        "bytes_read_per_step": args.config4_frames * c4w * (r1 - r0) * 4, **launch_shape(args.config4_frames)},
        })
    plan = {
        "dry_run": True, "metric": METRIC, "n_gpus": n, "workload": wl, "scaling": "strong" if args.strong else "weak",
        "value_is": "pixels all ranks reduced in exactly K steps / the slowest rank's time between barrier + synchronize (MAX over ranks)",
        "steps": args.steps, "warmup": args.warmup,
        "launch": {"how": "bench.py starts one subprocess per GPU itself (or supervises the ranks torch.distributed.run started)",
                   "rendezvous": "127.0.0.1, gloo control plane on CPU tensors", "env": {"HSA_ENABLE_IPC_MODE_LEGACY": "0"},
                   "deadline_s_per_attempt": args.launch_timeout,
                   "ladder": ["C-ABI: scope_table_allreduce (RCCL, dlopen)"] + [" ".join(x) for x in reduce_ladder(args)[1:]]},
        "table_sum": ({"what": f"in-place ncclAllReduce(sum, uint32) of {wire_words} words = {wire_words * 4} bytes per step per rank"
                               + (" (two clamped bins per word, SCOPE_OUT_PACKED16 straight out of the accumulation), then unpack + clamp" if wl.startswith("vectorscope") else ""),
                       "ranks_in_communicator_expected": n,
                       "stream": "the accumulation's own stream" if args.serial_reduce else
                                 "a second hipStream + scope context: the sum of step i runs under the accumulation of step i + 1 (two wire buffers, events both ways)",
                       "config4_wire_bytes_per_step": args.config4_frames * 32768 * 4}
                      if sums else {"what": "none" + (" (the waveform shards by column strip: no collective)" if wl == "waveform" else " (replicas)" if n > 1 else " (one rank)")}),
        "table_bytes_out_per_frame": table_bytes.get(wl),
        "ranks": ranks,
    }
    return plan
