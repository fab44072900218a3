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
