"""bench.py's N-rank path on CPU: `--backend emul` runs the same file against tests/emul/libscope_emul.so (the unmodified
kernel sources compiled for the CPU), gloo as the control plane and tests/emul/libshm_rccl_emul.so as the collective
library the C-ABI binds (SCOPE_RCCL_LIBRARY) -- so `python bench.py --gpus 2` with no WORLD_SIZE around starts its own two
ranks here exactly as it would on a multi-GPU node, sums the tables through scope_table_allreduce, runs BASELINE config 4
in row tiles, and a hung or missing collective still ends in ONE JSON line."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--backend", "emul", "--geometry", "128x24", "--frames", "2", "--pool", "4", "--steps", "2", "--warmup", "1",
         "--settle-seconds", "0", "--min-seconds", "0", "--no-cpu"]
C4 = ["--config4-geometry", "256x40", "--config4-frames", "2", "--strong"]


@pytest.fixture(scope="module")
def shm_rccl(emul_lib):
    from conftest import locked_make

    d = os.path.join(ROOT, "tests", "emul")
    locked_make(d, "libshm_rccl_emul.so")
    return os.path.join(d, "libshm_rccl_emul.so")


DETAIL_DIR = tempfile.mkdtemp(prefix="bench_detail_")  # never the repository's bench_detail.json: that file belongs to real runs


def clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BENCH_SELF_LAUNCHED", "SCOPE_RCCL_LIBRARY")}
    env.update(OMP_NUM_THREADS="1", BENCH_DETAIL=os.path.join(DETAIL_DIR, f"detail_{os.getpid()}_{len(os.listdir(DETAIL_DIR))}.json"), **extra)
    return env


def run_bench(args, env, prefix=(), timeout=600):
    out = subprocess.run([sys.executable, *prefix, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True,
                         timeout=timeout, cwd=ROOT)
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), \
        f"exactly one line on stdout, the JSON line, expected; rc={out.returncode}\n{out.stdout[-2000:]}\n{out.stderr[-4000:]}"
    # the driver's record lost round 4's 21.7 KB line: the line is SHORT, whatever N is and however the run ended
    assert len(lines[0].encode()) < 8000, f"the stdout line is {len(lines[0].encode())} bytes"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config"):
        assert k in d, k
    assert list(d)[:4] == ["metric", "value", "unit", "n_gpus"], "the contract's fields lead the line"
    assert "workload" in d["config"]
    if d["value"]:
        assert d["roofline"]["frac"] >= 0 and d["roofline"]["bound"] == "hbm" and d["roofline"]["algorithmic_bytes_per_launch"] > 0
        assert all(not isinstance(v, (dict, list)) or k in ("config", "roofline", "cpu_baseline", "config4", "launcher")
                   for k, v in d.items()), "no nested legs on the line"
    return out.returncode, d, out.stderr


def test_gpus2_without_world_size_starts_its_own_ranks(shm_rccl):
    rc, d, err = run_bench(["--gpus", "2", *SMALL, *C4], clean_env())
    assert rc == 0, err[-3000:]
    assert d["n_gpus"] == 2 and d["parity"].startswith("bit-exact") and d["unit"] == "Gpix/s"
    la = d["launcher"]
    assert len(la["attempts"]) == 1 and la["attempts"][0]["exit_codes"] == [0, 0] and la["attempts"][0]["failure"] is None
    cfg = d["config"]
    assert cfg["launch"] == "self-launched rank"
    assert cfg["rccl"]["ranks_in_communicator"] == 2, "the line must say how many ranks the collective library saw"
    assert cfg["reduce"].startswith("C-ABI: scope_table_allreduce")
    c4 = d["config4"]
    assert c4["parity"] is True and c4["scaling"] == "strong" and c4["rccl_ranks_in_communicator"] == 2
    assert c4["rows_of_this_rank"] == [0, 20]  # rank 0's half of the 40 rows
    assert c4["ms_per_step_reduce_on_the_accumulation_stream"] > 0 and c4["ms_per_step_reduce_overlapped_second_stream"] > 0
    assert c4["step_latency_ms_reduce_waited"] > 0
    assert d["scaling"] == "strong"  # --strong made config 4 the headline


def test_gpus1_line_carries_config4_with_a_one_rank_communicator(shm_rccl):
    rc, d, err = run_bench(["--gpus", "1", *SMALL, *C4], clean_env(SCOPE_RCCL_LIBRARY=shm_rccl))
    assert rc == 0, err[-3000:]
    assert d["n_gpus"] == 1 and "launcher" not in d and d["config"]["launch"] == "single process"
    c4 = d["config4"]
    assert c4["parity"] is True and c4["rccl_ranks_in_communicator"] == 1 and c4["rows_of_this_rank"] == [0, 40]


def test_under_torch_distributed_run(shm_rccl):
    env = clean_env(MASTER_ADDR="127.0.0.1")
    prefix = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29519"]
    rc, d, err = run_bench(["--gpus", "2", *SMALL, "--no-extras"], env, prefix=prefix)
    assert rc == 0, err[-3000:]
    assert d["n_gpus"] == 2 and d["config"]["launch"] == "supervised rank under torch.distributed.run"
    assert len(d["launcher"]["attempts"]) == 1 and d["launcher"]["attempts"][0]["exit_codes"] == [0, 0]
    assert d["config"]["rccl"]["ranks_in_communicator"] == 2 and d["parity"].startswith("bit-exact")


def test_hung_collective_under_torch_distributed_run(shm_rccl):
    # the driver's N > 1 command line: the processes torchrun starts supervise their ranks, agree that the first attempt
    # failed (rank 1 never entered its collective) and run the next rung together
    env = clean_env(MASTER_ADDR="127.0.0.1", SHM_RCCL_HANG_RANK="1", SHM_RCCL_TIMEOUT_S="1000")
    prefix = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29521"]
    rc, d, err = run_bench(["--gpus", "2", *SMALL, "--no-extras", "--launch-timeout", "20"], env, prefix=prefix)
    assert rc == 0, err[-3000:]
    attempts = d["launcher"]["attempts"]
    assert len(attempts) == 2 and attempts[0]["failure"] and attempts[1]["failure"] is None and attempts[1]["extra_args"] == ["--torch-reduce"]
    assert d["config"]["reduce"].startswith("torch.distributed") and d["parity"].startswith("bit-exact")


def test_hung_collective_still_prints_a_line(shm_rccl):
    # rank 1 never enters its first ncclAllReduce: rank 0 waits, the ranks' own watchdog (deadline - 15 s) dumps stacks and
    # leaves, the launcher ends the attempt and the next rung (torch.distributed) produces the line
    env = clean_env(SHM_RCCL_HANG_RANK="1", SHM_RCCL_TIMEOUT_S="1000")
    rc, d, err = run_bench(["--gpus", "2", *SMALL, "--no-extras", "--launch-timeout", "20"], env)
    assert rc == 0, err[-3000:]
    attempts = d["launcher"]["attempts"]
    assert len(attempts) == 2 and attempts[0]["failure"] and attempts[1]["failure"] is None
    assert attempts[1]["extra_args"] == ["--torch-reduce"]
    assert d["config"]["reduce"].startswith("torch.distributed") and d["parity"].startswith("bit-exact")
    assert "Timeout" in err  # faulthandler said where the stuck rank was


def test_missing_collective_library_falls_back_together(shm_rccl):
    rc, d, err = run_bench(["--gpus", "2", *SMALL, "--no-extras"], clean_env(SCOPE_RCCL_LIBRARY="/nonexistent/librccl.so"))
    assert rc == 0, err[-3000:]
    assert len(d["launcher"]["attempts"]) == 1, "no rung is burnt: the ranks agree on the fall-back inside the first attempt"
    assert d["config"]["reduce"].startswith("torch.distributed") and d["config"]["rccl"] is None


def test_replicas_rung_says_what_it_is(shm_rccl):
    rc, d, err = run_bench(["--gpus", "2", *SMALL, "--no-extras", "--no-collective"], clean_env())
    assert rc == 0, err[-3000:]
    assert d["config"]["reduce"].startswith("NONE") and "NO table sum" in d["config"]["partition"]
    assert "config4" not in d


def test_replicas_are_never_strong(shm_rccl):
    # the last rung of the ladder has no table sum: whatever was asked for, its line is weak scaling and carries no config-4 value
    rc, d, err = run_bench(["--gpus", "2", *SMALL, *C4, "--no-collective"], clean_env())
    assert rc == 0, err[-3000:]
    assert d["scaling"] == "weak" and "config4" not in d and "config4_gpix" not in d
    assert d["config"]["reduce"].startswith("NONE") and d["config"]["rccl"] is None


def test_line_of_a_full_run_is_short_and_complete(tmp_path):
    """compact_line on a REAL full result (round 4's 21.7 KB line, the one the driver's record lost): under 6 KB, headline
    first, roofline and cpu_baseline with the keys a reader recomputes from, at most 40 flat scalars, no nested legs."""
    sys.path.insert(0, ROOT)
    import bench_core

    full = json.load(open(os.path.join(ROOT, "profiles", "round4", "bench_x4.json")))
    for k in list(full):  # round 4 repeated every flat scalar at the top level: a result as bench.py builds it today has none
        if not isinstance(full[k], (dict, list)) and k.endswith(("_gpix", "_frac", "_frac_kernel", "_us", "_fps", "_ms", "_per_frame", "_communicator")):
            full.pop(k)
    full["launcher"] = {"mode": "m" * 300, "deadline_s_per_attempt": 480.0,
                        "attempts": [{"extra_args": ["--torch-reduce"], "exit_codes": [0] * 8, "seconds": 99.9, "failure": "f" * 500,
                                      "rendezvous": "127.0.0.1:1"}] * 3}
    full["n_gpus"] = 8
    line = bench_core.compact_line(full, "bench_detail.json")
    assert len(line.encode()) < bench_core.LINE_LIMIT < 8000
    d = json.loads(line)
    assert list(d)[:7] == ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step"]
    assert d["roofline"]["frac"] == 0.8039 and d["roofline"]["frac_kernel"] == 0.8194 and d["roofline"]["avg_launch_ms"] == 0.32458
    assert d["roofline"]["algorithmic_bytes_per_launch"] == 2127560704 and d["roofline"]["peak"] == 8000.0
    assert d["cpu_baseline"]["value"] == 1.5679 and d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["host_cores"] == 256
    assert d["config"]["workload"].startswith("vectorscope 256x256") and d["config"]["kernel_sources_sha256_16"] == "57263b4608419f98"
    assert d["config4"]["value"] == 1557.44 and d["config4"]["rccl_ranks_in_communicator"] == 1  # N > 1: the strong leg is on the line
    flat = [k for k, v in d.items() if not isinstance(v, (dict, list)) and k in bench_core.LINE_SCALARS]
    assert 30 <= len(flat) <= 40 and d["three_scope_frac_kernel"] == 0.594 and d["ramp_three_scope_frac"] > 0 and d["all_parity"] is True
    assert "variants" not in d and "latency" not in d and "distributions" not in d and "stream" not in d
    path = bench_core.write_detail(full, str(tmp_path / "bench_detail.json"))
    assert json.load(open(path))["variants"]["waveform"]["value"] == 1436.104


def test_nothing_works_still_one_line(shm_rccl, tmp_path):
    # every rung dies at once (an unknown option value): the launcher's own line, exit code 5
    rc, d, err = run_bench(["--gpus", "2", *SMALL, "--no-extras", "--option", "no_such_option=1", "--launch-timeout", "60"], clean_env())
    assert rc == 5 and d["value"] == 0.0 and "error" in d and len(d["launcher"]["attempts"]) == 3


# ---- N = 4 and N = 8: the driver's 8-GPU run must not be the launcher's first 8-process run ----------------------


def test_gpus8_config4_real_tile_self_launched(shm_rccl):
    """BASELINE config 4's real shape: 7680x4320 frames, rank r owns rows [540 r, 540 (r + 1)), eight ranks, the tables summed
    through scope_table_allreduce and compared with the oracle's table of every rank's tile (summed over gloo, clamped once)."""
    rc, d, err = run_bench(["--gpus", "8", *SMALL, "--config4-geometry", "7680x4320", "--config4-frames", "2", "--strong"], clean_env(),
                           timeout=900)
    assert rc == 0, err[-3000:]
    assert d["n_gpus"] == 8 and d["parity"].startswith("bit-exact") and d["scaling"] == "strong"
    la = d["launcher"]
    assert len(la["attempts"]) == 1 and la["attempts"][0]["exit_codes"] == [0] * 8 and la["attempts"][0]["failure"] is None
    assert d["config"]["rccl"]["ranks_in_communicator"] == 8
    c4 = d["config4"]
    assert c4["parity"] is True and c4["rccl_ranks_in_communicator"] == 8 and c4["rows_of_this_rank"] == [0, 540]
    assert "8 row tile(s) of 7680x540" in c4["partition"]


def test_gpus4_config4_row_tiles_self_launched(shm_rccl):
    rc, d, err = run_bench(["--gpus", "4", *SMALL, "--config4-geometry", "1920x1080", "--config4-frames", "2", "--strong"], clean_env(),
                           timeout=900)
    assert rc == 0, err[-3000:]
    assert d["n_gpus"] == 4 and d["parity"].startswith("bit-exact")
    assert d["launcher"]["attempts"][0]["exit_codes"] == [0] * 4 and d["config"]["rccl"]["ranks_in_communicator"] == 4
    c4 = d["config4"]
    assert c4["parity"] is True and c4["rccl_ranks_in_communicator"] == 4 and c4["rows_of_this_rank"] == [0, 270]


def test_gpus8_under_torch_distributed_run(shm_rccl):
    """The driver's own command line at N = 8: eight supervised ranks, one stdout line, config 4 in eight row tiles."""
    env = clean_env(MASTER_ADDR="127.0.0.1")
    prefix = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1", "--master-port", "29523"]
    rc, d, err = run_bench(["--gpus", "8", *SMALL, "--config4-geometry", "256x64", "--config4-frames", "2", "--strong"], env, prefix=prefix, timeout=900)
    assert rc == 0, err[-3000:]
    assert d["n_gpus"] == 8 and d["config"]["launch"] == "supervised rank under torch.distributed.run"
    assert len(d["launcher"]["attempts"]) == 1 and d["launcher"]["attempts"][0]["exit_codes"] == [0] * 8
    assert d["config"]["rccl"]["ranks_in_communicator"] == 8 and d["parity"].startswith("bit-exact")
    assert d["config4"]["parity"] is True and d["config4"]["rows_of_this_rank"] == [0, 8] and d["config4"]["rccl_ranks_in_communicator"] == 8
    assert d["scaling"] == "strong"


def test_gpus8_one_hung_rank_still_prints_a_line(shm_rccl):
    """Rank 5 of 8 never enters its first collective: seven ranks wait, the watchdogs end the attempt, the next rung carries the line."""
    env = clean_env(SHM_RCCL_HANG_RANK="5", SHM_RCCL_TIMEOUT_S="1000")
    rc, d, err = run_bench(["--gpus", "8", *SMALL, "--no-extras", "--launch-timeout", "40"], env, timeout=900)
    assert rc == 0, err[-3000:]
    attempts = d["launcher"]["attempts"]
    assert len(attempts) == 2 and attempts[0]["failure"] and attempts[1]["failure"] is None and attempts[1]["extra_args"] == ["--torch-reduce"]
    assert d["n_gpus"] == 8 and d["config"]["reduce"].startswith("torch.distributed") and d["parity"].startswith("bit-exact")


def test_dry_run_prints_the_plan_without_a_device(shm_rccl):
    """`bench.py --gpus 8 --dry-run`: one JSON line, every rank's rows / launches / wire bytes / reduce path, no device touched (it
    runs where there is none) -- and the plan agrees with what a real (emulated) N = 2 run then reports."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run"], env=clean_env(), capture_output=True, text=True,
                         timeout=120, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    p = json.loads(lines[0])
    assert p["dry_run"] is True and p["n_gpus"] == 8 and len(p["ranks"]) == 8 and p["scaling"] == "weak"
    assert [r["config4"]["rows"] for r in p["ranks"]] == [[540 * r, 540 * (r + 1)] for r in range(8)]  # BASELINE config 4's row tiles
    assert all(r["headline"]["frames"] == 64 and r["headline"]["slices_per_frame"] == 4 for r in p["ranks"])
    assert "INSIDE the launch" in p["ranks"][0]["headline"]["partial_tables"]
    assert p["table_sum"]["ranks_in_communicator_expected"] == 8 and "ncclAllReduce" in p["table_sum"]["what"]
    assert p["table_sum"]["config4_wire_bytes_per_step"] == 16 * 32768 * 4
    assert p["launch"]["ladder"] == ["C-ABI: scope_table_allreduce (RCCL, dlopen)", "--torch-reduce", "--no-collective"]
    # the same command line shape at N = 2 on the emulation: the plan's rows and communicator size are what the run reports
    args2 = ["--gpus", "2", *SMALL, *C4]
    plan = json.loads(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args2, "--dry-run"], env=clean_env(), capture_output=True,
                                     text=True, timeout=120, cwd=ROOT).stdout)
    rc, d, err = run_bench(args2, clean_env())
    assert rc == 0, err[-3000:]
    assert d["config4"]["rows_of_this_rank"] == plan["ranks"][0]["config4"]["rows"] == [0, 20]
    assert d["config"]["rccl"]["ranks_in_communicator"] == plan["table_sum"]["ranks_in_communicator_expected"] == 2
    # replicas never claim a table sum, a waveform never needs one
    assert json.loads(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--workload", "waveform", "--dry-run"], env=clean_env(),
                                     capture_output=True, text=True, timeout=120, cwd=ROOT).stdout)["table_sum"]["what"].startswith("none")
