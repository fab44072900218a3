"""world_size-2 run of the row-tile path on CPU: gloo + the emulated kernels (no GPU needed)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_row_tiles_two_ranks_gloo(emul_lib):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "tests", "multi_rank_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "MULTI_RANK_OK" in out.stdout


def test_tile_rows_cover_the_frame():
    sys.path.insert(0, ROOT)
    from obs_color_monitor_amd.row_tiles import tile_rows

    for h in (1, 7, 540, 2160, 4320):
        for world in (1, 2, 3, 8):
            edges = [tile_rows(h, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == h
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
