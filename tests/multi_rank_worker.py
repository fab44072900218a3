"""Worker for tests/test_multi_rank.py: one rank of a row-tiled frame, gloo backend, emulated kernels.

Launched by torch.distributed.run with --nproc-per-node 2.  Each rank reduces ITS row tile of the same
seeded frame through the C-ABI (CPU emulation build), the tables are all-reduced with the package's
TableAllReduce (the code bench.py uses over RCCL), and every rank compares with the oracle's table of the
WHOLE frame."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import obs_color_monitor_amd as ocm  # noqa: E402
import scope_oracle  # noqa: E402
from obs_color_monitor_amd.row_tiles import TableAllReduce, tile_cols, tile_rows  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    o = scope_oracle.COracle()
    ctx = ocm.ScopeContext(0, os.path.join(ROOT, "tests", "emul", "libscope_emul.so"))
    w, h = 96, 50
    red = TableAllReduce(ctx, 65536, "cpu", "sat_u8")
    red32 = TableAllReduce(ctx, 4096, "cpu", "sum_u32")
    results, results32, wants, wants32 = [], [], [], []
    for step, dist_kind in enumerate((3, 2, 1)):  # three frames through the double-buffered reducer
        rgb = o.fill_frame(w, h, dist_kind, 5 + step)
        yuv = o.rgb_to_yuv(rgb, w, 2)
        wants.append(o.vectorscope(yuv, w))
        wants32.append(o.histogram_counts(rgb, yuv, w, 0x07))
        r0, r1 = tile_rows(h, rank, world)
        part, st = ctx.vectorscope(yuv=np.ascontiguousarray(yuv[r0:r1]), width=w)
        assert st == 0
        done = red.submit(torch.from_numpy(part.reshape(-1)))
        if done is not None:
            results.append(done.clone())
        cnt, st = ctx.histogram_counts(rgb=np.ascontiguousarray(rgb[r0:r1]), yuv=None, width=w, components=0x07)
        assert st == 0
        done = red32.submit(torch.from_numpy(cnt.view(np.int32).reshape(-1)))
        if done is not None:
            results32.append(done.clone())
    results.append(red.drain().clone())
    results32.append(red32.drain().clone())
    # the same sum with the tables written straight into the reducer's wire buffer (what bench.py does on the GPUs)
    redp = TableAllReduce(ctx, 65536, "cpu", "sat_u8")
    packed_results = []
    for step, dist_kind in enumerate((3, 2, 1)):
        rgb = o.fill_frame(w, h, dist_kind, 5 + step)
        yuv = o.rgb_to_yuv(rgb, w, 2)
        r0, r1 = tile_rows(h, rank, world)
        tile = torch.from_numpy(np.ascontiguousarray(yuv[r0:r1])).unsqueeze(0)
        if step == 1:
            ctx.set_option("slices", 1)  # a single partial per frame: the "reduction" is only the repacking
        batch = ctx.make_device_batch(tile, tile.shape[2], w, r1 - r0, planes_are_rgb=False)
        ctx.vectorscope_batch_dev(batch, redp.wire_buffer(), packed16=True)
        ctx.set_option("slices", 0)
        done = redp.submit_packed()
        if done is not None:
            packed_results.append(done.clone())
    packed_results.append(redp.drain().clone())
    for got, want in zip(packed_results, wants):
        assert np.array_equal(got.numpy().reshape(256, 256), want), "packed-wire row-tiled vectorscope != whole frame"
    assert len(results) == 3 and len(results32) == 3
    for got, want in zip(results, wants):
        assert np.array_equal(got.numpy().reshape(256, 256), want), "row-tiled vectorscope != whole frame"
    for got, want in zip(results32, wants32):
        assert np.array_equal(got.numpy().view(np.uint32).reshape(256, 4), want), "row-tiled histogram != whole frame"
    # waveform: ranks own column strips and need no reduction at all
    w2, h2 = 200, 30  # 4 strips of 64 columns, the last one ragged
    rgb = o.fill_frame(w2, h2, 5, 77)
    yuv = o.rgb_to_yuv(rgb, w2, 2)
    c0, c1 = tile_cols(w2, rank, world)
    slabs = [None] * world
    mine, st = ctx.waveform(rgb=np.ascontiguousarray(rgb[:, 4 * c0:4 * c1]), yuv=np.ascontiguousarray(yuv[:, 4 * c0:4 * c1]),
                            width=c1 - c0, components=0x07)
    assert st == 0
    dist.all_gather_object(slabs, (c0, c1, mine))
    whole = np.zeros((256, w2, 4), dtype=np.uint8)
    for a, b, slab in slabs:
        whole[:, a:b] = slab
    assert sorted((a, b) for a, b, _ in slabs)[0][0] == 0 and max(b for _, b, _ in slabs) == w2
    assert np.array_equal(whole, o.waveform(rgb, yuv, w2, 0x07)), "column-tiled waveform != whole frame"
    # BASELINE config 4's geometry: a 7680-pixel-wide frame (a few rows of it: the emulator is slow) in row tiles through
    # the packed wire format bench.py's config4 leg uses, and the waveform of the same frame by column slab
    w4, h4 = 7680, 12
    rgb = o.fill_frame(w4, h4, 2, 404)
    yuv = o.rgb_to_yuv(rgb, w4, 2)
    r0, r1 = tile_rows(h4, rank, world)
    red4 = TableAllReduce(ctx, 65536, "cpu", "sat_u8")
    tile = torch.from_numpy(np.ascontiguousarray(yuv[r0:r1])).unsqueeze(0)
    batch = ctx.make_device_batch(tile, tile.shape[2], w4, r1 - r0, planes_are_rgb=False)
    ctx.vectorscope_batch_dev(batch, red4.wire_buffer(), packed16=True)
    red4.submit_packed()
    got = red4.drain()
    assert np.array_equal(got.numpy().reshape(256, 256), o.vectorscope(yuv, w4)), "7680-wide row tiles != whole frame"
    c0, c1 = tile_cols(w4, rank, world)
    assert (c1 - c0) % 64 == 0 and c1 - c0 == w4 // world
    mine, st = ctx.waveform(rgb=np.ascontiguousarray(rgb[:, 4 * c0:4 * c1]), yuv=None, width=c1 - c0, components=0x07)
    assert st == 0 and np.array_equal(mine, o.waveform(rgb, yuv, w4, 0x07)[:, c0:c1]), "7680-wide column slab != whole frame's columns"
    # a flat frame saturates on every rank: clamped partials 255 + 255 must still give 255
    assert wants[0].max() == 255
    dist.barrier()
    if rank == 0:
        print("MULTI_RANK_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
