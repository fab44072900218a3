"""Multi-GPU glue for row-tiled frames: one process per GPU, tables summed over RCCL.

A frame split into row tiles gives every rank a *clamped* u8 partial table (vectorscope) or a plain
u32 count table (histogram).  Saturating addition of clamped partials is exact --
min(min(a,255)+min(b,255),255) == min(a+b,255) -- so the ranks pack two bins per int32 word
(scope_table_pack_u8_u16x2; halves cannot carry for <= 257 ranks), all-reduce(sum) over
``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests) and
clamp once (scope_table_unpack_clamp_u16x2_u8).  The payload is tiny (128 KiB per 256x256 table), so
the collective is latency-bound; it is double-buffered here so that the reduction of step i overlaps
the accumulation of step i+1.

The reference has no distributed path; this is the build's own row-tile decomposition.
Waveform needs no collective at all: ranks own column strips and simply keep their slab.
"""
from __future__ import annotations


def tile_rows(height: int, rank: int, world: int):
    """Rows [r0, r1) of the frame owned by `rank` (same split the kernels use for slices)."""
    return height * rank // world, height * (rank + 1) // world


def tile_cols(width: int, rank: int, world: int, align: int = 64):
    """Columns [c0, c1) of the frame owned by `rank` for the waveform: whole 64-column strips (the kernel's unit), so
    every rank's slab is produced without partial strips.  A column's counters depend on that column only, so the
    slabs are final: no collective, just concatenate (all_gather) if one owner needs the whole table."""
    strips = (width + align - 1) // align
    s0, s1 = strips * rank // world, strips * (rank + 1) // world
    return min(s0 * align, width), min(s1 * align, width)


class TableAllReduce:
    """Double-buffered all-reduce of per-rank tables.

    kind="sat_u8":  u8 tables, saturating at 255 (vectorscope, waveform row tiles)
    kind="sum_u32": u32 tables viewed as int32, plain sums (histogram counts)
    """

    def __init__(self, ctx, n_bytes: int, device, kind: str = "sat_u8", group=None):
        import torch

        assert kind in ("sat_u8", "sum_u32")
        self.ctx, self.kind, self.group, self.n_bytes = ctx, kind, group, n_bytes
        self.pending = [None, None]
        self.slot = 0
        if kind == "sat_u8":
            assert n_bytes % 2 == 0
            self.wire = [torch.empty(n_bytes // 2, dtype=torch.int32, device=device) for _ in range(2)]
            self.final = [torch.empty(n_bytes, dtype=torch.uint8, device=device) for _ in range(2)]
        else:
            assert n_bytes % 4 == 0
            self.wire = [torch.empty(n_bytes // 4, dtype=torch.int32, device=device) for _ in range(2)]
            self.final = self.wire

    def submit(self, table):
        """Queue the reduction of `table` (u8 tensor of n_bytes for sat_u8; u32-as-bytes for sum_u32).
        Returns the finished result of the *previous* submission (or None)."""
        import torch.distributed as dist

        s = self.slot
        if self.kind == "sat_u8":
            self.ctx.pack16_dev(table, self.wire[s], self.n_bytes // 2)
        else:
            self.wire[s].copy_(table.view(-1).view(self.wire[s].dtype))
        done = self._finish(s ^ 1)
        self.pending[s] = dist.all_reduce(self.wire[s], group=self.group, async_op=True)
        self.slot ^= 1
        return done

    def wire_buffer(self):
        """sat_u8 only: the int32 buffer (two bins per word) the next submit_packed() will reduce.  Let the scope
        kernels write it directly (vectorscope_batch_dev(..., packed16=True)) and the packing pass disappears."""
        assert self.kind == "sat_u8"
        return self.wire[self.slot]

    def submit_packed(self):
        """Queue the reduction of wire_buffer(), which the caller has just filled on the current stream."""
        import torch.distributed as dist

        s = self.slot
        done = self._finish(s ^ 1)
        self.pending[s] = dist.all_reduce(self.wire[s], group=self.group, async_op=True)
        self.slot ^= 1
        return done

    def _finish(self, s):
        if self.pending[s] is None:
            return None
        self.pending[s].wait()
        self.pending[s] = None
        if self.kind == "sat_u8":
            self.ctx.unpack16_clamp_dev(self.wire[s], self.final[s], self.n_bytes // 2)
        return self.final[s]

    def drain(self):
        """Finish everything in flight; returns the most recent result."""
        first = self._finish(self.slot)      # older one
        last = self._finish(self.slot ^ 1)   # most recently submitted
        return last if last is not None else first
