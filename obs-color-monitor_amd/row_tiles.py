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
    """Double-buffered, bucketed all-reduce of per-rank tables.

    kind="sat_u8":  u8 tables, saturating at 255 (vectorscope, waveform row tiles)
    kind="sum_u32": u32 tables viewed as int32, plain sums (histogram counts)

    On a GPU the collective and the unpack/clamp pass run on a side stream of their own (with a second scope context
    bound to it): the stream that launches the accumulation kernels only records an event and never waits for a
    collective.  `bucket` steps share one collective -- the tables are tiny (128 KiB each), so what a collective per
    step costs is host time (~100 us of Python + c10d + RCCL launch, as long as a whole 32-frame step): one call per
    `bucket` steps keeps the launch thread ahead of the GPU.  Results are consumed with drain(), which flushes a
    partly filled bucket and makes the caller's stream wait for the side stream.
    On the CPU (gloo, the tests) everything is synchronous and unbucketed.
    """

    def __init__(self, ctx, n_bytes: int, device, kind: str = "sat_u8", group=None, bucket: int = 1):
        import torch

        assert kind in ("sat_u8", "sum_u32") and bucket >= 1
        self.ctx, self.kind, self.group, self.n_bytes = ctx, kind, group, n_bytes
        self.pending = [None, None]
        self.slot = 0
        self.async_side = torch.device(device).type == "cuda"
        self.bucket = bucket if self.async_side else 1
        assert n_bytes % 4 == 0
        self.n_words = n_bytes // 2 if kind == "sat_u8" else n_bytes // 4
        self.wire = [torch.empty(self.bucket * self.n_words, dtype=torch.int32, device=device) for _ in range(2)]
        if kind == "sat_u8":
            self.final = [torch.empty(self.bucket * n_bytes, dtype=torch.uint8, device=device) for _ in range(2)]
        else:
            self.final = self.wire
        if self.async_side:
            self.side = torch.cuda.Stream()
            self.side_ctx = type(ctx)(ctx.device, ctx.lib_path)
            self.side_ctx.set_stream(self.side.cuda_stream)
            self.ready = [torch.cuda.Event() for _ in range(2)]  # the caller's stream has filled wire[s]
            self.done = [torch.cuda.Event() for _ in range(2)]   # the side stream is through with wire[s] / final[s]
            self.in_flight = [False, False]
            self.filled = 0   # tables in the bucket being filled
            self.last = None  # (bucket, index) of the most recent table

    def _wire_slice(self, s, k):
        return self.wire[s][k * self.n_words:(k + 1) * self.n_words]

    def _final_slice(self, s, k):
        if self.kind == "sat_u8":
            return self.final[s][k * self.n_bytes:(k + 1) * self.n_bytes]
        return self._wire_slice(s, k)

    # -- GPU: collective + unpack on the side stream ------------------------------------------------------------
    def _claim(self):
        """The caller's stream may write a bucket again only after the side stream's use of it two buckets ago."""
        import torch

        s = self.slot
        if self.filled == 0 and self.in_flight[s]:
            torch.cuda.current_stream().wait_event(self.done[s])
            self.in_flight[s] = False

    def _table_done(self):
        self.filled += 1
        self.last = (self.slot, self.filled - 1)
        if self.filled == self.bucket:
            self._launch_side()

    def _launch_side(self):
        import torch
        import torch.distributed as dist

        s, count = self.slot, self.filled
        if count == 0:
            return
        self.ready[s].record(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.ready[s])
            work = dist.all_reduce(self.wire[s][:count * self.n_words], group=self.group, async_op=True)
            work.wait()  # the side stream waits for the collective's stream; the host does not block
            if self.kind == "sat_u8":
                self.side_ctx.unpack16_clamp_dev(self.wire[s], self.final[s], count * (self.n_bytes // 2))
            self.done[s].record(self.side)
        self.in_flight[s] = True
        self.filled = 0
        self.slot ^= 1

    def submit(self, table):
        """Queue the reduction of `table` (u8 tensor of n_bytes for sat_u8; u32-as-bytes for sum_u32).
        CPU: returns the finished result of the *previous* submission (or None).  GPU: returns None; see drain()."""
        import torch.distributed as dist

        if self.async_side:
            self._claim()
            dst = self._wire_slice(self.slot, self.filled)
            if self.kind == "sat_u8":
                self.ctx.pack16_dev(table, dst, self.n_bytes // 2)
            else:
                dst.copy_(table.view(-1).view(dst.dtype))
            self._table_done()
            return None
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
        if self.async_side:
            self._claim()
            return self._wire_slice(self.slot, self.filled)
        return self.wire[self.slot]

    def submit_packed(self):
        """Queue the reduction of wire_buffer(), which the caller has just filled on the current stream."""
        import torch.distributed as dist

        if self.async_side:
            self._table_done()
            return None
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
        """Finish everything in flight; returns the most recent result (valid on the caller's stream)."""
        if self.async_side:
            import torch

            self._launch_side()  # a partly filled bucket
            for s in (0, 1):
                if self.in_flight[s]:
                    torch.cuda.current_stream().wait_event(self.done[s])
                    self.in_flight[s] = False
            return None if self.last is None else self._final_slice(*self.last)
        first = self._finish(self.slot)      # older one
        last = self._finish(self.slot ^ 1)   # most recently submitted
        return last if last is not None else first
