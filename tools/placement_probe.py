#!/usr/bin/env python3
"""How much does WHERE a frame pool lies in HBM decide (profiles/HISTORY.md section 5)?  One process, K allocations of the same
64 x 3840x2160 frames; per allocation the vectorscope kernel's own time (64 frames per launch) and the histogram's,
which does not depend on content.  Prints GB/s of each and the rank correlation.

    python tools/placement_probe.py [--k 10] [--hold]     (--hold keeps earlier candidates allocated: distinct ranges)
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import obs_color_monitor_amd as ocm  # noqa: E402
from obs_color_monitor_amd.synth import make_frames  # noqa: E402
from sweep import W, H, timed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--no-hold", action="store_true", help="free each candidate before the next is allocated")
    args = ap.parse_args()
    n = args.frames
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx = ocm.ScopeContext(0)
    ctx.set_stream(stream.cuda_stream)
    rgb, yuv = make_frames(torch, ctx, n, W, H, "uniform", 7)
    del rgb
    torch.cuda.empty_cache()
    out = torch.empty((n, 65536), dtype=torch.uint8, device="cuda")
    outh = torch.empty((n, 4096), dtype=torch.uint8, device="cuda")
    keep, rows = [], []
    for k in range(args.k):
        t = yuv if k == 0 else yuv.clone()
        b = ctx.make_device_batch(t, 4 * W, W, H, planes_are_rgb=False)
        bh = ctx.make_device_batch(t, 4 * W, W, H, planes_are_rgb=True, colorspace=2)
        for _ in range(30):
            ctx.vectorscope_batch_dev(b, out, sync=False)
        ctx.sync()
        kv, _, _ = timed(ctx, lambda: ctx.vectorscope_batch_dev(b, out, sync=False), reps=40)
        kh, _, _ = timed(ctx, lambda: ctx.histogram_counts_batch_dev(bh, 0x07, outh, sync=False), reps=40)
        gv = n * (4 * W * H + 65536) / (kv * 1e-3) / 1e9
        gh = n * (4 * W * H + 4096) / (kh * 1e-3) / 1e9
        rows.append((k, t.data_ptr(), gv, gh))
        print(f"candidate {k:2d}  at {t.data_ptr():#014x}  vectorscope {gv:7.1f} GB/s ({kv * 1e3:6.1f} us)   histogram {gh:7.1f} GB/s", flush=True)
        if args.no_hold and k:
            del t, b, bh
            torch.cuda.empty_cache()
        else:
            keep.append(t)
    gv = [r[2] for r in rows]
    gh = [r[3] for r in rows]
    rank = lambda xs: [sorted(xs).index(x) for x in xs]  # noqa: E731
    rv, rh = rank(gv), rank(gh)
    m = (len(rows) - 1) / 2
    cov = sum((a - m) * (b - m) for a, b in zip(rv, rh))
    var = sum((a - m) ** 2 for a in rv)
    print(f"vectorscope min {min(gv):.1f} max {max(gv):.1f} (spread {100 * (max(gv) / min(gv) - 1):.1f} %); histogram min {min(gh):.1f} max {max(gh):.1f}; "
          f"rank correlation {cov / var:.2f}")


if __name__ == "__main__":
    main()
