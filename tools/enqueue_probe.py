#!/usr/bin/env python3
"""Host-side cost of the bench's step: per-call enqueue times of scope_vectorscope_batch over 64 resident 4K frames from an
idle stream (what the first of bench.py's K timed steps pays before the GPU has anything to do), and the final synchronize.

    python tools/enqueue_probe.py [--frames 64] [--calls 20]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402

import obs_color_monitor_amd as ocm  # noqa: E402
from sweep import W, H  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--calls", type=int, default=20)
    args = ap.parse_args()
    n = args.frames
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    c = ocm.ScopeContext(0)
    c.set_stream(stream.cuda_stream)
    from obs_color_monitor_amd.synth import make_frames

    rgb, yuv = make_frames(torch, c, n, W, H, "uniform", 7)
    out = torch.empty((n, 65536), dtype=torch.uint8, device="cuda")
    b = c.make_device_batch(yuv, 4 * W, W, H, planes_are_rgb=False)
    for _ in range(200):
        c.vectorscope_batch_dev(b, out, sync=False)
    torch.cuda.synchronize()
    for rep in range(4):
        ts = [time.perf_counter()]
        for _ in range(args.calls):
            c.vectorscope_batch_dev(b, out, sync=False)
            ts.append(time.perf_counter())
        torch.cuda.synchronize()
        t_end = time.perf_counter()
        d = [(ts[i + 1] - ts[i]) * 1e6 for i in range(args.calls)]
        print(f"rep {rep}: enqueue us per call: " + " ".join(f"{x:.0f}" for x in d) +
              f" | all enqueued after {(ts[-1] - ts[0]) * 1e3:.3f} ms, synchronized after {(t_end - ts[0]) * 1e3:.3f} ms"
              f" = {(t_end - ts[0]) * 1e3 / args.calls:.4f} ms per call", flush=True)
        time.sleep(0.05)
    # the ramp after an idle period (bench.py checks tables on the host between its legs: the GPU idles for seconds):
    # chunks of 20 calls, synchronized, for 2.5 s
    for idle in (3.0, 0.5):
        time.sleep(idle)
        t_start = time.perf_counter()
        line = []
        while time.perf_counter() - t_start < 2.5:
            t0 = time.perf_counter()
            for _ in range(args.calls):
                c.vectorscope_batch_dev(b, out, sync=False)
            torch.cuda.synchronize()
            line.append((t0 - t_start, (time.perf_counter() - t0) * 1e3 / args.calls))
        print(f"after {idle} s idle, ms per call of successive {args.calls}-call chunks (at s): " +
              " ".join(f"{m:.4f}@{t:.2f}" for t, m in line[:12]) + " ... " + " ".join(f"{m:.4f}@{t:.2f}" for t, m in line[-4:]), flush=True)
        tail = sorted(m for _, m in line[len(line) // 2:])
        print(f"   median of the second half: {tail[len(tail) // 2]:.4f}; chunks above 1.01 x that: " +
              " ".join(f"{t:.2f}" for t, m in line if m > 1.01 * tail[len(tail) // 2])[:400], flush=True)
    # the same K calls after the queue was primed with one untimed call (the GPU busy when the clock starts)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(args.calls):
        c.vectorscope_batch_dev(b, out, sync=False)
    e1.record(stream)
    torch.cuda.synchronize()
    print(f"events around {args.calls} calls: {e0.elapsed_time(e1) / args.calls:.4f} ms per call")


if __name__ == "__main__":
    main()
