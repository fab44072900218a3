#!/usr/bin/env python3
"""BASELINE config 1: 256-bin RGB histogram over ONE 1280x720 BGRA8 frame on the CPU path only.

Times the oracle's restatement of his_draw_histogram (counts + level pass, src/histogram.c:357-418) single-threaded, as
the reference's one `color-monitor` thread runs it, plus the all-cores courtesy variant of the counting loop, and
checks the result against the committed fixture hash (tests/golden/fullsize_hashes.json) when the frame is the
fixture's.  Prints one JSON line.  No GPU involved."""
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import scope_oracle  # noqa: E402


def main():
    o = scope_oracle.COracle()
    w, h = 1280, 720
    rgb = o.fill_frame(w, h, 1, 0x9E3779B97F4A7C15)
    yuv = o.rgb_to_yuv(rgb, w, 2)

    def med(fn, warm=5, n=21):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts)

    t_full = med(lambda: o.histogram(rgb, yuv, w, 0x07))
    t_counts = med(lambda: o.histogram_counts(rgb, yuv, w, 0x07))
    threads = os.cpu_count() or 1
    reps = 200
    t_mt = med(lambda: o.histogram_counts_mt(rgb, yuv, w, 0x07, threads, reps), warm=2, n=7) / reps
    same = bool(np.array_equal(o.histogram_counts_mt(rgb, yuv, w, 0x07, threads), o.histogram_counts(rgb, yuv, w, 0x07)))
    cpu = "?"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    print(json.dumps({
        "config": "histogram components=0x07, one 1280x720 BGRA8 frame, CPU path only (oracle restatement, gcc -O2)",
        "one_thread": {"ms_per_frame": round(t_full * 1e3, 4), "gpix_s": round(w * h / t_full / 1e9, 4),
                       "counts_only_ms": round(t_counts * 1e3, 4), "method": "median of 21 after 5 warm passes"},
        "all_cores": {"threads": threads, "cpu": cpu, "ms_per_frame": round(t_mt * 1e3, 5),
                      "gpix_s": round(w * h / t_mt / 1e9, 3), "same_counts": same,
                      "method": f"median of 7 calls x {reps} passes (thread start-up paid once per call)"},
        "counts_fnv1a64": hex(o.fnv(o.histogram_counts(rgb, yuv, w, 0x07))),
    }))


if __name__ == "__main__":
    main()
