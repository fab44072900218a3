#!/usr/bin/env python3
"""BASELINE config 5: a stream of 3840x2160 frames, histogram + waveform + vectorscope per frame on one
MI355X with upload / compute / download hipStreams overlapped (scope_pipe_* of the C-ABI).

Frames start in PINNED HOST memory (what a capture path would fill); every frame crosses PCIe, so this
is the PCIe-inclusive rate of the whole drop-in path -- not bench.py's HBM-resident kernel rate.
Reports sustained fps, Gpix/s, mean/max per-frame latency (submit -> tables on the host) and checks
sampled frames bit-exactly against the oracle.  One JSON line on stdout.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import obs_color_monitor_amd as ocm  # noqa: E402
import scope_oracle  # noqa: E402
from obs_color_monitor_amd.synth import make_rgb_frames  # noqa: E402

W, H = 3840, 2160


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=960, help="frames streamed in the timed region")
    ap.add_argument("--sources", type=int, default=16, help="distinct pinned source frames rotated through")
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--dist", default="natural")
    ap.add_argument("--target-fps", type=float, default=0.0, help="0 = as fast as possible; 240 = paced like the config")
    args = ap.parse_args()

    ctx = ocm.ScopeContext(0)
    info = ctx.device_info()
    rgb = make_rgb_frames(torch, args.sources, W, H, args.dist, 7)
    torch.cuda.synchronize()
    pinned = torch.empty((args.sources, H, W * 4), dtype=torch.uint8, pin_memory=True)
    pinned.copy_(rgb)
    torch.cuda.synchronize()
    del rgb
    ptrs = [int(pinned[i].data_ptr()) for i in range(args.sources)]

    pipe = ocm.ScopePipe(ctx, W, H, depth=args.depth, colorspace=2, wvs_components=0x07, his_components=0x07)
    o = scope_oracle.COracle()

    def run(n, check_every=0, pace=0.0):
        lat, submit_t, done, results = [], {}, 0, {}
        t0 = time.perf_counter()
        for i in range(n):
            if pace:
                while time.perf_counter() - t0 < i / pace:
                    pass
            while True:
                submit_t[i] = time.perf_counter()
                if pipe.submit(ptrs[i % args.sources], 4 * W):
                    if pace:  # paced like a live source: take each frame's tables as soon as they exist
                        got = pipe.collect()
                        lat.append(time.perf_counter() - submit_t.pop(done))
                        if check_every and done % check_every == 0:
                            results[done] = tuple(np.array(t) for t in got)
                        done += 1
                    break
                got = pipe.collect()
                lat.append(time.perf_counter() - submit_t.pop(done))
                if check_every and done % check_every == 0:
                    results[done] = tuple(np.array(t) for t in got)
                done += 1
        while done < n:
            got = pipe.collect()
            lat.append(time.perf_counter() - submit_t.pop(done))
            if check_every and done % check_every == 0:
                results[done] = tuple(np.array(t) for t in got)
            done += 1
        return time.perf_counter() - t0, lat, results

    run(3 * args.depth + 8)  # warm: allocations, clocks
    elapsed, lat, results = run(args.frames, check_every=max(1, args.frames // 4), pace=args.target_fps)

    parity = True
    for idx, (v, wv, hs) in results.items():
        src = pinned[idx % args.sources].numpy()
        yuv = o.rgb_to_yuv(src, W, 2)
        parity &= np.array_equal(v, o.vectorscope(yuv, W))
        parity &= np.array_equal(wv, o.waveform(src, yuv, W, 0x07))
        parity &= np.array_equal(hs, o.histogram_counts(src, yuv, W, 0x07))

    fps = args.frames / elapsed
    out = {
        "metric": "sustained frames/s, 3840x2160, histogram + waveform + vectorscope per frame, PCIe-inclusive",
        "value": round(fps, 1),
        "unit": "frames/s",
        "gpix_per_s": round(fps * W * H / 1e9, 2),
        "h2d_gbytes_per_s": round(fps * W * H * 4 / 1e9, 2),
        "latency_ms_mean": round(1e3 * sum(lat) / len(lat), 3),
        "latency_ms_max": round(1e3 * max(lat), 3),
        "frames": args.frames,
        "frames_in_flight": args.depth,
        "target_fps": args.target_fps,
        "meets_240fps": fps >= 240.0,
        "parity": "bit-exact vs oracle on %d sampled frames" % len(results) if parity else "MISMATCH",
        "input": f"{args.sources} distinct pinned host frames ({args.dist}), RGB plane only, K0 fused on the GPU",
        "device": info["name"],
    }
    print(json.dumps(out), flush=True)
    pipe.close()
    ctx.close()
    sys.exit(0 if parity else 3)


if __name__ == "__main__":
    main()
