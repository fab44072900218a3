#!/usr/bin/env python3
"""Same-process A/B of two builds of libscope_hip.so on the SAME HBM-resident frames.

Between processes the streaming kernels land in one of two placement modes 3 % apart (profiles/HISTORY.md section 5), which
drowns the 1-2 % effects kernel tuning is about; here both libraries read the very same buffers, alternately.

    python tools/ab.py /path/to/libA.so /path/to/libB.so [more libraries ...] [--workload vectorscope|vectorscope_rgb|histogram|histogram_yuv|waveform|waveform_yuv|three_scope]
(the *_rgb / *_yuv forms read RGB planes and convert in the kernel: the fused-K0 path; a content name with "+pad" behind it is
the same frames at linesize 4 W + 256; --opts "k=v,k=v;;k=v" gives the i-th library its options)
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
from sweep import W, H, timed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--workload", default="vectorscope")
    ap.add_argument("--dists", default="uniform,natural,screen")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--frames", type=int, default=32, help="frames per launch (bench.py's default is 64)")
    ap.add_argument("--comp", type=lambda x: int(x, 0), default=None, help="component mask of the waveform / histogram workloads (default 0x07, 0x70 for *_yuv)")
    ap.add_argument("--opt-a", action="append", default=[], help="key=value for the first library's context")
    ap.add_argument("--opt-b", action="append", default=[], help="key=value for the second (the same .so twice compares options)")
    ap.add_argument("--whole", action="store_true", help="rate of the WHOLE call (every kernel it enqueues and the gaps between them: wall clock over 60 calls "
                    "between synchronizes, no profiling), not of the accumulation kernel alone")
    ap.add_argument("--opts", default="", help="options per library, ';'-separated lists of k=v[,k=v] (overrides --opt-a / --opt-b)")
    args = ap.parse_args()
    per_lib = [[kv for kv in part.split(",") if kv] for part in args.opts.split(";")] if args.opts else [args.opt_a, args.opt_b]
    per_lib += [[]] * (len(args.libs) - len(per_lib))
    n = args.frames
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctxs = []
    for p in args.libs:
        c = ocm.ScopeContext(0, os.path.abspath(p))
        c.set_stream(stream.cuda_stream)
        for kv in per_lib[len(ctxs)]:
            k, v = kv.split("=")
            c.set_option(k, int(v))
        ctxs.append(c)
    from obs_color_monitor_amd.synth import make_frames

    wl = args.workload
    table = {"vectorscope": 65536, "vectorscope_rgb": 65536, "histogram": 4096, "histogram_yuv": 4096,
             "waveform": 256 * W * 4, "waveform_yuv": 256 * W * 4, "three_scope": 65536 + 4096 + 256 * W * 4}[wl]
    comp = args.comp if args.comp is not None else (0x70 if wl.endswith("_yuv") else 0x07)
    out = torch.empty((n, table), dtype=torch.uint8, device="cuda")
    if wl == "three_scope":
        o_vss = torch.empty((n, 65536), dtype=torch.uint8, device="cuda")
        o_wvs = torch.empty((n, 256 * W * 4), dtype=torch.uint8, device="cuda")
        o_his = torch.empty((n, 1024), dtype=torch.int32, device="cuda")
    for d in args.dists.split(","):
        rgb, yuv = make_frames(torch, ctxs[0], n, W, H, d.split("+")[0], 7)
        ls = 4 * W
        if d.endswith("+pad"):
            from obs_color_monitor_amd.synth import pad_pitch

            rgb, yuv = pad_pitch(torch, rgb, W), pad_pitch(torch, yuv, W)
            ls = 4 * W + 256
        res = [[] for _ in args.libs]
        for r in range(args.rounds):
            for i, c in enumerate(ctxs):
                if wl == "vectorscope":
                    b = c.make_device_batch(yuv, ls, W, H, planes_are_rgb=False)
                    fn = lambda: c.vectorscope_batch_dev(b, out, sync=False)  # noqa: E731
                elif wl == "vectorscope_rgb":
                    b = c.make_device_batch(rgb, ls, W, H, planes_are_rgb=True, colorspace=2)
                    fn = lambda: c.vectorscope_batch_dev(b, out, sync=False)  # noqa: E731
                elif wl == "three_scope":
                    b = c.make_device_batch(rgb, ls, W, H, planes_are_rgb=True, colorspace=2)
                    fn = lambda: c.three_scopes_batch_dev(b, comp, comp, o_vss, o_wvs, o_his, sync=False)  # noqa: E731
                elif wl.startswith("histogram"):
                    b = c.make_device_batch(rgb, ls, W, H, planes_are_rgb=True, colorspace=2)
                    fn = lambda: c.histogram_counts_batch_dev(b, comp, out, sync=False)  # noqa: E731
                else:
                    b = c.make_device_batch(rgb, ls, W, H, planes_are_rgb=True, colorspace=2)
                    fn = lambda: c.waveform_batch_dev(b, comp, out, sync=False)  # noqa: E731
                if r == 0:
                    t_end = time.perf_counter() + 0.5
                    while time.perf_counter() < t_end:
                        fn()
                        c.sync()
                if args.whole:
                    for _ in range(5):
                        fn()
                    c.sync()
                    t0 = time.perf_counter()
                    for _ in range(60):
                        fn()
                    c.sync()
                    k = (time.perf_counter() - t0) / 60 * 1e3
                else:
                    k, wall, _ = timed(c, fn, reps=30)
                res[i].append(n * (4 * W * H + table) / (k * 1e-3) / 1e9)
        for i, p in enumerate(args.libs):
            r = sorted(res[i])
            print(f"{wl:16s} {d:12s} {os.path.basename(p) + ' ' + ','.join(per_lib[i]):36s} median {r[len(r) // 2]:7.1f} GB/s   all " +
                  " ".join(f"{x:.0f}" for x in res[i]), flush=True)
        del rgb, yuv


if __name__ == "__main__":
    main()
