#!/usr/bin/env python3
"""Where a vectorscope launch spends its time: per-workgroup time stamps (s_memrealtime, 100 MHz) at the start of the
pixel loop, at its end and after the flush, from the development build `make -C obs-color-monitor_amd/csrc trace`
(tools/libscope_hip_trace.so, -DSCOPE_TRACE_WG).  Prints, for 8 and 64 frames per launch, how far apart the 256
workgroups start and finish.  Run on the GPU box: python tools/wg_trace.py"""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import obs_color_monitor_amd as ocm
from sweep import W, H
from obs_color_monitor_amd.synth import make_frames
lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libscope_hip_trace.so")
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
c = ocm.ScopeContext(0, lib); c.set_stream(stream.cuda_stream)
for n in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("8", "64"))]:
    rgb, yuv = make_frames(torch, c, n, W, H, "uniform", 7)
    out = torch.empty((n, 65536), dtype=torch.uint8, device="cuda")
    b = c.make_device_batch(yuv, 4 * W, W, H, planes_are_rgb=False)
    for _ in range(200): c.vectorscope_batch_dev(b, out, sync=False)
    c.sync()
    res = []
    for rep in range(5):
        if os.environ.get("WG_TRACE_BACK_TO_BACK"):  # the last of a train of launches: what a stream of calls sees
            for _ in range(20):
                c.vectorscope_batch_dev(b, out, sync=False)
        c.vectorscope_batch_dev(b, out, sync=True)
        buf = (ctypes.c_ulonglong * 4096)()
        assert c.L.scope_debug_read(buf) == 0
        nwg = n * max(1, 256 // n) if n < 256 else 256
        t = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 4)[:nwg].astype(np.int64)
        t = t[t[:, 0] > 0]  # a launch with fewer workgroups than nwg (one frame: the slices the launcher chose) leaves the rest unstamped
        t0 = t[:, 0].min()
        us = (t - t0) / 100.0   # 100 MHz
        res.append((us[:, 0].max(), np.median(us[:, 1]), us[:, 1].min(), us[:, 1].max(), us[:, 2].max(), np.percentile(us[:, 1], 10), np.percentile(us[:, 1], 90)))
        durs = globals().setdefault("_durs", [])
        durs.append((us[:, 1] - us[:, 0]).copy())
    # is a workgroup that is slow in one launch slow in the next one too (same frames, same slices)?  correlation of the
    # per-workgroup loop durations between consecutive traced launches, and what is left after taking the mean pattern out
    d = np.array(globals()["_durs"][-5:])
    cc = [float(np.corrcoef(d[i], d[i + 1])[0, 1]) for i in range(len(d) - 1)]
    print("   loop duration per workgroup: correlation between consecutive launches", " ".join(f"{x:.2f}" for x in cc),
          f"; spread of the mean pattern {d.mean(axis=0).std():.2f} us, of the residual {(d - d.mean(axis=0)).std():.2f} us; slowest workgroups (mean): ",
          " ".join(f"{i}:{v:.0f}" for i, v in sorted(enumerate(d.mean(axis=0)), key=lambda t: -t[1])[:8]))
    ends = us[:, 1]  # last launch: when each workgroup left its pixel loop
    starts = us[:, 0]
    print("   by XCD (workgroup id mod 8), mean START [us]:", " ".join(f"{starts[x::8].mean():.1f}" for x in range(8)),
          " loop duration:", " ".join(f"{(ends[x::8] - starts[x::8]).mean():.1f}" for x in range(8)))
    print("   by XCD (workgroup id mod 8), mean end [us]:", " ".join(f"{ends[x::8].mean():.1f}" for x in range(8)))
    sl = max(1, 256 // n)
    print(f"   by slice of its frame (id mod {sl}), mean end [us]:", " ".join(f"{ends[x::sl].mean():.1f}" for x in range(min(sl, 8))))
    r = np.median(np.array(res), axis=0)
    print(f"n={n}: last start +{r[0]:.1f} us; main loop ends median {r[1]:.1f} (min {r[2]:.1f}, p10 {r[5]:.1f}, p90 {r[6]:.1f}, max {r[3]:.1f}); last flush done {r[4]:.1f} us")
    del rgb, yuv, out
