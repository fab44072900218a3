#!/usr/bin/env python3
"""Where scope_trio_kernel's waves spend their time: per-wave sums of 100 MHz ticks -- waiting at the strip's two barriers, the
read-out, the clamp passes (their barriers included), the whole loop -- from the development build
`make -C obs-color-monitor_amd/csrc trio_variant VARIANT_FLAGS=-DTRIO_TRACE VARIANT_TAG=trace` (tools/_lib_trio_trace.so).
Run on the GPU box: python tools/trio_trace.py [frames [content,content...]]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import obs_color_monitor_amd as ocm  # noqa: E402
from obs_color_monitor_amd.synth import make_frames  # noqa: E402
from sweep import H, W  # noqa: E402

lib = os.path.join(ROOT, "tools", "_lib_trio_trace.so")
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
c = ocm.ScopeContext(0, lib)
c.set_stream(stream.cuda_stream)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for dist in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("uniform", "screen")):
    rgb, yuv = make_frames(torch, c, n, W, H, dist, 7)
    b = c.make_device_batch(rgb, 4 * W, W, H, planes_are_rgb=True, colorspace=2)
    o_vss = torch.empty((n, 65536), dtype=torch.uint8, device="cuda")
    o_wvs = torch.empty((n, 256 * W * 4), dtype=torch.uint8, device="cuda")
    o_his = torch.empty((n, 1024), dtype=torch.int32, device="cuda")
    for _ in range(50):
        c.three_scopes_batch_dev(b, 0x07, 0x07, o_vss, o_wvs, o_his, sync=False)
    c.sync()
    rows = []
    for rep in range(5):
        for _ in range(10):
            c.three_scopes_batch_dev(b, 0x07, 0x07, o_vss, o_wvs, o_his, sync=False)
        c.three_scopes_batch_dev(b, 0x07, 0x07, o_vss, o_wvs, o_his, sync=True)
        buf = (ctypes.c_ulonglong * (256 * 16 * 4))()
        assert c.L.scope_trio_debug_read(buf) == 0
        t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 16, 4).astype(np.float64) / 100.0  # us
        rows.append(t)
    t = np.stack(rows)  # [rep][workgroup][wave][total, barrier, readout, clamp]
    tot, bar, ro, cl = (t[..., k] for k in range(4))
    print(f"{dist:8s} {n} frames: loop per wave  median {np.median(tot):7.1f} us  (min {tot.min():.1f}, max {tot.max():.1f});  "
          f"barrier waits {np.median(bar):6.1f} us = {100 * np.median(bar / tot):4.1f} %  (wave max {100 * (bar / tot).max():.1f} %);  "
          f"read-outs {np.median(ro):6.1f} us = {100 * np.median(ro / tot):4.1f} %;  clamp passes {np.median(cl):6.1f} us = {100 * np.median(cl / tot):4.1f} %;  "
          f"slots (the rest) {100 * np.median((tot - bar - ro - cl) / tot):4.1f} %", flush=True)
    # per wave index (16 waves of a workgroup): who waits -- a wave that waits little is the one the others wait for
    share = np.nan_to_num(bar / np.maximum(tot, 1e-9))
    print("         barrier share by wave index: " + " ".join(f"{100 * np.median(share[:, :, w]):4.1f}" for w in range(16)))
    print("         slot time by wave index, us: " + " ".join(f"{np.median((tot - bar - ro - cl)[:, :, w]):5.0f}" for w in range(16)))
    # how far apart the waves of a workgroup arrive: the fastest wave's barrier share against the slowest's
    wmin, wmax = (bar / tot).min(axis=2), (bar / tot).max(axis=2)
    print(f"         barrier share of a workgroup's waves: least {100 * np.median(wmin):.1f} % ... most {100 * np.median(wmax):.1f} %")
    del rgb, yuv
