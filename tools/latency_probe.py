#!/usr/bin/env python3
"""The reference's real call shape: ONE surface per callback (src/common.c:366-368), ring depth 3 (src/common.h:46).

For launches of 1 and 3 frames of 3840x2160 (device-resident, rotating over a pool of distinct frames that is larger
than the 256 MB Infinity Cache) this prints, per scope, the GPU time of one call in steady state -- calls enqueued back
to back on one stream, so the figure is everything the stream has to do for a call: accumulation kernel, the
partial-table reduction where there is one, and the boundaries between them -- next to the accumulation kernel's own
HIP-event time, and checks one table per configuration against the oracle.

    python tools/latency_probe.py [--lib path] [--option k=v ...] [--out file.jsonl]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import obs_color_monitor_amd as ocm  # noqa: E402

W, H = 3840, 2160


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--pool", type=int, default=24, help="distinct frames (24 x 33 MB = 800 MB)")
    ap.add_argument("--dists", default="uniform,natural")
    ap.add_argument("--frames", default="1,3")
    ap.add_argument("--scopes", default="vectorscope,vectorscope_rgb,histogram,waveform")
    ap.add_argument("--calls", type=int, default=600)
    ap.add_argument("--option", action="append", default=[])
    ap.add_argument("--sweep", default="", help="option sets 'k=v,k=v;k=v;...' each measured on the same frames (after the plain run)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from parity_cases import DEFAULTS
    sets = [[]] + [x.split(",") for x in args.sweep.split(";") if x]

    import scope_oracle
    from obs_color_monitor_amd.synth import make_frames

    o = scope_oracle.COracle()
    ctx = ocm.ScopeContext(0, args.lib)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    for kv in args.option:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    out = open(args.out, "w") if args.out else None
    pool = args.pool

    for dist in args.dists.split(","):
        rgb, yuv = make_frames(torch, ctx, pool, W, H, dist, 11)
        host_rgb0, host_yuv0 = rgb[0].cpu().numpy(), yuv[0].cpu().numpy()
        for nf in [int(x) for x in args.frames.split(",")]:
            groups = pool // nf
            for scope, optset in [(sc, os_) for sc in args.scopes.split(",") for os_ in sets]:
                if optset and not any(kv.split("=")[0].split(".")[0] in ("slices", "nt", {"vectorscope": "vss", "vectorscope_rgb": "vss", "histogram": "his", "waveform": "wvs", "three_scope": "dock"}[scope], "fold", "dock")
                                      for kv in optset):
                    continue  # this option set does not concern this scope
                for kv in optset:
                    k, v = kv.split("=")
                    ctx.set_option(k, int(v))
                if scope == "vectorscope":
                    batches = [ctx.make_device_batch(yuv[i * nf:(i + 1) * nf], 4 * W, W, H, planes_are_rgb=False) for i in range(groups)]
                    tbytes, algo = 65536, 4 * W * H + 65536
                else:
                    batches = [ctx.make_device_batch(rgb[i * nf:(i + 1) * nf], 4 * W, W, H, planes_are_rgb=True, colorspace=2) for i in range(groups)]
                    tbytes = {"vectorscope_rgb": 65536, "histogram": 4096, "waveform": 256 * W * 4, "three_scope": 65536}[scope]
                    algo = 4 * W * H + tbytes
                tabs = torch.empty((groups, nf, tbytes), dtype=torch.uint8, device="cuda")
                if scope == "three_scope":  # the dock's three scopes (scope_three_scopes_batch); the vectorscope table is checked
                    algo = 4 * W * H + 65536 + 256 * W * 4 + 4096
                    wtab = torch.empty((2, nf, 256 * W * 4), dtype=torch.uint8, device="cuda")
                    htab = torch.empty((2, nf, 1024), dtype=torch.int32, device="cuda")

                def call(i):
                    g = i % groups
                    if scope == "three_scope":
                        ctx.three_scopes_batch_dev(batches[g], 0x07, 0x07, tabs[g], wtab[i & 1], htab[i & 1], sync=False)
                    elif scope.startswith("vectorscope"):
                        ctx.vectorscope_batch_dev(batches[g], tabs[g], sync=False)
                    elif scope == "histogram":
                        ctx.histogram_counts_batch_dev(batches[g], 0x07, tabs[g], sync=False)
                    else:
                        ctx.waveform_batch_dev(batches[g], 0x07, tabs[g], sync=False)

                t_end = time.perf_counter() + 0.4
                i = 0
                while time.perf_counter() < t_end:
                    for _ in range(32):
                        call(i)
                        i += 1
                    ctx.sync()
                # steady state, back to back; GPU time from stream events around the whole run
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ctx.sync()
                t0 = time.perf_counter()
                e0.record(stream)
                for i in range(args.calls):
                    call(i)
                e1.record(stream)
                ctx.sync()
                wall = (time.perf_counter() - t0) / args.calls
                gpu = e0.elapsed_time(e1) * 1e-3 / args.calls
                # the accumulation kernel alone (bracketed by events inside the C-ABI): a separate, shorter run
                ctx.profile(True)
                for i in range(200):
                    call(i)
                ctx.sync()
                n, ms = ctx.kernel_ms()
                ctx.profile(False)
                kern = ms * 1e-3 / max(n, 1)
                # parity of frame 0 (group 0)
                call(0)
                ctx.sync()
                got = tabs[0, 0].cpu().numpy()
                if scope.startswith("vectorscope") or scope == "three_scope":
                    ok = np.array_equal(got.reshape(256, 256), o.vectorscope(host_yuv0, W))
                elif scope == "histogram":
                    ok = np.array_equal(got.view(np.uint32).reshape(256, 4), o.histogram_counts(host_rgb0, host_yuv0, W, 0x07))
                else:
                    ok = np.array_equal(got.reshape(256, W, 4), o.waveform(host_rgb0, host_yuv0, W, 0x07))
                rec = dict(scope=scope, dist=dist, frames_per_call=nf, us_per_call_stream=round(gpu * 1e6, 2),
                           us_per_call_host=round(wall * 1e6, 2), us_accumulation_kernel_events=round(kern * 1e6, 2),
                           gbps_call=round(nf * algo / gpu / 1e9, 1), frac_call=round(nf * algo / gpu / 8e12, 3),
                           parity=bool(ok), options=args.option + optset)
                print(json.dumps(rec), flush=True)
                if out:
                    out.write(json.dumps(rec) + "\n")
                    out.flush()
                del tabs, batches
                for kv in optset:
                    k = kv.split("=")[0]
                    ctx.set_option(k, DEFAULTS.get(k, 0))
        del rgb, yuv
    ctx.close()


if __name__ == "__main__":
    main()
