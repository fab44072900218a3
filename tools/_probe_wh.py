import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import obs_color_monitor_amd as ocm
from sweep import W, H, timed
from obs_color_monitor_amd.synth import make_frames
ctx = ocm.ScopeContext(0)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); ctx.set_stream(stream.cuda_stream)
for n in (64, 1, 3):
    rgb, yuv = make_frames(torch, ctx, max(n, 24), W, H, "uniform", 3)
    groups = max(n, 24) // n
    bs = [ctx.make_device_batch(rgb[i*n:(i+1)*n], 4 * W, W, H, planes_are_rgb=True, colorspace=2) for i in range(groups)]
    wv = torch.empty((n, 256 * W * 4), dtype=torch.uint8, device="cuda")
    cnt = torch.empty((n, 1024), dtype=torch.int32, device="cuda")
    it = [0]
    def nxt():
        it[0] += 1
        return bs[it[0] % groups]
    for name, fn in (("waveform alone", lambda: ctx.waveform_batch_dev(nxt(), 0x07, wv, sync=False)),
                     ("histogram alone", lambda: ctx.histogram_counts_batch_dev(nxt(), 0x07, cnt, sync=False)),
                     ("waveform + histogram, one pass", lambda: ctx.waveform_histogram_batch_dev(nxt(), 0x07, 0x07, wv, cnt, sync=False)),
                     ("waveform YUV (fused K0) alone", lambda: ctx.waveform_batch_dev(nxt(), 0x70, wv, sync=False)),
                     ("waveform + histogram YUV, one pass", lambda: ctx.waveform_histogram_batch_dev(nxt(), 0x70, 0x70, wv, cnt, sync=False))):
        for _ in range(20): fn()
        ctx.sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 30 if n == 64 else 400
        e0.record(stream)
        for _ in range(reps): fn()
        e1.record(stream); ctx.sync()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print(json.dumps(dict(frames=n, what=name, us_per_call=round(us, 2), gbps_input=round(n * 4 * W * H / us / 1e3, 1))), flush=True)
    del rgb, yuv
