import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import obs_color_monitor_amd as ocm
from obs_color_monitor_amd.synth import make_frames
W, H = 3840, 2160
ctx = ocm.ScopeContext(0)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); ctx.set_stream(stream.cuda_stream)
rgb, yuv = make_frames(torch, ctx, 24, W, H, "uniform", 3)
for nf in (1, 3):
    groups = 24 // nf
    bs = [ctx.make_device_batch(yuv[i*nf:(i+1)*nf], 4 * W, W, H, planes_are_rgb=False) for i in range(groups)]
    br = [ctx.make_device_batch(rgb[i*nf:(i+1)*nf], 4 * W, W, H, planes_are_rgb=True, colorspace=2) for i in range(groups)]
    tab = torch.empty((nf, 65536), dtype=torch.uint8, device="cuda")
    wv = torch.empty((nf, 256 * W * 4), dtype=torch.uint8, device="cuda")
    cnt = torch.empty((nf, 1024), dtype=torch.int32, device="cuda")
    for name, fn in (("vectorscope", lambda g: ctx.vectorscope_batch_dev(bs[g], tab, sync=False)),
                     ("histogram", lambda g: ctx.histogram_counts_batch_dev(br[g], 0x07, cnt, sync=False)),
                     ("waveform", lambda g: ctx.waveform_batch_dev(br[g], 0x07, wv, sync=False)),
                     ("three_scope", lambda g: (ctx.vectorscope_batch_dev(br[g], tab, sync=False), ctx.waveform_histogram_batch_dev(br[g], 0x07, 0x07, wv, cnt, sync=False)))):
        for g in range(groups): fn(g)
        ctx.sync()
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                for g in range(groups): fn(g)
            for _ in range(5): graph.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 40
            e0.record(stream)
            for _ in range(reps): graph.replay()
            e1.record(stream); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (reps * groups)
            print(json.dumps(dict(frames=nf, scope=name, us_per_call_graph_replay=round(us, 2))), flush=True)
        except Exception as e:
            print("graph failed", name, repr(e)[:300], flush=True)
            torch.cuda.synchronize()
