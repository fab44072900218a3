"""BASELINE config 5 as a measurement leg: a stream of 3840x2160 frames, histogram + waveform + vectorscope per frame on
one MI355X with upload / compute / download hipStreams overlapped (scope_pipe_* of the C-ABI, ring depth 3 as the
reference's CM_SURFACE_QUEUE_SIZE, src/common.h:46).

Frames start in PINNED HOST memory (what a capture path would fill); every frame crosses PCIe, so these are the
PCIe-inclusive rates of the whole drop-in path -- never bench.py's `value`.  A measurement leg, so it lives beside
bench.py and not in the package (it takes the oracle as the checker of sampled frames): used by bench.py (`stream`
object of the default run, `--workload stream`) and tests/tools/stream_bench.py.
"""
from __future__ import annotations

import time

W, H = 3840, 2160


def _run(pipe, ptrs, n, np, check_every=0, pace=0.0, seconds=0.0):
    """Streams n frames (or for `seconds`), at most pipe depth in flight.  Returns (elapsed, latencies, sampled tables)."""
    lat, submit_t, done, results = [], {}, 0, {}
    sources = len(ptrs)
    t0 = time.perf_counter()
    i = 0

    def collect_one():
        nonlocal done
        got = pipe.collect()
        lat.append(time.perf_counter() - submit_t.pop(done))
        if check_every and done % check_every == 0:
            results[done] = tuple(np.array(t) for t in got)
        done += 1

    while (i < n) if not seconds else (time.perf_counter() - t0 < seconds):
        if pace:
            while time.perf_counter() - t0 < i / pace:
                pass
        while True:
            submit_t[i] = time.perf_counter()
            if pipe.submit(ptrs[i % sources], 4 * W):
                if pace:  # paced like a live source: take each frame's tables as soon as they exist
                    collect_one()
                break
            collect_one()
        i += 1
    while done < i:
        collect_one()
    return time.perf_counter() - t0, lat, results, i


def stream_leg(torch, np, ctx, oracle, seconds=2.0, sources=16, depth=3, dist_kind="natural"):
    """Flat out for `seconds`, then paced at 240 frames/s for `seconds`; sampled frames checked against the oracle."""
    from obs_color_monitor_amd import ScopePipe
    from obs_color_monitor_amd.synth import make_rgb_frames

    rgb = make_rgb_frames(torch, sources, W, H, dist_kind, 7)
    torch.cuda.synchronize()
    pinned = torch.empty((sources, H, W * 4), dtype=torch.uint8, pin_memory=True)
    pinned.copy_(rgb)
    torch.cuda.synchronize()
    del rgb
    ptrs = [int(pinned[i].data_ptr()) for i in range(sources)]
    pipe = ScopePipe(ctx, W, H, depth=depth, colorspace=2, wvs_components=0x07, his_components=0x07)
    try:
        _run(pipe, ptrs, 3 * depth + 8, np)  # warm: allocations, clocks
        out = {}
        parity = True
        checked = 0
        for name, pace in (("flat_out", 0.0), ("paced_240fps", 240.0)):
            elapsed, lat, results, frames = _run(pipe, ptrs, 0, np, check_every=97, pace=pace, seconds=seconds)
            for idx, (v, wv, hs) in list(results.items())[:3]:
                src = pinned[idx % sources].numpy()
                yuv = oracle.rgb_to_yuv(src, W, 2)
                parity &= bool(np.array_equal(v, oracle.vectorscope(yuv, W)))
                parity &= bool(np.array_equal(wv, oracle.waveform(src, yuv, W, 0x07)))
                parity &= bool(np.array_equal(hs, oracle.histogram_counts(src, yuv, W, 0x07)))
                checked += 1
            fps = frames / elapsed
            out[name] = {
                "fps": round(fps, 1), "frames": frames, "seconds": round(elapsed, 3),
                "gpix_per_s": round(fps * W * H / 1e9, 2), "h2d_gbytes_per_s": round(fps * W * H * 4 / 1e9, 2),
                "latency_ms_mean": round(1e3 * sum(lat) / len(lat), 3),
                "latency_ms_p99": round(1e3 * sorted(lat)[min(len(lat) - 1, (99 * len(lat)) // 100)], 3),
                "latency_ms_max": round(1e3 * max(lat), 3),
            }
        out["paced_240fps"]["meets_240fps"] = out["paced_240fps"]["fps"] >= 239.5
        out["parity"] = (f"bit-exact vs oracle on {checked} sampled frames (parity unpinned)" if parity else "MISMATCH")
        out["what"] = (f"{sources} distinct pinned host frames ({dist_kind}), RGB plane only over PCIe, K0 fused on the GPU; "
                       f"vectorscope + waveform 0x07 + histogram 0x07 per frame (the histogram rides on the waveform's pass); "
                       f"{depth} frames in flight; latency = submit -> tables on the host")
        return out
    finally:
        pipe.close()
