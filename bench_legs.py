"""bench_legs.py -- the short legs of bench.py's default one-GPU run: the other workloads at the headline's batch (variants),
other input distributions, BASELINE config 1 (CPU path), config 4 (one 7680x4320 batch in N row tiles, tables summed over RCCL
through the C-ABI) and the CPU baseline (the oracle timed on the host: the only place besides Workload.check where bench.py
touches oracle/, and never the thing measured as `value`)."""
import os
import time

from bench_core import HBM_PEAK_GBPS, Workload, kernel_us, log, pmc_traffic, roofline, spread, timed_run  # noqa: F401


class Legs:
    """The short legs of the default one-GPU run that share the headline's frame geometry."""

    def __init__(self, dev, ctl, ctx, stream, o, np, torch, w, h, n):
        self.dev, self.ctl, self.ctx, self.stream, self.o, self.np, self.torch = dev, ctl, ctx, stream, o, np, torch
        self.w, self.h, self.n = w, h, n

    def variant(self, wl, rgb, yuv, comp, steps=20, seconds=0.5, check=True):
        torch, np = self.torch, self.np
        vw = Workload(torch, self.ctx, wl, rgb, yuv, self.w, self.h, self.n, comp)
        el, sus, ln, kms = timed_run(self.dev, self.ctl, self.ctx, self.stream, vw.step, steps, 3, 0.2, seconds)
        ok = vw.check(self.o, rgb, yuv, np, frames=spread(self.n, 3)) if check else None
        tr, _ = pmc_traffic(wl, self.n, self.w, self.h, comp == 0x07 and vw.linesize == 4 * self.w)
        vr = roofline(wl, vw.algo_bytes, ln, kms, sus["ms_per_step"], tr)
        if wl == "three_scope":
            # SURVEY 8(d)'s config-5 bytes (ONE read of the frames + the three tables).  The label "hbm" would flatter this
            # workload: its ceiling is VALU issue (profiles/round4/pmc_sq_three_scope_x64_*.txt).
            vr["bound"] = "valu issue (K0 + three waveform adds + one vectorscope add per texel); hbm is the yardstick, not the ceiling"
            vr["ceiling_estimate"] = {"gpix_per_s": 1300, "frac_of_hbm_peak": 0.73,
                                      "from": "the per-texel instruction count of a fast slot (K0 and its vote 11.5 + the waveform's three adds 4 + "
                                              "the vectorscope's pair 4.5 + loop and load 1-2 = 22 VALU instructions) at the 70 % VALU occupancy four "
                                              "waves per SIMD reach (measured: 26.2 instructions per texel over the whole kernel, 997-1039 Gpix/s); "
                                              "DESIGN.md 3.5"}
            vr["note"] = "algorithmic bytes = SURVEY 8(d)'s config-5 figure: one read of the frames + the three tables"
        out = {"value": round(self.n * self.w * self.h / (sus["ms_per_step"] * 1e-3) / 1e9, 3), "unit": "Gpix/s", "components": f"{comp:#04x}",
               "ms_per_step": sus["ms_per_step"], "steps": sus["steps"], "parity": ok, "roofline": vr}
        del vw
        return out

    def distributions(self, make_frames, pad_pitch, punch_alpha_holes, rgb_uniform, yuv_uniform):
        """SURVEY.md 8(d) D2-D6 (+ natural / screen stand-ins) under this clock: kernel frac of the headline kernel and of
        the fused-K0 vectorscope on each; alpha holes only change what histogram and waveform do, so those run there."""
        torch = self.torch
        out = {}

        def both(rgb, yuv, trio=True):
            a = self.variant("vectorscope", rgb, yuv, 0x07, steps=10, seconds=0.5)
            b = self.variant("vectorscope_rgb", rgb, yuv, 0x07, steps=10, seconds=0.5)
            res = {"vectorscope": {"frac": a["roofline"]["frac"], "frac_kernel": a["roofline"]["frac_kernel"], "gpix_per_s": a["value"], "parity": a["parity"]},
                   "vectorscope_rgb": {"frac": b["roofline"]["frac"], "frac_kernel": b["roofline"]["frac_kernel"], "gpix_per_s": b["value"], "parity": b["parity"]}}
            if trio:  # the one-read three-scope kernel is LDS / VALU bound: same-address LDS atomics make it content dependent
                c = self.variant("three_scope", rgb, yuv, 0x07, steps=6, seconds=0.4)
                res["three_scope"] = {"gpix_per_s": c["value"], "frac": c["roofline"]["frac"], "parity": c["parity"]}
            return res

        for kind in ("natural", "concentrated", "flat", "screen", "ramp"):
            rgb, yuv = make_frames(torch, self.ctx, self.n, self.w, self.h, kind, 0xD157 + len(out), device=self.dev.device)
            out[kind] = both(rgb, yuv)
            del rgb, yuv
            self.dev.empty_cache()
        prgb, pyuv = pad_pitch(torch, rgb_uniform[:self.n], self.w), pad_pitch(torch, yuv_uniform[:self.n], self.w)
        out["padded_pitch"] = dict(both(prgb, pyuv), linesize=int(prgb.shape[2]))
        del prgb, pyuv
        holes = punch_alpha_holes(torch, rgb_uniform[:self.n].clone(), self.w)
        hy = yuv_uniform[:self.n]
        a = self.variant("histogram", holes, hy, 0x07, steps=10, seconds=0.5)
        b = self.variant("waveform", holes, hy, 0x07, steps=10, seconds=0.5)
        out["alpha_holes"] = {"histogram": {"frac": a["roofline"]["frac"], "frac_kernel": a["roofline"]["frac_kernel"], "gpix_per_s": a["value"], "parity": a["parity"]},
                              "waveform": {"frac": b["roofline"]["frac"], "frac_kernel": b["roofline"]["frac_kernel"], "gpix_per_s": b["value"], "parity": b["parity"]},
                              "note": "the vectorscope ignores alpha (src/vectorscope.c:217-238): histogram and waveform are what D5 exercises"}
        del holes
        self.dev.empty_cache()
        out["what"] = (f"{self.n} frames per launch; frac = algorithmic bytes / the whole step / 8 TB/s, frac_kernel = / the accumulation "
                       "kernel's own begin/end stamps; pools taken as allocated; uniform noise is the headline")
        return out


def config1_leg(np, ctx, o, reps=25):
    """BASELINE configs[0]: 256-bin RGB histogram over one 1280x720 frame on the CPU path (the oracle = this build's port of
    src/histogram.c:330-418, one thread, counts + level pass), and the GPU's result for the same frame."""
    w, h = 1280, 720
    rgb = o.fill_frame(w, h, 1, 0x720)
    times = []
    for _ in range(5 + reps):
        t0 = time.perf_counter()
        want, want_hi = o.histogram(rgb, None, w, 0x07)
        times.append(time.perf_counter() - t0)
    times = sorted(times[5:])
    ms = times[len(times) // 2] * 1e3
    got, hi, st = ctx.histogram(rgb=rgb, yuv=None, width=w, components=0x07)
    same = st == 0 and bool(np.array_equal(got.view(np.uint32), want.view(np.uint32))) and bool(np.array_equal(hi, want_hi))
    return {"workload": "histogram RGB (0x07), one 1280x720 BGRA8 frame, CPU path only (oracle port, one thread, -O2)",
            "ms_per_frame": round(ms, 4), "gpix_per_s": round(w * h / (ms * 1e-3) / 1e9, 3), "median_of": reps,
            "gpu_tables_bit_exact": same, "kind": "port (parity unpinned)",
            "note": "excludes the reference's GPU->CPU readback; the GPU call here is a host-buffer call (PCIe-inclusive), not timed"}


def config4_leg(dev, ctl, torch, np, ctx, o, comm, rank, world, w, h, frames=16, seconds=0.6, overlap=None):
    """ONE batch of w x h (7680x4320) frames, the same frames whatever N is (shared seed); rank r owns rows
    [h r/N, h (r+1)/N) of every frame; per step every rank reduces its tiles, the packed tables are summed over RCCL
    through the C-ABI and clamped.  Figures: throughput with the steps enqueued back to back (step i's sum under step
    i+1's accumulation when `overlap`), the same with the sum on the accumulation's stream, and the latency of one step
    with its reduce waited for (nothing bucketed or deferred)."""
    from obs_color_monitor_amd.row_tiles import tile_rows
    from obs_color_monitor_amd.synth import make_frame_tiles

    r0, r1 = tile_rows(h, rank, world)
    th = r1 - r0
    rgb, yuv = make_frame_tiles(torch, ctx, frames, w, h, r0, r1, 0xC0FFEE, device=dev.device)
    del rgb
    batch = ctx.make_device_batch(yuv, 4 * w, w, th, planes_are_rgb=False)
    wires = [torch.empty((frames, 32768), dtype=torch.int32, device=dev.device) for _ in range(2)]
    outs = [torch.empty((frames, 65536), dtype=torch.uint8, device=dev.device) for _ in range(2)]
    count = [0]

    def step(ov):
        buf = count[0] & 1
        count[0] += 1
        rctx = ctx
        if ov:
            ov.before_compute(buf)
            rctx = ov.ctx
        ctx.vectorscope_batch_dev(batch, wires[buf], sync=False, packed16=True)
        if ov:
            ov.after_compute(buf)
        rctx.table_allreduce(comm, wires[buf], frames * 32768)
        rctx.unpack16_clamp_dev(wires[buf], outs[buf], frames * 32768)
        if ov:
            ov.after_reduce(buf)
        return buf

    def throughput(ov):
        """every rank runs the SAME number of steps (a step holds a collective): fixed chunks, the ranks agree after each
        whether to go on"""
        for _ in range(6):
            step(ov)
        dev.sync()
        ctl.barrier()
        t0 = time.perf_counter()
        steps, chunk = 0, 40 if not dev.emul else 2
        while True:
            for _ in range(chunk):
                step(ov)
            steps += chunk
            dev.sync()
            if not ctl.any_true(time.perf_counter() - t0 < seconds):
                break
        dev.sync()
        ctl.barrier()
        return ctl.max(time.perf_counter() - t0) / steps * 1e3, steps

    ms_serial, steps_serial = throughput(None)
    ms_overlap, steps_overlap = throughput(overlap) if overlap else (None, 0)
    dev.sync()
    lat = []
    for _ in range(30 if not dev.emul else 2):
        ctl.barrier()
        dev.sync()
        t1 = time.perf_counter()
        last = step(None)
        dev.sync()
        lat.append(time.perf_counter() - t1)
    lat_ms = ctl.max(sorted(lat)[len(lat) // 2]) * 1e3
    # parity of frames 0 and last of the batch: the sum of every rank's oracle table of its tile, clamped once; checked on
    # the table the serial path left AND on one the overlapped path left (the two must be the same table)
    ok = True
    for f in sorted({0, frames - 1}):
        want = np.minimum(ctl.sum_array(o.vectorscope(yuv[f].cpu().numpy(), w).astype(np.int64).reshape(-1)), 255)
        ok &= bool(np.array_equal(outs[last][f].cpu().numpy().astype(np.int64), want))
        if overlap:
            b2 = step(overlap)
            dev.sync()
            ok &= bool(np.array_equal(outs[b2][f].cpu().numpy().astype(np.int64), want))
    ok = ctl.all_true(ok)
    best = min(ms_serial, ms_overlap) if ms_overlap else ms_serial
    return {
        "value": round(frames * w * h / (best * 1e-3) / 1e9, 3), "unit": "Gpix/s", "scaling": "strong", "ms_per_step": round(best, 4),
        "ms_per_step_reduce_on_the_accumulation_stream": round(ms_serial, 4),
        "ms_per_step_reduce_overlapped_second_stream": round(ms_overlap, 4) if ms_overlap else None,
        "step_latency_ms_reduce_waited": round(lat_ms, 4),
        "steps_timed": [steps_serial, steps_overlap],
        "partition": f"{frames} frames of {w}x{h} (shared seed: the same frames at every N), {world} row tile(s) of {w}x{th} per frame "
                     f"+ RCCL sum of the 256x256 tables through the C-ABI, every step",
        "frames_per_step": frames, "rows_of_this_rank": [r0, r1], "parity": bool(ok),
    }


def cpu_baseline(np, o, wl, plane_rgb, plane_yuv, w, h, components, seconds):
    def one(kind=wl, comp=components):
        if kind in ("vectorscope", "vectorscope_rgb"):
            return o.vectorscope(plane_yuv, w)
        if kind == "waveform":
            return o.waveform(plane_rgb, plane_yuv, w, comp)
        return o.histogram_counts(plane_rgb, plane_yuv, w, comp)

    def rate(kind, comp, secs):
        t_end = time.perf_counter() + secs
        reps, t_start = 0, time.perf_counter()
        while time.perf_counter() < t_end or reps < 3:
            one(kind, comp)
            reps += 1
        dt = time.perf_counter() - t_start
        return reps, dt, reps * w * h / dt / 1e9

    reps, cpu_t, gpix = rate(wl, components, seconds)
    res = {
        "value": round(gpix, 4),
        "unit": "Gpix/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{reps} passes of oracle {wl} (scope_oracle.c, gcc -O2, one thread as the reference's "
                  f"single color-monitor thread) over frame 0 of the same batch, {cpu_t:.1f} s; "
                  f"host has {os.cpu_count()} cores; excludes the reference's GPU->CPU readback",
    }
    # the other two loops of the path on the same frame, one thread, short samples
    others = {}
    for kind, comp in (("vectorscope", 0x07), ("waveform", 0x07), ("waveform", 0x20), ("histogram", 0x07)):
        if kind == wl and comp == components:
            continue
        r, dt, g = rate(kind, comp, min(2.0, seconds / 4))
        others[f"{kind}_{comp:#04x}"] = {"value": round(g, 4), "unit": "Gpix/s", "cores": 1, "passes": r, "seconds": round(dt, 2)}
    res["other_loops_one_thread"] = others
    # courtesy: every host core, row/column tiles with private tables; same tables, checked here
    threads = os.cpu_count() or 1
    cpu_model = "?"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass

    def all_cores(nthreads, reps=1):
        if wl in ("vectorscope", "vectorscope_rgb"):
            return o.vectorscope_mt(plane_yuv, w, nthreads, reps)
        if wl == "waveform":
            return o.waveform_mt(plane_rgb, plane_yuv, w, components, nthreads, reps)
        return o.histogram_counts_mt(plane_rgb, plane_yuv, w, components, nthreads, reps)

    mt_same = bool(np.array_equal(all_cores(threads).astype(np.int64).reshape(-1), one().astype(np.int64).reshape(-1)))
    mt_reps = 20  # per call; thread start-up is paid once per call
    best = None
    for nthreads in sorted({t for t in (16, 32, 64, 128, 256, threads) if t <= threads}):
        all_cores(nthreads, 2)
        t_start, calls = time.perf_counter(), 0
        while time.perf_counter() - t_start < min(seconds, 10.0) / 8 or calls < 1:
            all_cores(nthreads, mt_reps)
            calls += 1
        r = calls * mt_reps * w * h / (time.perf_counter() - t_start) / 1e9
        if best is None or r > best[0]:
            best = (r, nthreads, calls)
    res["all_cores"] = {
        "value": round(best[0], 3), "unit": "Gpix/s", "cores": best[1], "host_cores": threads, "cpu": cpu_model,
        "same_table": mt_same,
        "sample": f"best of 16..{threads} POSIX threads: {best[2]} calls x {mt_reps} passes over frame 0 (row tiles with "
                  "private tables; column tiles for the waveform; thread start-up once per call)",
    }
    return res
