#!/usr/bin/env python3
"""Randomised parity sweep on the GPU: random geometries (widths with and without the vector shape, heights around the
row-group and register limits of the one-frame kernels, padded pitches), component masks, matrices and launch sizes through
every accumulation entry point, each table compared with the oracle.  Test infrastructure (the oracle is the checker).

    python tools/fuzz_parity.py [--cases 200] [--seed 1]
"""
import argparse
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

import obs_color_monitor_amd as ocm  # noqa: E402
import parity_cases as pc  # noqa: E402
import scope_oracle  # noqa: E402


def run(ctx, o, cases, seed, log=print, to_dev=None, to_host=None):
    """to_dev / to_host: how "device" buffers are made and read back (default: torch on the GPU; the CPU emulation passes
    numpy buffers, which are their own device memory there)."""
    rnd = random.Random(seed)
    if to_dev is None:
        import torch

        def to_dev(a):
            t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
            return t, int(t.data_ptr())

        to_host = lambda t: t.cpu().numpy()  # noqa: E731
    masks = [0x07, 0x01, 0x02, 0x04, 0x03, 0x05, 0x06]
    done = 0
    for case in range(cases):
        shape = rnd.choice(["tiny", "wide", "tall", "mid", "odd"])
        if shape == "tiny":
            w, h = rnd.choice([4, 8, 12, 16, 20, 36, 60, 64, 68]), rnd.randint(1, 40)
        elif shape == "wide":
            w, h = 4 * rnd.randint(200, 1024), rnd.randint(1, 24)
        elif shape == "tall":
            w, h = 4 * rnd.randint(1, 24), rnd.choice([2160, 2175, 2176, 2177, 2300, 4081]) if rnd.random() < 0.5 else rnd.randint(500, 2400)
        elif shape == "mid":
            w, h = 4 * rnd.randint(16, 160), rnd.randint(30, 400)
        else:
            w, h = rnd.randint(1, 300), rnd.randint(1, 200)  # any width: the scalar shapes
        pad = rnd.choice([0, 0, 8, 16, 64, 4])
        ls = 4 * w + pad
        dist = rnd.choice([1, 2, 3, 4, 5])
        cs = rnd.choice([1, 2])
        wc, hc = rnd.choice(masks), rnd.choice(masks)
        if rnd.random() < 0.6:
            hc = wc & hc or wc  # mostly a subset: the one-pass forms
        n = rnd.choice([1, 1, 1, 2, 3])
        if w * h <= 64 * 1024 and rnd.random() < 0.3:
            n = rnd.choice([5, 8, 17, 66])  # batches of small frames: several strips per slice, more frames than a launch takes
        tag = f"case {case}: {w}x{h} pitch {ls} dist {dist} cs {cs} wc {wc:#x} hc {hc:#x} n {n}"
        try:
            with pc.options(ctx, **{"trio.strip": rnd.choice([0, 0, 16, 32])}):  # round 4: both strip widths on any launch size
                pc.check_three_scopes_dev(ctx, o, to_dev, to_host, w, h, [dist] * n if n > 1 else [dist], wc, hc, ls, cs, reps=1)
            with pc.options(ctx, dock=rnd.choice([0, 2])):  # the two-pass form / round 2's one-frame kernel
                pc.check_three_scopes_dev(ctx, o, to_dev, to_host, w, h, [dist], wc, hc, ls, cs, reps=1)
            comp = rnd.choice(pc.COMPONENTS)
            pc.check_waveform(ctx, o, w, h, dist, comp, ls, fuse=bool(comp & 0x70) and rnd.random() < 0.5, colorspace=cs)
            pc.check_histogram_counts(ctx, o, w, h, dist, comp, ls, colorspace=cs)
            fuse = rnd.random() < 0.5
            opts = {}
            if fuse and rnd.random() < 0.7:  # the batch form of the fused kernel (pair-addressed table) on any geometry
                opts = {"vss.fused_hash": rnd.choice([0, 2, 2]), "vss.fused_unroll": rnd.choice([1, 2, 4, 4]),
                        "slices": rnd.choice([0, 1, 3]), "vss.merge": rnd.choice([0, 2, 3, 3]), "vss.groups": rnd.choice([-1, 0, 2, 3]),
                        "vss.rle": rnd.choice([0, 1, 1])}  # round 4: the run-length mode voted / forced (lives in the unroll-4 form)
            with pc.options(ctx, **opts):
                pc.check_vectorscope(ctx, o, w, h, dist, ls, fuse=fuse, colorspace=cs)
                if fuse and w >= 4:  # screen-like content: flat areas, rectangles, sprinkled detail, greys, random alpha
                    rgb = pc.screen_frame(w, h, case, detail=rnd.choice([0.0, 0.01, 0.03, 0.3]), rects=rnd.choice([0, 3, 12, 60]), linesize=ls,
                                          grey=rnd.random() < 0.2, alpha_noise=rnd.random() < 0.3)
                    pc.check_vectorscope_rgb_frame(ctx, o, rgb, w, cs, "fuzz screen")
            if not fuse and w >= 4:  # the staged-plane kernel's run-length form on keys (its unroll-4 form), voted / forced
                with pc.options(ctx, **{"vss.unroll": rnd.choice([0, 4, 4]), "vss.rle": rnd.choice([0, 1]), "slices": rnd.choice([0, 1, 3]),
                                        "vss.groups": rnd.choice([-1, 0, 2, 3])}):
                    rgb = pc.screen_frame(w, h, case, detail=rnd.choice([0.0, 0.01, 0.03, 0.3]), rects=rnd.choice([0, 3, 12, 60]), linesize=ls,
                                          grey=rnd.random() < 0.2, alpha_noise=rnd.random() < 0.3)
                    yuv = o.rgb_to_yuv(rgb, w, cs)
                    got, st = ctx.vectorscope(yuv=yuv, width=w)
                    assert st == 0
                    pc.assert_same(got, o.vectorscope(yuv, w), "fuzz screen, staged planes")
                    pc.check_vectorscope(ctx, o, w, h, dist, ls, fuse=False, colorspace=cs)
        except Exception as e:  # noqa: BLE001
            log("FAIL " + tag + " -- " + repr(e)[:300])
            raise
        done += 1
        if done % 20 == 0:
            log(f"{done} cases ok (last: {tag})")
    return done


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import torch  # noqa: F401  (first, so that the library shares torch's HIP runtime)

    ctx = ocm.ScopeContext(0)
    done = run(ctx, scope_oracle.COracle(), args.cases, args.seed, lambda m: print(m, flush=True))
    print(f"all {done} cases bit-exact vs the oracle")


if __name__ == "__main__":
    main()
