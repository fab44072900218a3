#!/usr/bin/env python3
"""Generates LDS address traces (see lds_trace_bench.hip) for candidate bin->word hashes and lane
mappings from oracle-generated frames, and prints the analytic conflict model next to each."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import scope_oracle as so  # noqa: E402

o = so.COracle()
W, H = 3840, 540
rng = np.random.default_rng(1)
N = 32


def make(dist):
    rgb = np.zeros((H, W, 4), np.uint8)
    rgb[..., 3] = 255
    if dist == "uniform":
        rgb[..., :3] = rng.integers(0, 256, (H, W, 3))
    elif dist == "concentrated":
        rgb[..., :3] = rng.integers(120, 136, (H, W, 3))
    elif dist == "gradient":
        xs = np.arange(W)[None, :]
        ys = np.arange(H)[:, None]
        n = rng.integers(0, 4, (H, W, 3))
        rgb[..., 0] = np.clip(xs * 255 // W + n[..., 0], 0, 255)
        rgb[..., 1] = np.clip(ys * 255 // (H * 4) + n[..., 1], 0, 255)
        rgb[..., 2] = np.clip((xs + ys) * 255 // (W + H * 4) + n[..., 2], 0, 255)
    return o.rgb_to_yuv(rgb.reshape(H, W * 4), W, 2).reshape(H, W, 4)


def rev5(r):
    r = r & 31
    return ((r & 1) << 4) | ((r & 2) << 2) | (r & 4) | ((r & 8) >> 2) | ((r & 16) >> 4)


HASHES = {
    "h0:((u>>1)+8row)&127": lambda u, r: (r << 7) + (((u >> 1) + (r << 3)) & 127),
    "hx:(u>>1)^((row&3)<<3)": lambda u, r: (r << 7) + ((u >> 1) ^ ((r & 3) << 3)),
    "h1:(u&127)^rev5(row)": lambda u, r: (r << 7) + ((u & 127) ^ rev5(r)),
    "h2:(u&127)+9*row": lambda u, r: (r << 7) + (((u & 127) + 9 * r) & 127),
    "h3:plain u>>1": lambda u, r: (r << 7) + (u >> 1),
}


def model(words):
    tot = 0.0
    for g in words.reshape(-1, 32):
        a, c = np.unique(g, return_counts=True)
        w = np.where(c == 1, 1, 2 * c)
        per = np.bincount(a & 31, weights=w, minlength=32)
        tot += max(per.max(), 2.25)
    return 2 * tot / (len(words.reshape(-1, 32)) / 1)  # per wave-instr = two groups


def trace_seq(yuv, hashf, glog2, order="kernel"):
    """The kernel's real issue order: for each of 8 consecutive batches k, the four pixel slots j of
    the lane's 16-byte chunk back to back ("kernel"), or the same 32 instructions shuffled."""
    u = yuv[..., 0].astype(np.int64).reshape(-1)
    v = yuv[..., 2].astype(np.int64).reshape(-1)
    nch = u.size // 4
    G, lg, tg = 1 << glog2, 64 >> glog2, 1024 >> glog2
    out = np.zeros((N, 64), np.uint32)
    wave, it0 = 5, 40
    lanes = np.arange(64)
    for s in range(N):
        k, j = s // 4, s % 4
        c = (nch * (lanes // lg)) // G + wave * lg + lanes % lg + (it0 + k) * tg
        px = c * 4 + j
        out[s] = hashf(u[px], 255 - v[px])
    if order == "shuffled":
        out = out[rng.permutation(N)]
    elif order == "slot-major":  # all batches' slot 0, then slot 1, ...
        out = out.reshape(8, 4, 64).transpose(1, 0, 2).reshape(N, 64)
    return out


def trace(yuv, hashf, glog2):
    u = yuv[..., 0].astype(np.int64).reshape(-1)
    v = yuv[..., 2].astype(np.int64).reshape(-1)
    nch = u.size // 4
    G, lg, tg = 1 << glog2, 64 >> glog2, 1024 >> glog2
    out = np.zeros((N, 64), np.uint32)
    for s in range(N):
        wave, it, j = rng.integers(0, 16), rng.integers(0, (nch // G) // tg - 1), s % 4
        lanes = np.arange(64)
        c = (nch * (lanes // lg)) // G + wave * lg + lanes % lg + it * tg
        px = c * 4 + j
        out[s] = hashf(u[px], 255 - v[px])
    return out


with open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/lds_traces.bin", "wb") as f:
    def emit(label, words):
        f.write(label.encode()[:63].ljust(64, b"\0"))
        f.write(np.ascontiguousarray(words, np.uint32).tobytes())
        print(f"{label:56s} model {model(words) / 2:6.2f} clk/wave-instr", file=sys.stderr)

    lanes = np.arange(64)
    emit("synthetic conflict-free", np.tile(lanes % 32 + 32 * (lanes // 32), (N, 1)) + 64 * np.arange(N)[:, None])
    emit("synthetic true-random words", rng.integers(0, 32768, (N, 64)))
    emit("synthetic random, 2 lanes per address", np.repeat(rng.integers(0, 32768, (N, 32)), 2, axis=1))
    emit("synthetic random, 4 lanes per address", np.repeat(rng.integers(0, 32768, (N, 16)), 4, axis=1))
    emit("synthetic 8 addresses x 8 lanes, 8 banks", np.tile(np.repeat(np.arange(8), 8), (N, 1)) + 32 * np.arange(N)[:, None])
    emit("synthetic 8 addresses x 8 lanes interleaved", np.tile(np.tile(np.arange(8), 8), (N, 1)) + 32 * np.arange(N)[:, None])
    for dist in ("uniform", "concentrated", "gradient"):
        yuv = make(dist)
        for hn, hf in HASHES.items():
            for gl in (0, 3):
                emit(f"{dist} {hn} groups={1 << gl}", trace(yuv, hf, gl))
        for hn in ("h0:(u>>1)^((row&3)<<3)", "h1:(u&127)^rev5(row)"):
            for gl in (0, 3):
                for order in ("kernel", "shuffled", "slot-major"):
                    emit(f"{dist} {hn[:2]} g={1 << gl} order={order}", trace_seq(yuv, HASHES[hn], gl, order))
