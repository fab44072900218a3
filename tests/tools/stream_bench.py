#!/usr/bin/env python3
"""BASELINE config 5 on its own: python tests/tools/stream_bench.py [--seconds 4] -- the same leg bench.py runs
(bench_stream.py: pinned host frames -> scope_pipe_* -> tables; flat out, then paced at 240 fps)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import obs_color_monitor_amd as ocm  # noqa: E402
import scope_oracle  # noqa: E402
from bench_stream import stream_leg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=4.0)
ap.add_argument("--depth", type=int, default=3)
ap.add_argument("--dist", default="natural")
args = ap.parse_args()
ctx = ocm.ScopeContext(0)
res = stream_leg(torch, np, ctx, scope_oracle.COracle(), seconds=args.seconds, depth=args.depth, dist_kind=args.dist)
print(json.dumps(res), flush=True)
ctx.close()
sys.exit(0 if res["parity"].startswith("bit-exact") else 3)
