#!/usr/bin/env python3
"""Regenerates tests/golden/*.  Inputs come from oracle_fill_frame (seeded xorshift64), expected
outputs from the C oracle (oracle/scope_oracle.c).  The reference ships no golden vectors for this
path and cannot be built in this image (no libobs headers, no stand-ins), so these fixtures pin the
ORACLE against regressions; they are not reference-produced ("parity unpinned", see scope_oracle.h).

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import scope_oracle  # noqa: E402

o = scope_oracle.COracle()
COMPS = (0x07, 0x20, 0x50, 0x70, 0x04)

small = {}
for dist in (1, 2, 3, 4, 5):
    for (w, h, ls) in ((64, 36, 256), (50, 20, 4 * 50 + 24)):
        key = f"d{dist}_{w}x{h}_ls{ls}"
        rgb = o.fill_frame(w, h, dist, 0, ls)
        small[key + "_rgb"] = rgb
        for cs in (1, 2):
            yuv = o.rgb_to_yuv(rgb, w, cs)
            small[key + f"_yuv{cs}"] = yuv
        vs = o.vectorscope(yuv, w)  # cs 2
        nz = np.flatnonzero(vs)
        small[key + "_vss_idx"] = nz.astype(np.uint16)
        small[key + "_vss_val"] = vs.reshape(-1)[nz]
        for c in COMPS:
            wv = o.waveform(rgb, yuv, w, c)
            nz = np.flatnonzero(wv)
            small[key + f"_wvs{c:02x}_idx"] = nz.astype(np.uint32)
            small[key + f"_wvs{c:02x}_val"] = wv.reshape(-1)[nz]
            small[key + f"_his{c:02x}"] = o.histogram_counts(rgb, yuv, w, c)
np.savez_compressed(os.path.join(HERE, "small_frames.npz"), **small)

# full-size frames are regenerated from their seed on the box; only hashes are committed
hashes = {}
for (w, h, dist) in ((1280, 720, 1), (3840, 2160, 1), (3840, 2160, 2), (3840, 2160, 5), (7680, 4320, 2)):
    rgb = o.fill_frame(w, h, dist, 0)
    yuv = o.rgb_to_yuv(rgb, w, 2)
    key = f"d{dist}_{w}x{h}"
    hashes[key] = {
        "rgb": f"{o.fnv(rgb):016x}",
        "yuv709": f"{o.fnv(yuv):016x}",
        "vectorscope": f"{o.fnv(o.vectorscope(yuv, w)):016x}",
        "histogram_07": f"{o.fnv(o.histogram_counts(rgb, yuv, w, 0x07)):016x}",
    }
    if w <= 3840:
        hashes[key]["waveform_07"] = f"{o.fnv(o.waveform(rgb, yuv, w, 0x07)):016x}"
        hashes[key]["waveform_20"] = f"{o.fnv(o.waveform(rgb, yuv, w, 0x20)):016x}"
json.dump(hashes, open(os.path.join(HERE, "fullsize_hashes.json"), "w"), indent=1, sort_keys=True)
# capture pre-steps (target_scale / ROI) and graticule geometry: hashes of the oracle's outputs on the small fixture frame
import graticule_oracle as go  # noqa: E402

extra = {"stage_region": {}, "graticule": {}}
frame = o.fill_frame(64, 36, 1, 0, 256)
for region in ((1, 0, 0, 0, 0), (2, 0, 0, 0, 0), (3, 0, 0, 0, 0), (2, 3, 1, 20, 11), (1, 7, 5, 40, 17), (5, 0, 0, 0, 0)):
    extra["stage_region"]["s%d_x%d_y%d_%dx%d" % region] = f"{o.fnv(o.stage_region(frame, 64, *region)):016x}"
for cs in (1, 2):
    for flags in (1, 257):
        labels, lines = go.vectorscope(cs, flags, 0x0054FF)
        extra["graticule"][f"vss_cs{cs}_f{flags}"] = f"{o.fnv(labels):016x}:{o.fnv(lines):016x}"
for display, n in ((0, 3), (1, 3), (2, 3), (1, 2)):
    extra["graticule"][f"wvs_d{display}_n{n}"] = f"{o.fnv(go.waveform(5, display, n, 1920)):016x}"
    extra["graticule"][f"his_d{display}_n{n}"] = f"{o.fnv(go.histogram(5, 50.0, 0, 200, 0, display, n, 200)):016x}"
json.dump(extra, open(os.path.join(HERE, "capture_graticule_hashes.json"), "w"), indent=1, sort_keys=True)

print("wrote", os.path.getsize(os.path.join(HERE, "small_frames.npz")), "bytes of small fixtures and", len(hashes), "hash sets")
