#!/usr/bin/env python3
"""Turns the FETCH_SIZE / WRITE_SIZE counter passes of `tools/gpu_job.sh <tag> traffic` into the JSON bench.py reads
(profiles/round1/pmc_traffic.json): HBM bytes per launch of the accumulation kernel of each workload.

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: both counters are in KiB...
FETCH_SIZE counts a wide coalesced stream at half its bytes (x2); WRITE_SIZE is taken as is (checked here against the
K0 kernel, whose output size is known exactly)."""
import collections
import csv
import glob
import json
import os
import sys

KERNEL = {"vectorscope": "scope_vss_kernel", "vectorscope_rgb": "scope_vss_kernel", "waveform": "scope_wvs_kernel",
          "histogram": "scope_his_kernel"}
W, H = 3840, 2160
FRAMES = int(sys.argv[2]) if len(sys.argv) > 2 else 64  # frames per launch of the profiled bench.py runs (its default)


def mean_counter(d, kernel, counter):
    vals, name = [], None
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if kernel in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                vals.append(float(row["Counter_Value"]))
                name = row["Kernel_Name"].split("(")[0]
    return (sum(vals) / len(vals) if vals else None), name, len(vals)


def main():
    out = sys.argv[1]
    entries = []
    for wl, kern in KERNEL.items():
        fd, wd = os.path.join(out, f"pmc_{wl}_FETCH_SIZE"), os.path.join(out, f"pmc_{wl}_WRITE_SIZE")
        if not (os.path.isdir(fd) and os.path.isdir(wd)):
            continue
        fetch, name, nf = mean_counter(fd, kern, "FETCH_SIZE")
        write, _, nw = mean_counter(wd, kern, "WRITE_SIZE")
        if fetch is None or write is None:
            continue
        k0_write, _, _ = mean_counter(wd, "scope_k0_kernel", "WRITE_SIZE")
        entries.append({
            "workload": wl, "kernel": name, "frames_per_step": FRAMES, "width": W, "height": H,
            "FETCH_SIZE_KB_per_launch": round(fetch, 1), "WRITE_SIZE_KB_per_launch": round(write, 1),
            "launches_averaged": [nf, nw], "fetch_correction": 2.0,
            "k0_WRITE_SIZE_KB_per_launch": k0_write, "k0_output_KB": W * H * 4 / 1024,
            "bytes_per_launch": int(round((2.0 * fetch + write) * 1024)),
        })
    json.dump({"source": f"{out}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace, "
                         "bench.py --steps 10 per workload (tools/gpu_job.sh traffic)",
               "note": "gfx950 FETCH_SIZE counts a wide coalesced stream at half its bytes (MI355X_MICROARCH.md, HBM "
                       "section): doubled; WRITE_SIZE taken as is (the K0 kernel's WRITE_SIZE equals its output bytes)",
               "entries": entries}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
