#!/usr/bin/env python3
"""Turns the FETCH_SIZE / WRITE_SIZE counter passes of `tools/gpu_job.sh <tag> traffic` into the JSON bench.py reads
(profiles/round2/pmc_traffic.json): HBM bytes per STEP of each workload -- every scope kernel of a step counted, i.e.
the accumulation kernel(s) AND the partial-table reduction where a step has one -- with the fingerprint of the kernel
sources they were measured on (bench.py refuses a file measured on other sources).

    python tools/pmc_traffic_json.py <job dir> <job name>
Directories: <job dir>/pmc_<workload>_x<frames>_{FETCH_SIZE,WRITE_SIZE}.

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: both counters are in KiB;
FETCH_SIZE counts a wide coalesced stream at half its bytes (x2); WRITE_SIZE is taken as is (checked here against the
K0 kernel, whose output size is known exactly)."""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEAD = {"vectorscope": "scope_vss_kernel", "vectorscope_rgb": "scope_vss_kernel", "waveform": "scope_wvs_kernel",
        "histogram": "scope_his_kernel", "three_scope": "scope_vss_kernel"}
W, H = 3840, 2160


def code_only(src):
    """As bench.py: the source text without comments and white space."""
    src = re.sub(rb"/\*.*?\*/", b"", src, flags=re.S)
    src = re.sub(rb"//[^\n]*", b"", src)
    return re.sub(rb"\s+", b"", src)


def fingerprint():
    hsh = hashlib.sha256()
    for f in ("scope_kernels.hip", "scope_trio.hip", "scope_device.h", "scope_kernels.h", "scope_vec.h", "scope_api.hip"):
        with open(os.path.join(ROOT, "obs-color-monitor_amd", "csrc", f), "rb") as fh:
            hsh.update(code_only(fh.read()))
    return hsh.hexdigest()[:16]


def counters(d, counter):
    """kernel base name -> (sum of counter over its dispatches, dispatches)"""
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = re.sub(r"^void ", "", row.get("Kernel_Name", "?")).split("<")[0].split("(")[0]
            if not name.startswith("scope_"):
                continue
            acc[name][0] += float(row["Counter_Value"])
            acc[name][1] += 1
    return acc


def main():
    out, job = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "?")
    entries = []
    for fd in sorted(glob.glob(os.path.join(out, "pmc_*_FETCH_SIZE"))):
        m = re.match(r"pmc_(.+)_x(\d+)_FETCH_SIZE$", os.path.basename(fd))
        if not m:
            continue
        wl, frames = m.group(1), int(m.group(2))
        wd = fd.replace("FETCH_SIZE", "WRITE_SIZE")
        if not os.path.isdir(wd) or wl not in LEAD:
            continue
        fetch, write = counters(fd, "FETCH_SIZE"), counters(wd, "WRITE_SIZE")
        lead = LEAD[wl]
        if wl == "three_scope":  # one kernel reads the planes once (round 3: any launch; round 2: one-frame launches)
            for k in ("scope_trio_kernel", "scope_dock_kernel"):
                if k in fetch:
                    lead = k
                    break
        if lead not in fetch or lead not in write:
            continue
        steps_f, steps_w = fetch[lead][1], write[lead][1]
        per_kernel = {}
        total = 0.0
        for k in sorted(set(fetch) | set(write)):
            if k == "scope_k0_kernel":  # input generation, not part of a step
                continue
            f_kb = fetch[k][0] / steps_f if k in fetch else 0.0
            w_kb = write[k][0] / steps_w if k in write else 0.0
            per_kernel[k] = {"FETCH_SIZE_KB_per_step": round(f_kb, 1), "WRITE_SIZE_KB_per_step": round(w_kb, 1),
                             "launches_per_step": round(fetch[k][1] / steps_f, 2) if k in fetch else None}
            total += 2.0 * f_kb + w_kb
        k0w = write.get("scope_k0_kernel")
        entries.append({
            "workload": wl, "frames_per_step": frames, "width": W, "height": H, "steps_averaged": [steps_f, steps_w],
            "kernels": per_kernel, "fetch_correction": 2.0,
            "k0_WRITE_SIZE_KB_per_launch": round(k0w[0] / k0w[1], 1) if k0w else None, "k0_output_KB": W * H * 4 / 1024,
            "bytes_per_launch": int(round(total * 1024)),
        })
    json.dump({"job": job, "kernel_sources_sha256_16": fingerprint(),
               "source": f"{out}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace, "
                         "bench.py --no-extras --no-cpu per workload and launch size (tools/gpu_job.sh traffic)",
               "note": "bytes_per_launch = HBM bytes per STEP: 2 x FETCH_SIZE + WRITE_SIZE summed over every scope kernel a step "
                       "launches (accumulation + partial-table reduction).  gfx950 FETCH_SIZE counts a wide coalesced stream at "
                       "half its bytes (MI355X_MICROARCH.md, HBM section): doubled; WRITE_SIZE taken as is (the K0 kernel's "
                       "WRITE_SIZE equals its output bytes)",
               "entries": entries}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
