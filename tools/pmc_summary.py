#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter CSVs per kernel name (one line per counter)."""
import collections
import csv
import glob
import sys

import os

for d in sys.argv[1:]:
    if not os.path.isdir(d):
        continue
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "?").split("(")[0][:60]
            acc[name][row.get("Counter_Name")].append(float(row.get("Counter_Value", 0)))
    print("==", d)
    for name, ctrs in acc.items():
        if "scope_" not in name and "flush_" not in name:
            continue
        print("  ", name)
        for c, vals in sorted(ctrs.items()):
            print(f"      {c:28s} mean {sum(vals) / len(vals):16.1f}  n={len(vals)}")
