#!/usr/bin/env python
"""Aggregate rocprofv3 counter_collection CSVs (one dir per pass) per kernel.
Usage: pmc_summary.py <pmc_dir> [out.txt]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    agg = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> values
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines = []
    for k in sorted(agg, key=lambda k: -sum(agg[k].get("SQ_WAVE_CYCLES", [0]))):
        c = agg[k]
        lines.append(f"== {k[:110]}  (dispatches: {max(len(v) for v in c.values())})")
        for name in sorted(c):
            v = c[name]
            lines.append(f"   {name:28s} avg/dispatch = {sum(v) / len(v):16.1f}")
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
