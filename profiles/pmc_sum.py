#!/usr/bin/env python3
"""Sums counter values over all launches of one kernel from rocprofv3 counter CSVs under a directory.
Usage: python profiles/pmc_sum.py <dir> <kernel-substring>"""
import csv, glob, os, sys
d, kernel = sys.argv[1], sys.argv[2]
vals, n = {}, {}
for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if kernel in row["Kernel_Name"]:
            k = row["Counter_Name"]
            vals[k] = vals.get(k, 0.0) + float(row["Counter_Value"]); n[k] = n.get(k, 0) + 1
for k in sorted(vals):
    print(f"{k:36s} {vals[k]:.4g}  ({n[k]} launches)")
