#!/usr/bin/env python3
"""Prints mean counter values per launch of one kernel from rocprofv3 counter CSVs under a directory.
Usage: python profiles/pmc_table.py <dir> <kernel-substring>"""
import csv, glob, os, sys
d, kernel = sys.argv[1], sys.argv[2]
vals = {}
for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if kernel in row["Kernel_Name"]:
            vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
for k in sorted(vals):
    print(f"{k:40s} {sum(vals[k]) / len(vals[k]):.4g}  ({len(vals[k])} launches)")
