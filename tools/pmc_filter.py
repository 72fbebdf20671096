#!/usr/bin/env python3
"""Shrink a rocprofv3 --pmc output directory to the rows of the kernels of interest (the counter CSV of a torch program is
dominated by torch's own kernels with kilobyte-long template names): python3 tools/pmc_filter.py DIR SUBSTR[,SUBSTR...]"""
import csv
import glob
import os
import sys

d, subs = sys.argv[1], sys.argv[2].split(",")
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    keep = [r for r in rows if any(s in r["Kernel_Name"] for s in subs)]
    if rows:
        with open(f, "w", newline="") as out:
            w = csv.DictWriter(out, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(keep)
for f in glob.glob(os.path.join(d, "**", "*"), recursive=True):
    if os.path.isfile(f) and not f.endswith("counter_collection.csv") and not f.endswith(".log"):
        os.remove(f)
