#!/usr/bin/env python3
"""Per-launch-group means of rocprofv3 --pmc counters for one kernel: python3 tools/pmc_groups.py DIR kernel_substr group_size"""
import collections, csv, glob, sys
d, sub, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
for f in sorted(glob.glob(d + "/**/*_counter_collection.csv", recursive=True)):
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            rows[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    names = sorted(rows)
    groups = len(rows[names[0]]) // n
    print("group " + " ".join("%16s" % c for c in names))
    for g in range(groups):
        vals = []
        for c in names:
            v = [x[1] for x in sorted(rows[c])[g * n:(g + 1) * n]]
            vals.append(sum(v) / len(v))
        print("%5d " % g + " ".join("%16.0f" % v for v in vals))
