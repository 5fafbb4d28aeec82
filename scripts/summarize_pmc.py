#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter_collection.csv files per dispatch of one kernel (full-width dispatches only)."""
import csv, glob, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
kernel = sys.argv[2] if len(sys.argv) > 2 else "resident_segment<false"
out = collections.OrderedDict()
for path in sorted(glob.glob(root + "/pmc_*/*/*counter_collection.csv")):
    sums, counts = collections.Counter(), collections.Counter()
    with open(path) as f:
        for row in csv.DictReader(f):
            if kernel not in row["Kernel_Name"]:
                continue
            if int(row["Grid_Size"]) != 256 * int(row["Workgroup_Size"]):
                continue
            sums[row["Counter_Name"]] += float(row["Counter_Value"])
            counts[row["Counter_Name"]] += 1
    for name in sums:
        out[name] = (sums[name] / counts[name], counts[name])
for name, (avg, n) in out.items():
    print(f"{name:28s} {avg:16.1f}  per dispatch over {n} dispatches")
