#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per (kernel, counter) dispatch count, sum, mean.

    python tools/pmc_summary.py <rocprof output dir> <out.csv>
"""
import collections
import csv
import glob
import sys

d, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(d + "/**/*counter_collection*.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:70], r["Counter_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
w = csv.writer(open(out, "w"))
w.writerow(["kernel", "counter", "dispatches", "sum", "mean_per_dispatch"])
for (k, c), (n, s) in sorted(agg.items()):
    if k.startswith(("k_", "void k_")):
        w.writerow([k, c, n, s, s / n])
