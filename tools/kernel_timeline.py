#!/usr/bin/env python3
"""Print the last N kernel dispatches of a rocprofv3 --kernel-trace CSV as a timeline (start us, duration us, queue, name)."""
import csv
import glob
import sys

d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("k_", "void k_", "(anonymous"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f %8.1f q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:44]))
