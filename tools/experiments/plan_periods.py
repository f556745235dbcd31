"""Per-iteration periods of a plan from a rocprofv3 --kernel-trace CSV: for the queue of the first pipeline part, the start of each goal-set /
layer launch and the durations of the launches behind it.  python tools/experiments/plan_periods.py <kernel_trace.csv>"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith(("k_", "void k_"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last plan in the trace: find the last 210+ goalset launches
gs = [r for r in rows if "k_goalset_queue" in r["Kernel_Name"] or "k_goalset_range" in r["Kernel_Name"]]
qs = {}
for r in gs:
    qs.setdefault(r["Queue_Id"], []).append(r)
q0 = max(qs, key=lambda q: len(qs[q]))
mine = qs[q0][-70:]
t0 = int(mine[0]["Start_Timestamp"])
upd = [r for r in rows if r["Queue_Id"] == q0 and "update_optimize" in r["Kernel_Name"] or (r["Queue_Id"] == q0 and "chomp_optimize" in r["Kernel_Name"])]
prev = None
for i, r in enumerate(mine):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nxt = [u for u in upd if int(u["Start_Timestamp"]) >= e - 10]
    u = nxt[0] if nxt else None
    ud = (int(u["End_Timestamp"]) - int(u["Start_Timestamp"])) / 1e3 if u else float("nan")
    per = (s - prev) / 1e3 if prev is not None else float("nan")
    print(f"it {i:2d} start {(s - t0) / 1e3:8.1f} us  period {per:6.1f}  goalset {(e - s) / 1e3:6.1f} (wgs {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])})  update {ud:6.1f}")
    prev = s
