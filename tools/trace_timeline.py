"""Text timeline of a rocprofv3 --kernel-trace CSV: the last `--last` dispatches by start time, per queue, in us relative to the first.
    python tools/trace_timeline.py <kernel_trace.csv> [--last 16]"""
import argparse
import csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--last", type=int, default=16)
    ap.add_argument("--skip-tail", type=int, default=8, help="ignore this many dispatches at the very end (drain)")
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.csv)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[: len(rows) - a.skip_tail][-a.last:]
    t0 = int(rows[0]["Start_Timestamp"])
    queues = sorted({r["Queue_Id"] for r in rows})
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:34]
        print(f'q{queues.index(r["Queue_Id"])} {name:34s} wgs {int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])):5d} '
              f'start {(int(r["Start_Timestamp"]) - t0) / 1e3:8.1f}  end {(int(r["End_Timestamp"]) - t0) / 1e3:8.1f}  dur {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:7.1f}')


if __name__ == "__main__":
    main()
