"""Where does a timed region of bench.py spend what it costs besides its steps?  From a rocprofv3 --kernel-trace CSV of a short-region run
(e.g. --steps 20 --regions 5): the regions (separated by idle gaps > --gap us), each one's GPU span, and the durations / start
offsets of its first goal-set and update launches beside the steady state's.
    python tools/trace_regions.py <kernel_trace.csv> [--gap 150]"""
import argparse
import csv
import statistics as st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--gap", type=float, default=150.0)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.csv)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ev = [(int(r["Start_Timestamp"]) / 1e3, int(r["End_Timestamp"]) / 1e3, r["Kernel_Name"]) for r in rows]
    regions, cur, last_end = [], [], None
    for s, e, n in ev:
        if last_end is not None and s - last_end > a.gap and cur:
            regions.append(cur)
            cur = []
        cur.append((s, e, n))
        last_end = e if last_end is None else max(last_end, e)
    if cur:
        regions.append(cur)
    for k, reg in enumerate(regions):
        gs = [(s, e) for s, e, n in reg if "k_goalset_queue<2, false, false, false" in n]
        up = [(s, e) for s, e, n in reg if "k_update_optimize_split" in n]
        if len(gs) < 8:
            continue
        t0, t1 = reg[0][0], max(e for _, e, _ in reg)
        d = [e - s for s, e in gs]
        print(f"region {k}: {len(gs)} goal-set launches, span {t1 - t0:8.1f} us = {(t1 - t0) / (len(gs) / 2):6.1f} us per step; first kernel {reg[0][2][:30]!r}; "
              f"goal-set durations: first four {[round(x) for x in d[:4]]}, median of the rest {st.median(d[4:]):.0f}; update: first two {[round(e - s) for s, e in up[:2]]}, "
              f"median {st.median([e - s for s, e in up]):.0f}; last goal-set end -> region end {t1 - gs[-1][1]:.0f} us; kernels after the last update: "
              f"{[n[:24] for s, e, n in reg if s >= up[-1][1]]}")


if __name__ == "__main__":
    main()
