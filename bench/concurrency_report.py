#!/usr/bin/env python3
"""How concurrent is a step?  From a rocprofv3 kernel trace (CSV or .csv.gz), over the last traced step (between the
last two fused-Adam kernels): wall span, union of kernel intervals, sum of kernel durations, the share of the span
with 0 / 1 / 2 / 3+ kernels in flight, and the busy time per hardware queue.  Written for the micro-batch lanes of
the pipeline stage (parallel/pp.py): lanes only pay if their kernels really overlap on the device.

    python bench/concurrency_report.py <kernel_trace.csv[.gz]>"""
import argparse
import csv
import gzip
from collections import defaultdict


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    a = ap.parse_args()
    rows = list(csv.DictReader(gzip.open(a.csv, "rt") if a.csv.endswith(".gz") else open(a.csv)))
    ks = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(ks) if "adam_flat" in r["Kernel_Name"]]
    step = ks[ends[-2] + 1:ends[-1] + 1]
    t0 = int(ks[ends[-2]]["End_Timestamp"])
    t1 = int(step[-1]["End_Timestamp"])
    span = (t1 - t0) / 1e3
    ev = []
    per_q = defaultdict(float)
    total = 0.0
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        ev.append((s, 1))
        ev.append((e, -1))
        per_q[r["Queue_Id"]] += (e - s) / 1e3
        total += (e - s) / 1e3
    ev.sort()
    depth, last, hist = 0, t0, defaultdict(float)
    for t, d in ev:
        hist[min(depth, 4)] += (t - last) / 1e3
        last, depth = t, depth + d
    print(f"last step: {len(step)} kernels, span {span:.1f} us, sum of kernel times {total:.1f} us "
          f"(average concurrency {total / span:.2f})")
    for k in sorted(hist):
        print(f"  {k}{'+' if k == 4 else ' '} kernels in flight: {hist[k]:9.1f} us  {100 * hist[k] / span:5.1f} %")
    for q, t in sorted(per_q.items(), key=lambda kv: -kv[1]):
        print(f"  queue {q:>3}: busy {t:9.1f} us  {100 * t / span:5.1f} % of the span")


if __name__ == "__main__":
    main()
