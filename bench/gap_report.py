#!/usr/bin/env python3
"""Idle gaps of the last traced step of a rocprofv3 kernel trace (CSV or .csv.gz): every interval longer than
--min us in which NO kernel of the process is running, with the kernels on either side.  Shows what a graph
boundary, an eager launch between graphs or a cross-stream fence costs in the DDP chain.

    python bench/gap_report.py <kernel_trace.csv[.gz]> [--min 6]"""
import argparse
import csv
import gzip


def short(n):
    return n.replace("void ", "").replace("ddl::", "").replace("(anonymous namespace)::", "").split("(")[0][:48]


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--min", type=float, default=6.0)
    a = ap.parse_args()
    rows = list(csv.DictReader(gzip.open(a.csv, "rt") if a.csv.endswith(".gz") else open(a.csv)))
    ks = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(ks) if "adam_flat" in r["Kernel_Name"]]
    step = ks[ends[-2] + 1:ends[-1] + 1]
    t0 = int(ks[ends[-2]]["End_Timestamp"])
    print(f"last step: {len(step)} kernels, {(int(step[-1]['End_Timestamp']) - t0) / 1e3:.1f} us from the previous Adam's end")
    busy_until, last, total = t0, ks[ends[-2]], 0.0
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s > busy_until:
            gap = (s - busy_until) / 1e3
            total += gap
            if gap >= a.min:
                print(f"  {(busy_until - t0) / 1e3:8.1f} us  idle {gap:6.1f} us   after {short(last['Kernel_Name'])} (q{last['Queue_Id']})"
                      f"   before {short(r['Kernel_Name'])} (q{r['Queue_Id']})")
        if e > busy_until:
            busy_until, last = e, r
    print(f"idle in all: {total:.1f} us")


if __name__ == "__main__":
    main()
