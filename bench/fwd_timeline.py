#!/usr/bin/env python3
"""Timeline of the forward kernels of one dense block from a rocprofv3 kernel trace (CSV): start offset, duration,
queue of every kernel in a window of the LAST traced step - shows whether the side-stream prefix passes of the
split 1x1 (native_densenet.py) really run next to the chain.

    python bench/fwd_timeline.py <kernel_trace.csv> [--grid-x 184] [--count 40]

--grid-x selects the window: it starts at the first kernel of the last step whose grid has that many workgroups
in x (184 = block 3 at batch 30 with 32-row tiles)."""
import argparse
import csv


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--count", type=int, default=40)
    ap.add_argument("--skip", type=int, default=0, help="kernels into the last step where the window starts")
    a = ap.parse_args()
    import gzip
    rows = list(csv.DictReader(gzip.open(a.csv, "rt") if a.csv.endswith(".gz") else open(a.csv)))
    ks = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
    # steps are delimited by the Adam kernel
    ends = [i for i, r in enumerate(ks) if "adam_flat" in r["Kernel_Name"]]
    lo = ends[-2] + 1 if len(ends) >= 2 else 0
    step = ks[lo:ends[-1] + 1]
    t0 = int(step[0]["Start_Timestamp"])
    print(f"last step: {len(step)} kernels, {(int(step[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
    for r in step[a.skip:a.skip + a.count]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"]
        short = name.replace("void ", "").replace("ddl::", "").replace("(anonymous namespace)::", "").split("(")[0][:64]
        grid = f'{r.get("Grid_Size_X", "?")}/{r.get("Workgroup_Size_X", "?")}'
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:6.1f}  q{r.get('Queue_Id', '?'):>2}  grid {grid:>12}  {short}")


if __name__ == "__main__":
    main()
