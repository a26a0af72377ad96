#!/usr/bin/env python3
"""Summarise bench/gpu_hbm.sh: per case the bytes fetched / written through L2 per dispatch of the native kernel and
where the kernel sits against the two roofs (157 TFLOP/s fp32 MFMA, ~5 TB/s achievable HBM3E).

    python bench/hbm_summary.py gpurun_out/hbm [undisturbed kernel_bench listing]

Counter collection serialises the dispatches, so kernel TIMES come from an undisturbed `bench/kernel_bench.py`
listing ("name  us  TFLOP/s") when one is given; the FLOPs of a case are us x TFLOP/s of either listing."""
import collections
import csv
import glob
import os
import re
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/hbm"
NATIVE = ("conv_gemm_kernel", "conv3x3_halo_kernel", "conv3x3_small_kernel", "wgrad1x1_kernel", "wgrad3x3_kernel",
          "wgrad_group_kernel")
LINE = re.compile(r"^(\S+)\s+([\d.]+) us\s+([\d.]+) TFLOP/s")


def listing(path):
    out = {}
    if path and os.path.exists(path):
        for line in open(path):
            m = LINE.match(line)
            if m:
                out[m.group(1)] = (float(m.group(2)), float(m.group(3)))
    return out


clean = listing(sys.argv[2] if len(sys.argv) > 2 else None)
cases = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sorted(glob.glob(os.path.join(root, "*_counter_collection.csv"))):
    m = re.match(r"(.*)_(FETCH_SIZE|WRITE_SIZE)_counter_collection.csv", os.path.basename(f))
    if not m:
        continue
    case, metric = m.groups()
    for r in csv.DictReader(open(f)):
        if any(s in r["Kernel_Name"] for s in NATIVE) and r["Counter_Name"] == metric:
            cases[case][metric] += float(r["Counter_Value"])
            cases[case]["n_" + metric] += 1
print("case                     read MB  write MB  time us   TB/s  TFLOP/s  FLOP/B   MFMA floor us  HBM floor us (5 TB/s)   bound by")
for case, c in cases.items():
    run = listing(os.path.join(root, f"{case}_FETCH_SIZE.out")).get(case)
    us, tf = clean.get(case, run or (0.0, 0.0))
    flop = (run[0] * run[1] if run else us * tf) * 1e6
    rd = c["FETCH_SIZE"] / max(c["n_FETCH_SIZE"], 1) * 1024           # KiB -> bytes per dispatch
    wr = c["WRITE_SIZE"] / max(c["n_WRITE_SIZE"], 1) * 1024
    f_mfma, f_hbm = flop / 157e12 * 1e6, (rd + wr) / 5e12 * 1e6
    roof = max(f_mfma, f_hbm)
    print(f"{case:22s} {rd / 1e6:9.1f} {wr / 1e6:9.1f} {us:8.1f} {(rd + wr) / max(us, 1e-9) / 1e6:6.2f} {flop / max(us, 1e-9) / 1e6:8.1f}"
          f" {flop / max(rd + wr, 1):7.1f} {f_mfma:15.1f} {f_hbm:12.1f}            {'HBM' if f_hbm > f_mfma else 'MFMA'}"
          f" ({100 * roof / max(us, 1e-9):.0f} % of the roof)")
