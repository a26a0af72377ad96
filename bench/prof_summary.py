#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace of bench.py: per-kernel time of the LAST training step
(delimited by the fused-Adam launches), so MIOpen/warm-up noise is excluded."""
import collections
import csv
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof/native_kernel_trace.csv"
detail = len(sys.argv) > 2
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_flat" in r["Kernel_Name"]]
sel = rows[adam[-2] + 1: adam[-1] + 1]
t0, t1 = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
agg = collections.defaultdict(lambda: [0, 0])
busy = 0
for r in sel:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    busy += d
    n = r["Kernel_Name"]
    g = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    m = re.search(r"conv_gemm_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)", n)
    if m:
        n = "conv_gemm<%s,%s,ks%s,nbuf%s,pf%s,%s>" % (m.group(1), m.group(2), m.group(5), m.group(6), m.group(7),
                                                     ("fwd", "tdgrad", "dgrad3x3", "dgrad1x1")[int(m.group(8))])
    else:
        m = re.search(r"(wgrad_group_kernel<\d(?:, \w+)?>|wgrad\dx\d_kernel|slab_reduce_multi_kernel|slab_reduce_kernel|bn_prep_kernel|bn_bwd_prep_kernel|rows_op_kernel|"
                      r"adam_flat_kernel|ce_fwd_bwd_kernel|conv3x3_small_kernel|conv3x3_halo_kernel|stem_pack_kernel|normalize_u8_kernel|bnrelu_gap_fwd_kernel|stem_\w+_kernel|"
                      r"linear_\w+_kernel|adam_inc_step_kernel)", n)
        n = m.group(1) if m else re.sub(r"\(.*", "", n)[:60]
    if detail:
        n += " grid%s" % (g,)
    agg[n][0] += d
    agg[n][1] += 1
print(f"step span {1e-6 * (t1 - t0):.3f} ms, kernel busy {1e-6 * busy:.3f} ms, {len(sel)} launches")
for n, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[: (80 if detail else 24)]:
    print(f"{d / 1e6:8.3f} ms {c:5d} calls  avg {d / c / 1e3:8.1f} us  {n}")
