#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc runs of bench/gpu_pmc.sh: per case, counters of the native kernel
(summed over its dispatches) and a few derived ratios."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc2"
cases = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sorted(glob.glob(os.path.join(root, "*_counter_collection.csv"))):
    case = os.path.basename(f).split("_SQ_")[0]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(s in k for s in ("conv_gemm_kernel", "conv3x3_halo_kernel", "wgrad1x1_kernel", "wgrad3x3_kernel")):
            continue
        c = cases[case]
        c[r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVES", "SQ_INSTS_LDS"):
            c["dispatches_" + r["Counter_Name"]] += 1
            c["ns_" + r["Counter_Name"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            c["vgpr"] = float(r["VGPR_Count"]) + float(r["Accum_VGPR_Count"])
            c["lds"] = float(r["LDS_Block_Size"])
for case, c in cases.items():
    n = max(c["dispatches_SQ_WAVES"], 1)
    us = c["ns_SQ_WAVES"] / n / 1e3
    mfma = c["SQ_INSTS_MFMA"]
    print(f"== {case}: {int(n)} dispatches, {us:.1f} us each (under counter collection), "
          f"{int(c['vgpr'])} VGPR+AGPR, {int(c['lds'])} B LDS/workgroup")
    print(f"   waves/dispatch {c['SQ_WAVES'] / n:.0f}   MFMA insts/wave {mfma / max(c['SQ_WAVES'], 1):.0f}   "
          f"VALU insts per MFMA {c['SQ_INSTS_VALU'] / max(mfma, 1):.2f}   SALU per MFMA {c['SQ_INSTS_SALU'] / max(mfma, 1):.2f}   "
          f"LDS insts per MFMA {c['SQ_INSTS_LDS'] / max(mfma, 1):.2f}   VMEM reads per MFMA {c['SQ_INSTS_VMEM_RD'] / max(mfma, 1):.3f}")
    util = (mfma / n) * 64 / max(us * 1e-6 * 2.4e9 * 1024, 1)
    print(f"   MFMA pipe utilisation {100 * util:.0f} %  (v_mfma_f32_32x32x2_f32 = 64 cycles per SIMD; 1024 SIMDs at 2.4 GHz)")
    print(f"   MFMA-busy / wave-cycles {c['SQ_VALU_MFMA_BUSY_CYCLES'] / max(c['SQ_WAVE_CYCLES'], 1):.3f}   "
          f"wait-any / wave-cycles {c['SQ_WAIT_ANY'] / max(c['SQ_WAVE_CYCLES'], 1):.3f}   "
          f"LDS bank-conflict cycles / LDS-active cycles {c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_ACTIVE_INST_LDS'], 1):.4f}")
    wc = max(c["SQ_WAVE_CYCLES"], 1)
    if c["SQ_ACTIVE_INST_ANY"]:
        # disjoint split of a wave's lifetime (MI355X_MICROARCH.md: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES):
        # parked at s_waitcnt / barrier, stalled at issue (MFMA RAW / pipe busy / LDS issue), issuing
        print(f"   wave lifetime: parked (s_waitcnt / barrier) {c['SQ_WAIT_ANY'] / wc:.3f} | issue-stalled {c['SQ_WAIT_INST_ANY'] / wc:.3f} | "
              f"issuing {c['SQ_ACTIVE_INST_ANY'] / wc:.3f}   (issuing: VALU {c['SQ_ACTIVE_INST_VALU'] / wc:.3f}, LDS {c['SQ_ACTIVE_INST_LDS'] / wc:.3f}, "
              f"MFMA {c['SQ_ACTIVE_INST_MFMA'] / wc:.3f})")
