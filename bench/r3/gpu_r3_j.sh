#!/bin/bash
# K-loop attribution of the three block-1 kernels the round-2 verdict names: PMC wave-lifetime split (parked / issue-stalled /
# issuing) on the fp32 MFMA path and with the MFMA 16x faster (bf16 inputs, --prec 1) - the knock-out that tells how much of
# the kernel time the matrix cores actually expose.  (rocprofv3 --att cannot be decoded on this image: the trace-decoder
# library is not installed, only its headers.)
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
export PMC_CASES="b1_fwd1x1_K160 b1_dgrad3x3 b1_dgrad1x1_N160 b1_wgrad1x1_C160"
PMC_OUT=pmc_r3_fp32 bash bench/gpu_pmc.sh || exit 1
export PMC_CASES="b1_fwd1x1_K160 b1_dgrad3x3 b1_dgrad1x1_N160"
PMC_OUT=pmc_r3_bf16 PMC_EXTRA="--prec 1" bash bench/gpu_pmc.sh || exit 1
cd $R
python bench/pmc_summary.py gpurun_out/pmc_r3_fp32 > gpurun_out/pmc_r3_fp32/summary.txt; cat gpurun_out/pmc_r3_fp32/summary.txt
python bench/pmc_summary.py gpurun_out/pmc_r3_bf16 > gpurun_out/pmc_r3_bf16/summary.txt; cat gpurun_out/pmc_r3_bf16/summary.txt
rm -f gpurun_out/pmc_r3_*/*.out; find gpurun_out/pmc_r3_fp32 gpurun_out/pmc_r3_bf16 -name "*.csv" -size +2M -delete
