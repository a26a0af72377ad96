#!/bin/bash
# rocprofv3 PMC counters for representative native kernels (one counter pass per run, own process each)
set -o pipefail
R=$GRAFT_REPO_ROOT
PMC_OUT=${PMC_OUT:-pmc2}
mkdir -p $R/gpurun_out/$PMC_OUT
cd /tmp && export TMPDIR=/tmp
for case in ${PMC_CASES:-b1_fwd1x1_K160 b1_fwd3x3 b1_dgrad1x1_N160 b3_fwd1x1_K640 b1_wgrad1x1_C160 b1_wgrad3x3}; do
  for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MFMA" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY"; do
    tag=$(echo $pass | cut -d' ' -f1)
    timeout -s KILL 90 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/$PMC_OUT -o ${case}_$tag -- python3 $R/bench/kernel_bench.py --only ${case} --iters 3 ${PMC_EXTRA:-} > $R/gpurun_out/$PMC_OUT/${case}_$tag.out 2>&1
    rc=$?; echo "pmc $case $tag rc $rc"
    if [ $rc -ne 0 ]; then tail -5 $R/gpurun_out/$PMC_OUT/${case}_$tag.out; exit $rc; fi
  done
done
