#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/pmc
R=$GRAFT_REPO_ROOT
timeout -k 10 200 python bench/kernel_bench.py > gpurun_out/kbench.txt 2>&1; cat gpurun_out/kbench.txt | grep -v Warn
cd /tmp && export TMPDIR=/tmp
timeout -k 10 60 rocprofv3 --list-avail > $R/gpurun_out/pmc/avail.txt 2>&1; echo "avail rc $?"
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MFMA" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 90 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmc -o ${PMC_CASE}_$tag -- python3 $R/bench/kernel_bench.py --only ${PMC_CASE} --iters 3 > $R/gpurun_out/pmc/${PMC_CASE}_$tag.out 2>&1
  echo "pmc $tag rc $?"
done
