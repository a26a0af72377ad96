#!/bin/bash
# HBM traffic of representative native kernels: rocprofv3 derived metrics FETCH_SIZE / WRITE_SIZE (KiB moved between
# L2 and the fabric), one pass per run, own process each.  Summary: bench/hbm_summary.py.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/hbm
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for case in ${HBM_CASES:-b1_fwd1x1_K160 b1_fwd3x3 b1_dgrad3x3 b1_dgrad1x1_N160 b1_wgrad1x1_C160 b1_wgrad3x3 b2_dgrad1x1_N320 b3_fwd1x1_K640}; do
  for pass in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 90 rocprofv3 --pmc $pass --output-format csv -d $O -o ${case}_$pass -- python3 $R/bench/kernel_bench.py --only ${case} --iters 3 > $O/${case}_$pass.out 2>&1
    rc=$?; echo "hbm $case $pass rc $rc"
    if [ $rc -ne 0 ]; then tail -5 $O/${case}_$pass.out; exit $rc; fi
  done
done
cd $R
python bench/hbm_summary.py $O
rm -f $O/*agent_info.csv
