#!/bin/bash
# A/B: fp32 forward / data gradients with bf16 weight gradients = upper bound of what faster fp32 wgrad kernels can buy
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3w
mkdir -p $O
cd $R
for b in 30 128; do
  for wp in 0 1; do
    DDL_EXPERIMENTAL=1 DDL_WGRAD_PREC=$wp timeout -k 10 300 python bench.py --steps 30 --warmup 8 --batch-per-gpu $b > $O/bench_b${b}_fp32_wgradprec$wp.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
    python -c "import json; d=json.load(open('$O/bench_b${b}_fp32_wgradprec$wp.json')); print('b$b fp32 wgrad prec $wp: %.3f ms %.0f samples/s' % (d['ms_per_step'], d['value']))"
  done
done
