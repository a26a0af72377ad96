#!/bin/bash
# full GPU suite + headline + pipeline / hybrid reference shapes on the current build
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3m
mkdir -p $O
cd $R
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > $O/pytest_gpu.txt 2>&1 || { tail -60 $O/pytest_gpu.txt; exit 1; }
tail -3 $O/pytest_gpu.txt
run() {
  name=$1; shift; shift
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 "$@" > $O/$name.json 2> $O/$name.err || { tail -20 $O/$name.err; exit 1; }
  python -c "import json,sys; d=json.load(open('$O/$name.json')); print('%-40s %.3f ms  %.0f samples/s host %.2f' % ('$name', d['ms_per_step'], d['value'], d['host_issue_ms_per_step']))"
}
run bench_default_b30 --
run bench_fp32x3_b30 -- --dtype fp32x3
run bench_bf16_b30 -- --dtype bf16
run bench_pp1_parity_lanes -- --mode pp --preset parity
run bench_hyb1_parity_lanes -- --mode ddpnpp --pp-size 1 --preset parity
