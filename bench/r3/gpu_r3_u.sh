#!/bin/bash
# bf16 weight gradients: numerics, kernel times, whole-step effect
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3u
mkdir -p $O
cd $R
timeout -k 10 400 python -u -m pytest tests/test_gpu_kernels_real.py -m gpu -x -q --timeout 120 --timeout-method thread -k "bf16 or wgrad" > $O/pytest_wgrad.txt 2>&1 || { tail -40 $O/pytest_wgrad.txt; exit 1; }
tail -3 $O/pytest_wgrad.txt
timeout -k 10 300 python -u -m pytest tests/test_gpu_native.py -m gpu -x -q --timeout 120 --timeout-method thread -k "bf16" > $O/pytest_native_bf16.txt 2>&1 || { tail -40 $O/pytest_native_bf16.txt; exit 1; }
tail -3 $O/pytest_native_bf16.txt
timeout -k 10 200 python bench/kernel_bench.py --batch 30 --prec 1 > $O/kernel_microbench_b30_prec1_wgrad.txt 2>&1 || { tail -20 $O/kernel_microbench_b30_prec1_wgrad.txt; exit 1; }
grep -i wgrad $O/kernel_microbench_b30_prec1_wgrad.txt
for b in 30 128; do
  for dt in bf16 fp32; do
    timeout -k 10 300 python bench.py --steps 30 --warmup 8 --batch-per-gpu $b --dtype $dt > $O/bench_b${b}_$dt.json 2> $O/bench_b${b}_$dt.err || { tail -20 $O/bench_b${b}_$dt.err; exit 1; }
    python -c "import json; d=json.load(open('$O/bench_b${b}_$dt.json')); print('b$b $dt %.3f ms %.0f samples/s' % (d['ms_per_step'], d['value']))"
  done
done
