#!/bin/bash
# software-pipelined LDS reads in conv_gemm (fp32): numerics, kernel times, step
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3ab
mkdir -p $O
cd $R
timeout -k 10 500 python -u -m pytest tests/test_gpu_kernels_real.py tests/test_gpu_native.py tests/test_gpu_ops.py -m gpu -x -q --timeout 180 --timeout-method thread > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
timeout -k 10 200 python bench/kernel_bench.py --batch 30 > $O/kernel_microbench_b30.txt 2>&1 || { tail -20 $O/kernel_microbench_b30.txt; exit 1; }
grep -v wgrad $O/kernel_microbench_b30.txt | grep "^b"
for b in 30 128; do
  timeout -k 10 300 python bench.py --steps 30 --warmup 8 --batch-per-gpu $b > $O/bench_b${b}.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
  python -c "import json; d=json.load(open('$O/bench_b${b}.json')); print('b$b fp32 %.3f ms %.0f samples/s' % (d['ms_per_step'], d['value']))"
done
