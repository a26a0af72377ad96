#!/bin/bash
# normal build after the weight-gradient restructure: wgrad tests, native-model tests, step time at batch 30 / 128
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3y
mkdir -p $O
cd $R
timeout -k 10 400 python -u -m pytest tests/test_gpu_kernels_real.py tests/test_gpu_native.py tests/test_gpu_bigviews.py -m gpu -x -q --timeout 180 --timeout-method thread > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
timeout -k 10 200 python bench/kernel_bench.py --batch 30 --only wgrad > $O/kernel_microbench_b30_wgrad.txt 2>&1 || { tail -20 $O/kernel_microbench_b30_wgrad.txt; exit 1; }
cat $O/kernel_microbench_b30_wgrad.txt | grep wgrad
for b in 30 128; do
  for dt in fp32 bf16; do
    timeout -k 10 300 python bench.py --steps 30 --warmup 8 --batch-per-gpu $b --dtype $dt > $O/bench_b${b}_$dt.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
    python -c "import json; d=json.load(open('$O/bench_b${b}_$dt.json')); print('b$b $dt %.3f ms %.0f samples/s' % (d['ms_per_step'], d['value']))"
  done
done
