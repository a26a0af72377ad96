#!/bin/bash
# numerics tests, then the per-kernel microbenchmark (default configs)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_native.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/pytest_native.log 2>&1
rc=$?; tail -3 gpurun_out/pytest_native.log; [ $rc -ne 0 ] && { tail -40 gpurun_out/pytest_native.log; exit $rc; }
timeout -k 10 300 python bench/kernel_bench.py ${KB_ARGS} > gpurun_out/kbench.txt 2>&1; rc=$?
grep -v Warn gpurun_out/kbench.txt; exit $rc
