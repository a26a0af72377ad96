#!/bin/bash
# weight-gradient kernel experiments on a probe build: numerics, then per-workgroup timings
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3x
mkdir -p $O/dump
cd $R
export DDL_EXPERIMENTAL=1
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels_real.py -m gpu -x -q --timeout 120 --timeout-method thread -k "wgrad" > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
timeout -k 10 120 python bench/kernel_bench.py --batch 30 --probe --dump $O/dump --only wgrad > $O/wgrad_probe.txt 2>&1 || { tail -20 $O/wgrad_probe.txt; exit 1; }
grep "workgroups:\|us$" $O/wgrad_probe.txt
