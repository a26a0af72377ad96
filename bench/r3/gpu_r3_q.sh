#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3q
mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_fences.py -m gpu -x -q --timeout 400 --timeout-method thread > $O/pytest_fences.txt 2>&1 || { tail -40 $O/pytest_fences.txt; exit 1; }
tail -2 $O/pytest_fences.txt
