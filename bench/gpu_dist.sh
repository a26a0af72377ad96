#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_gpu_dist.py -m gpu -x -v --timeout 600 --timeout-method thread > gpurun_out/pytest_gpu_dist.log 2>&1
rc=$?; tail -60 gpurun_out/pytest_gpu_dist.log; exit $rc
