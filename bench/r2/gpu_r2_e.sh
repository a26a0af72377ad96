#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2e
mkdir -p $O
timeout -k 10 900 python -u -m pytest tests/test_gpu_dist.py tests/test_gpu_ops.py tests/test_gpu_kernels_real.py -m gpu -q --timeout 400 --timeout-method thread > $O/pytest.log 2>&1
rc=$?; tail -15 $O/pytest.log
exit $rc
