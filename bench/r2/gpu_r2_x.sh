#!/bin/bash
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r2x
mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_native.py tests/test_gpu_kernels_real.py -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1
rc=$?; tail -2 $O/pytest.log; [ $rc -ne 0 ] && { tail -40 $O/pytest.log; exit $rc; }
timeout -k 10 200 python bench/kernel_bench.py --only dgrad > $O/kb_dgrad.txt 2>&1; grep dgrad $O/kb_dgrad.txt
for i in 1 2; do timeout -k 10 200 python bench.py --steps 40 --warmup 10 2>/dev/null | grep "^{" | cut -c135-165; done
