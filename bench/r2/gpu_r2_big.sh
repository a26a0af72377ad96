#!/bin/bash
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/big
mkdir -p $O
timeout -k 10 900 python -u -m pytest tests/test_gpu_native.py tests/test_gpu_kernels_real.py tests/test_gpu_ops.py -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest_regress.log 2>&1
rc=$?; tail -3 $O/pytest_regress.log; [ $rc -ne 0 ] && { tail -40 $O/pytest_regress.log; exit $rc; }
timeout -k 10 900 python -u -m pytest tests/test_gpu_bigviews.py -m gpu -x -v --timeout 400 --timeout-method thread > $O/pytest_big.log 2>&1
rc=$?; grep -E "PASSED|FAILED|ERROR|passed|failed" $O/pytest_big.log | tail -12; [ $rc -ne 0 ] && { tail -60 $O/pytest_big.log; exit $rc; }
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_b30.json 2>$O/bench.err && tail -1 $O/bench_b30.json | cut -c1-200
timeout -k 10 400 python bench.py --steps 6 --warmup 3 --batch-per-gpu 1024 > $O/bench_b1024.json 2>>$O/bench.err && tail -1 $O/bench_b1024.json | cut -c1-200 || tail -20 $O/bench.err
