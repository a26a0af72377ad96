#!/bin/bash
# one optimisation iteration: numerics tests, bench at B=30 (+B=128), kernel profile of the last step
set -o pipefail
mkdir -p gpurun_out/prof
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_native.py tests/test_gpu_ops.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/pytest_native.log 2>&1
rc=$?; tail -5 gpurun_out/pytest_native.log; [ $rc -ne 0 ] && { tail -40 gpurun_out/pytest_native.log; exit $rc; }
timeout -k 10 300 python bench.py --steps 20 --warmup 5 ${BENCH_ARGS} > gpurun_out/bench_native_b30.json 2> gpurun_out/bench_native.err || { tail -20 gpurun_out/bench_native.err; exit 1; }
cat gpurun_out/bench_native_b30.json
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --batch-per-gpu 128 ${BENCH_ARGS} > gpurun_out/bench_native_b128.json 2>> gpurun_out/bench_native.err && cat gpurun_out/bench_native_b128.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o native -- python3 $R/bench.py --steps 4 --warmup 3 ${BENCH_ARGS} > $R/gpurun_out/prof/bench.out 2> $R/gpurun_out/prof/bench.err
echo "prof rc $?"
