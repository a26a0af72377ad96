#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -v --timeout 180 --timeout-method thread > gpurun_out/pytest_native.log 2>&1
rc=$?; tail -30 gpurun_out/pytest_native.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_native_b30.json 2> gpurun_out/bench_native.err || { tail -20 gpurun_out/bench_native.err; exit 1; }
cat gpurun_out/bench_native_b30.json
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --batch-per-gpu 128 > gpurun_out/bench_native_b128.json 2>> gpurun_out/bench_native.err && cat gpurun_out/bench_native_b128.json
