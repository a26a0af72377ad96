#!/bin/bash
# first GPU session: gpu tests, torch(MIOpen)-impl bench as the on-hardware baseline, kernel profile
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests -m gpu -x -v --timeout 120 --timeout-method thread 2>&1 | tail -25 > gpurun_out/pytest_gpu.log
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; tail -5 gpurun_out/pytest_gpu.log
timeout -k 10 300 python bench.py --impl torch --steps 20 --warmup 5 > gpurun_out/bench_torch_b30.json 2> gpurun_out/bench_torch_b30.err && cat gpurun_out/bench_torch_b30.json
timeout -k 10 300 python bench.py --impl torch --steps 10 --warmup 3 --batch-per-gpu 128 > gpurun_out/bench_torch_b128.json 2>> gpurun_out/bench_torch_b30.err && cat gpurun_out/bench_torch_b128.json
