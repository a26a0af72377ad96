#!/bin/bash
# round-end check: all GPU tests, smoke(), the default bench, and a kernel trace of the default step
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $R/gpurun_out/final/pytest_gpu.log 2>&1 || { tail -30 $R/gpurun_out/final/pytest_gpu.log; exit 1; }
tail -2 $R/gpurun_out/final/pytest_gpu.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $R/gpurun_out/final/smoke.log 2>&1 || { tail -20 $R/gpurun_out/final/smoke.log; exit 1; }
tail -1 $R/gpurun_out/final/smoke.log
timeout -k 10 200 python bench.py > $R/gpurun_out/final/bench_default.json 2> $R/gpurun_out/final/bench_default.err || exit 1
grep '^{' $R/gpurun_out/final/bench_default.json | cut -c1-220
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final -o v11 -- python3 $R/bench.py --steps 6 --warmup 3 > $R/gpurun_out/final/prof_bench.out 2> $R/gpurun_out/final/prof_bench.err || exit 1
cd $R && python bench/prof_summary.py gpurun_out/final/v11_kernel_trace.csv > gpurun_out/final/kernel_profile_v11.txt && head -12 gpurun_out/final/kernel_profile_v11.txt
