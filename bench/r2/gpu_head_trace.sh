#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2/head
timeout -k 10 400 python -u -m pytest tests/test_gpu_native.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
timeout -k 10 200 python bench.py --steps 60 --warmup 10 2>/dev/null | cut -c1-200 || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r2/head -o h -- python3 $R/bench.py --steps 4 --warmup 3 > $R/gpurun_out/r2/head/b.out 2> $R/gpurun_out/r2/head/b.err || exit 1
cd $R
f=$(ls gpurun_out/r2/head/*h_kernel_trace.csv gpurun_out/r2/head/*/*h_kernel_trace.csv 2>/dev/null | head -1)
python bench/fwd_timeline.py $f --skip 146 --count 22
gzip -f $f
