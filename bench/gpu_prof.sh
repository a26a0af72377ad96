#!/bin/bash
# kernel-level profile of the native training step (rocprofv3 kernel trace + stats)
set -o pipefail
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o native -- python3 $R/bench.py --steps 6 --warmup 3 ${BENCH_ARGS} > $R/gpurun_out/prof/bench.out 2> $R/gpurun_out/prof/bench.err
echo "rc $?"; tail -2 $R/gpurun_out/prof/bench.out; find $R/gpurun_out/prof -name "*stats*" | head
