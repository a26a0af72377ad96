#!/bin/bash
# kernel traces of the default step with fused statistics on / off (eager launches, last step)
export DDL_EXPERIMENTAL=1   # experiment switches are ignored without it (ddl/utils/knobs.py)
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
for f in 1 0; do
  export DDL_FUSED_BWD=$f
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof -o fused$f -- python3 $R/bench.py --steps 4 --warmup 3 ${BENCH_ARGS} > $R/gpurun_out/prof/bench$f.out 2> $R/gpurun_out/prof/bench$f.err || exit 1
done
cd $R
for f in 1 0; do echo "== DDL_FUSED_BWD=$f"; python bench/prof_summary.py gpurun_out/prof/fused${f}_kernel_trace.csv | head -9; done
