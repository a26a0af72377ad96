#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2h
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o trace -- python3 $R/bench.py --steps 6 --warmup 3 > $O/prof_bench.out 2> $O/prof_bench.err || { tail -5 $O/prof_bench.err; exit 1; }
cd $R && python bench/prof_summary.py $O/trace_kernel_trace.csv > $O/kernel_profile.txt; head -30 $O/kernel_profile.txt
