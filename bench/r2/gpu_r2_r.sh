#!/bin/bash
# kernel timeline of the one-stage pipeline step (5 micro-batches of 6)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2r
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o pp1 -- python3 $R/bench.py --mode pp --steps 6 --warmup 3 > $O/bench.out 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
tail -1 $O/bench.out | cut -c1-200
cd $R && python bench/prof_summary.py $(find $O/prof -name "*kernel_trace.csv" | head -1) > $O/kernel_profile_pp1.txt; head -30 $O/kernel_profile_pp1.txt
find $O/prof -name "*kernel_trace.csv" -exec gzip -9 {} \;
