#!/bin/bash
# rocprofv3 kernel trace + stats of the round-3 build: the default step, and the reference pipeline shape with lanes
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o default -- python3 $R/bench.py --steps 6 --warmup 3 > $O/default.out 2> $O/default.err || { tail -5 $O/default.err; exit 1; }
f=$(find $O -name "default_kernel_trace.csv" | head -1)
python3 $R/bench/prof_summary.py $f > $O/kernel_profile_b30.txt; cat $O/kernel_profile_b30.txt
gzip -f $f
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o lanes -- python3 $R/bench.py --mode pp --preset parity --steps 4 --warmup 2 > $O/lanes.out 2> $O/lanes.err || { tail -5 $O/lanes.err; exit 1; }
f=$(find $O -name "lanes_kernel_trace.csv" | head -1)
python3 $R/bench/concurrency_report.py $f > $O/concurrency_pp1_parity_lanes.txt; cat $O/concurrency_pp1_parity_lanes.txt
s=$(find $O -name "lanes_kernel_stats.csv" | head -1); head -25 $s > $O/kernel_stats_pp1_parity_lanes.csv
gzip -f $f
cd $R && timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
