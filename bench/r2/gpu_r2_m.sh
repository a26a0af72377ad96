#!/bin/bash
export DDL_EXPERIMENTAL=1   # experiment switches are ignored without it (ddl/utils/knobs.py)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2m
mkdir -p $O
for cfg in "3 1024" "4 512" "4 2048" "6 1024" "4 768"; do
  set -- $cfg
  DDL_WGRAD_GROUP=$1 DDL_WGRAD_GROUP_WGS=$2 timeout -k 10 200 python bench.py > $O/bench_g$1_w$2.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
  echo "group=$1 wgs=$2: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_g$1_w$2.json | head -1)"
done
cd /tmp && export TMPDIR=/tmp
DDL_WGRAD_GROUP=4 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o trace -- python3 $R/bench.py --steps 6 --warmup 3 > $O/prof_bench.out 2> $O/prof_bench.err || { tail -5 $O/prof_bench.err; exit 1; }
cd $R && python bench/prof_summary.py $O/trace_kernel_trace.csv > $O/kernel_profile.txt; head -16 $O/kernel_profile.txt
