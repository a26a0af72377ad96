#!/bin/bash
export DDL_EXPERIMENTAL=1   # experiment switches are ignored without it (ddl/utils/knobs.py)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2n
mkdir -p $O
for cfg in "0 0" "90000 0" "90000 90000" "0 60000" "90000 60000"; do
  set -- $cfg
  DDL_WGRAD_GROUP_LDS1=$1 DDL_WGRAD_GROUP_LDS9=$2 timeout -k 10 200 python bench.py > $O/bench_$1_$2.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
  echo "lds1=$1 lds9=$2: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$1_$2.json | head -1)"
done
