#!/bin/bash
export DDL_EXPERIMENTAL=1   # experiment switches are ignored without it (ddl/utils/knobs.py)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2l
mkdir -p $O
for cfg in "1 0 auto" "1 1 auto" "0 0 auto" "1 0 4"; do
  set -- $cfg
  DDL_TAIL_PREP=$1 DDL_TAIL_FENCE=$2 DDL_WGRAD_GROUP=$3 timeout -k 10 200 python bench.py > $O/bench_t$1_f$2_g$3.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
  echo "tail=$1 fence=$2 group=$3: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_t$1_f$2_g$3.json | head -1)"
done
