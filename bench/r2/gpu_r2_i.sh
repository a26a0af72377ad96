#!/bin/bash
export DDL_EXPERIMENTAL=1   # experiment switches are ignored without it (ddl/utils/knobs.py)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2i
mkdir -p $O
for cfg in "0 1" "1 1" "0 auto"; do
  set -- $cfg
  DDL_FUSED_BWD=$1 DDL_WGRAD_GROUP=$2 timeout -k 10 200 python bench.py > $O/bench_f$1_g$2.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
  echo "fused_bwd=$1 group=$2: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_f$1_g$2.json | head -1)"
done
