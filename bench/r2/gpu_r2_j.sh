#!/bin/bash
# knock-out timing experiment (results wrong by design): now bench/experiments/knockout.py - bench.py refuses DDL_KNOCKOUT
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2j
mkdir -p $O
for k in none wgrad bnprep; do
  timeout -k 10 200 python bench/experiments/knockout.py --knock $k > $O/bench_$k.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
  echo "knockout=$k: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$k.json | head -1)"
done
