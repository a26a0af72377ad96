#!/bin/bash
export DDL_EXPERIMENTAL=1   # experiment switches are ignored without it (ddl/utils/knobs.py)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2d
mkdir -p $O
for pf in 1 0; do for tg in 512 768 1024; do
  echo "== sub1 pf2=$pf target=$tg"
  DDL_WGRAD_PF2=$pf DDL_WGRAD_TARGET1=$tg timeout -k 10 120 python bench/kernel_bench.py --only wgrad1x1 --sub 1 --batch 30 2>&1 | grep -E "b3_|b4_|b2_wgrad1x1_C160"
done; done | tee $O/wgrad_sub1.txt
echo "== heuristic"
timeout -k 10 120 python bench/kernel_bench.py --only wgrad1x1 2>&1 | grep wgrad | tee $O/wgrad_heur.txt
for d in layer 0 block; do
  DDL_DEFER_SLABS=$d timeout -k 10 200 python bench.py > $O/bench_defer_$d.json 2> $O/bench_defer_$d.err || { tail -20 $O/bench_defer_$d.err; exit 1; }
  echo "defer=$d: $(cut -c1-260 $O/bench_defer_$d.json | grep -o '"ms_per_step": [0-9.]*')"
done
