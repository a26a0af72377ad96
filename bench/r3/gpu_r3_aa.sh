#!/bin/bash
# experiment: staggered workgroup starts in conv_gemm (break the lock-step of co-resident workgroups)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3aa
mkdir -p $O
cd $R
export DDL_EXPERIMENTAL=1
for st in 0 1 2 4; do
  echo "== stagger $st"
  DDL_CONV_STAGGER=$st timeout -k 10 200 python bench/kernel_bench.py --batch 30 > $O/kb_st$st.txt 2>&1 || { tail -20 $O/kb_st$st.txt; exit 1; }
  grep "^b1_\|^b2_" $O/kb_st$st.txt | grep -v wgrad
  DDL_CONV_STAGGER=$st timeout -k 10 300 python bench.py --steps 30 --warmup 8 > $O/bench_b30_st$st.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
  python -c "import json; d=json.load(open('$O/bench_b30_st$st.json')); print('stagger $st: %.3f ms %.0f samples/s' % (d['ms_per_step'], d['value']))"
done
