#!/bin/bash
# wave-quantisation probe: the block-1 kernels at batch sizes around one / two rounds of resident workgroups
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3o
mkdir -p $O
cd $R
for B in 20 24 26 27 28 30 34 40 48 52 53 56 60; do
  timeout -k 10 120 python bench/kernel_bench.py --batch $B --only b1_ --iters 30 2>/dev/null | grep -E "b1_fwd1x1_K160|b1_fwd3x3 |b1_dgrad3x3|b1_dgrad1x1_N160|b1_wgrad1x1_C160|b1_wgrad3x3" | awk -v b=$B '{printf "B=%d tiles64=%d %s %s us %s TF\n", b, (b*3136+63)/64, $1, $2, $4}'
done | tee $O/quantisation_b1.txt
