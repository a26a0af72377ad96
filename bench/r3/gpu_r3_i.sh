#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3i
mkdir -p $O
cd $R
timeout -k 10 200 python bench/split_accuracy.py > $O/split_accuracy.txt 2>&1 || { tail -30 $O/split_accuracy.txt; exit 1; }
cat $O/split_accuracy.txt
for P in 0 3 1; do
  timeout -k 10 300 python bench/kernel_bench.py --prec $P > $O/kbench_prec$P.txt 2>&1 || { tail -30 $O/kbench_prec$P.txt; exit 1; }
done
paste <(grep -E "fwd|dgrad" $O/kbench_prec0.txt | awk '{print $1, $2}') <(grep -E "fwd|dgrad" $O/kbench_prec3.txt | awk '{print $2}') <(grep -E "fwd|dgrad" $O/kbench_prec1.txt | awk '{print $2}')
timeout -k 10 600 python bench/kernel_bench.py --prec 3 --sweep > $O/ksweep_prec3.txt 2>&1 || { tail -30 $O/ksweep_prec3.txt; exit 1; }
cat $O/ksweep_prec3.txt
