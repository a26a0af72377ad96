#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3k
mkdir -p $O
cd $R
timeout -k 10 900 python bench/kernel_bench.py --sweep > $O/ksweep_b30_96rows.txt 2>&1 || { tail -30 $O/ksweep_b30_96rows.txt; exit 1; }
cat $O/ksweep_b30_96rows.txt
