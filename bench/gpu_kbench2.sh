#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for bm in 0 128 64 32; do
  echo "== force_bm=$bm"; timeout -k 10 200 python bench/kernel_bench.py --bm $bm 2>&1 | grep -E "b1_|b2_" | grep -v wgrad
done
