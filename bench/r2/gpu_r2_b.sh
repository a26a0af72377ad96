#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2b
mkdir -p $O
timeout -k 10 120 python bench/dbg_fusedk.py 2>&1 | tail -5
timeout -k 10 200 python bench/host_issue_probe.py 2>&1 | tail -4 | tee $O/host_issue_probe.txt
for mr in 32 64 128 256 512; do
  echo "== min_rows $mr"
  timeout -k 10 120 python bench/kernel_bench.py --min-rows $mr 2>&1 | grep wgrad
done | tee $O/wgrad_min_rows.txt
