#!/bin/bash
# same-box A/B of prebuilt extension variants: bench/ab_variants.sh v1 v2 ...
set -e
for v in "$@"; do
  cp ab/_ddl_hip_$v.so distributed-deep-learning_amd/csrc/_ddl_hip.so
  timeout -k 10 120 python bench.py --steps 60 --warmup 10 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', r['ms_per_step'], r['value'])" | tee -a gpurun_out/ab.txt
done
