#!/bin/bash
# kernel + native-engine GPU tests, then the default bench (quick check of a kernel / engine change);
# QUICK_AB="VAR=a VAR=b" adds a same-box A/B of an environment knob
set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_gpu_kernels_real.py tests/test_gpu_native.py tests/test_gpu_dist.py -x -q --timeout 200 --timeout-method thread 2>&1 | tail -3 || exit 1
for rep in 1 2; do
  for kv in ${QUICK_AB:-X=0}; do
    echo "== $kv"
    env $kv timeout -k 10 200 python bench.py --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" || exit 1
  done
done
