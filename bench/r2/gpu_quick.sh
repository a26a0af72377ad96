#!/bin/bash
# kernel + native-engine GPU tests, then the default bench twice (quick check of a kernel / engine change)
set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_gpu_kernels_real.py tests/test_gpu_native.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
for i in 1 2; do
  timeout -k 10 200 python bench.py --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" || exit 1
done
