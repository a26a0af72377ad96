#!/bin/bash
# in-step A/B of the weight-gradient group size per feature-map size: DDL_WGRAD_GROUPS="minM:G,..." (batch 30:
# block 1 M=94080, block 2 23520, block 3 5880, block 4 1470)
export DDL_EXPERIMENTAL=1   # experiment switches are ignored without it (ddl/utils/knobs.py)
set -o pipefail
for rules in "" "90000:4,20000:3,0:4" "90000:4,20000:6,0:4" "90000:4,20000:4,5000:3,0:4" "90000:4,20000:4,5000:6,0:4" "90000:4,20000:4,5000:8,0:4" "20000:4,5000:4,0:2" "20000:4,5000:4,0:8" "20000:4,5000:4,0:16" ""; do
  echo "== DDL_WGRAD_GROUPS=$rules"
  DDL_WGRAD_GROUPS=$rules timeout -k 10 200 python bench.py --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" || exit 1
done
