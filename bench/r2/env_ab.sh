#!/bin/bash
# same-box in-step A/B of environment knobs: bench/r2/env_ab.sh "VAR=a" "VAR=b" ... (each run twice, interleaved)
set -o pipefail
for rep in 1 2; do
  for kv in "$@"; do
    echo "== $kv"
    env $kv timeout -k 10 200 python bench.py --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" || exit 1
  done
done
