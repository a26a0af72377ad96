#!/bin/bash
# same-box in-step A/B of environment knobs: [BENCH_ARGS="--batch-per-gpu 512 --steps 20 --warmup 5"] bench/r2/env_ab.sh "VAR=a" "VAR=b" ...
# (each setting runs twice, interleaved)
set -o pipefail
for rep in 1 2; do
  for kv in "$@"; do
    echo "== $kv"
    env $kv timeout -k 10 300 python bench.py ${BENCH_ARGS:---steps 60 --warmup 10} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" || exit 1
  done
done
