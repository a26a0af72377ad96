#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3t
mkdir -p $O
cd $R
run() {
  name=$1; shift; shift
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 "$@" > $O/$name.json 2> $O/$name.err || { tail -20 $O/$name.err; exit 1; }
  python -c "import json,sys; d=json.load(open('$O/$name.json')); print('%-40s %.3f ms  %.0f samples/s host %.2f' % ('$name', d['ms_per_step'], d['value'], d['host_issue_ms_per_step']))"
}
for cfg in "2 2" "2 3" "3 2" "3 3" "2 1" "4 2"; do
  set -- $cfg
  run pp1_parity_l$1_f$2 -- --mode pp --preset parity --lanes $1 --fwd-lanes $2
done
for cfg in "2 2" "2 3" "3 2"; do
  set -- $cfg
  run hyb1_parity_l$1_f$2 -- --mode ddpnpp --pp-size 1 --preset parity --lanes $1 --fwd-lanes $2
done
