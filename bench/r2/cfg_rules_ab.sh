#!/bin/bash
# in-step A/B of tile configurations: one DDL_CFG_RULES override per run, same box, batch 30
# rule = kind:taps:minM:maxM:cfg   (M at batch 30: block 1 94080, block 2 23520, block 3 5880, block 4 1470)
export DDL_EXPERIMENTAL=1   # experiment switches are ignored without it (ddl/utils/knobs.py)
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/cfgab
mkdir -p $O
run() {
  DDL_CFG_RULES="$1" timeout -k 10 120 python bench.py --steps 40 --warmup 8 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%-28s %.3f' % ('${1:-baseline}', r['ms_per_step']))" | tee -a $O/results.txt
}
for rule in "$@"; do run "$rule" || exit 1; done
