#!/bin/bash
# borrowed lanes: trailing backwards on the idle forward streams - lanes tests, one-stage benches with / without, timeline
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3s
mkdir -p $O
cd $R
timeout -k 10 900 python -u -m pytest tests/test_gpu_dist.py -m gpu -x -q --timeout 400 --timeout-method thread -k "lanes or pipeline or hybrid" > $O/pytest_lanes.txt 2>&1 || { tail -60 $O/pytest_lanes.txt; exit 1; }
tail -2 $O/pytest_lanes.txt
run() {
  name=$1; shift; shift
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 "$@" > $O/$name.json 2> $O/$name.err || { tail -20 $O/$name.err; exit 1; }
  python -c "import json,sys; d=json.load(open('$O/$name.json')); print('%-40s %.3f ms  %.0f samples/s host %.2f' % ('$name', d['ms_per_step'], d['value'], d['host_issue_ms_per_step']))"
}
run pp1_parity_borrow -- --mode pp --preset parity
DDL_EXPERIMENTAL=1 DDL_LANE_BORROW=0 run pp1_parity_noborrow -- --mode pp --preset parity
run pp1_parity_gpipe_borrow -- --mode pp --preset parity --schedule gpipe
DDL_EXPERIMENTAL=1 DDL_LANE_BORROW=0 run pp1_parity_gpipe_noborrow -- --mode pp --preset parity --schedule gpipe
run pp1_mi355x_borrow -- --mode pp --preset mi355x
DDL_EXPERIMENTAL=1 DDL_LANE_BORROW=0 run pp1_mi355x_noborrow -- --mode pp --preset mi355x
run hyb1_parity_borrow -- --mode ddpnpp --pp-size 1 --preset parity
timeout -k 10 120 python bench/lane_timeline_probe.py --lanes 2 --fwd-lanes 2 > $O/lane_timeline_borrow.txt 2>&1; tail -14 $O/lane_timeline_borrow.txt
