#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3d
mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_dist.py -x -q -k "lanes" --timeout 300 --timeout-method thread > $O/pytest_lanes.txt 2>&1 || { tail -60 $O/pytest_lanes.txt; exit 1; }
tail -3 $O/pytest_lanes.txt
run() {  # name, env..., -- args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 200 python bench.py --steps 20 --warmup 5 "$@" > $O/$name.json 2> $O/$name.err || { tail -20 $O/$name.err; exit 1; }
  python -c "import json,sys; d=json.load(open('$O/$name.json')); print('%-40s %.3f ms  host %.2f' % ('$name', d['ms_per_step'], d['host_issue_ms_per_step']))"
}
run pp_l2_f1 X=1 -- --mode pp --preset parity --lanes 2
run pp_l2_f2 X=1 -- --mode pp --preset parity --lanes 2 --fwd-lanes 2
run pp_l1_f2 X=1 -- --mode pp --preset parity --lanes 1 --fwd-lanes 2
run pp_l3_f2 X=1 -- --mode pp --preset parity --lanes 3 --fwd-lanes 2
run pp_l2_f3 X=1 -- --mode pp --preset parity --lanes 2 --fwd-lanes 3
run pp_l2_f2_gpipe X=1 -- --mode pp --preset parity --lanes 2 --fwd-lanes 2 --schedule gpipe
run hyb_l2_f2 X=1 -- --mode ddpnpp --pp-size 1 --preset parity --lanes 2 --fwd-lanes 2
timeout -k 10 120 python bench/lane_timeline_probe.py --lanes 2 --fwd-lanes 2 2>&1 | tee $O/lane_timeline_l2_f2.txt || exit 1
