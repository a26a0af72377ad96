#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c
mkdir -p $O
cd $R
timeout -k 10 120 python bench/lane_timeline_probe.py --lanes 2 2>&1 | tee $O/lane_timeline_l2.txt || exit 1
DDL_EXPERIMENTAL=1 DDL_LANE_WGRAD_FORK=0 timeout -k 10 120 python bench/lane_timeline_probe.py --lanes 2 2>&1 | tee $O/lane_timeline_l2_nofork.txt || exit 1
DDL_EXPERIMENTAL=1 DDL_LANE_WGRAD_FORK=0 timeout -k 10 120 python bench/lane_timeline_probe.py --lanes 3 --schedule gpipe 2>&1 | tee $O/lane_timeline_l3_gpipe_nofork.txt || exit 1
