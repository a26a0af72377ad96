#!/bin/bash
# lanes: what limits the overlap?  hardware queues per process (GPU_MAX_HW_QUEUES, 4 by default), the weight-gradient
# fork inside every micro-batch graph, the lane count - and a kernel trace of the lanes=2 step
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3b
mkdir -p $O
cd $R
run() {  # name, env..., -- args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 200 python bench.py --mode pp --preset parity --steps 20 --warmup 5 "$@" > $O/$name.json 2> $O/$name.err || { tail -20 $O/$name.err; exit 1; }
  python -c "import json,sys; d=json.load(open('$O/$name.json')); print('%-40s %.3f ms  host %.2f' % ('$name', d['ms_per_step'], d['host_issue_ms_per_step']))"
}
run l1_q4 X=1 -- --lanes 1
run l2_q4 X=1 -- --lanes 2
run l2_q8 GPU_MAX_HW_QUEUES=8 -- --lanes 2
run l3_q8 GPU_MAX_HW_QUEUES=8 -- --lanes 3
run l2_q16 GPU_MAX_HW_QUEUES=16 -- --lanes 2
run l3_q16 GPU_MAX_HW_QUEUES=16 -- --lanes 3
run l4_q16 GPU_MAX_HW_QUEUES=16 -- --lanes 4
run l2_q4_nofork DDL_EXPERIMENTAL=1 DDL_LANE_WGRAD_FORK=0 -- --lanes 2
run l3_q4_nofork DDL_EXPERIMENTAL=1 DDL_LANE_WGRAD_FORK=0 -- --lanes 3
run l3_q8_nofork DDL_EXPERIMENTAL=1 DDL_LANE_WGRAD_FORK=0 GPU_MAX_HW_QUEUES=8 -- --lanes 3
run l4_q8_nofork DDL_EXPERIMENTAL=1 DDL_LANE_WGRAD_FORK=0 GPU_MAX_HW_QUEUES=8 -- --lanes 4
run l5_q16_nofork DDL_EXPERIMENTAL=1 DDL_LANE_WGRAD_FORK=0 GPU_MAX_HW_QUEUES=16 -- --lanes 5
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -o lanes2 -- python3 $R/bench.py --mode pp --preset parity --lanes 2 --steps 4 --warmup 2 > $O/prof_lanes2.out 2> $O/prof_lanes2.err || { tail -5 $O/prof_lanes2.err; exit 1; }
f=$(find $O -name "lanes2_kernel_trace.csv" | head -1)
python3 $R/bench/concurrency_report.py $f | tee $O/concurrency_lanes2.txt
gzip -f $f
