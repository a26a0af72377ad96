#!/bin/bash
# lanes as the default of the pipeline modes: full suite, reference-shaped one-stage numbers, the large-micro-batch
# presets with and without lanes, kernel trace + concurrency report of the final configuration
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3e
mkdir -p $O
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest_gpu.txt 2>&1 || { tail -60 $O/pytest_gpu.txt; exit 1; }
tail -3 $O/pytest_gpu.txt
run() {  # name, -- args
  name=$1; shift; shift
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 "$@" > $O/$name.json 2> $O/$name.err || { tail -20 $O/$name.err; exit 1; }
  python -c "import json,sys; d=json.load(open('$O/$name.json')); print('%-40s %.3f ms  %.0f samples/s host %.2f' % ('$name', d['ms_per_step'], d['value'], d['host_issue_ms_per_step']))"
}
run bench_default_b30 -- 
run bench_pp1_parity_serial -- --mode pp --preset parity --lanes 1
run bench_pp1_parity_lanes -- --mode pp --preset parity
run bench_hyb1_parity_serial -- --mode ddpnpp --pp-size 1 --preset parity --lanes 1
run bench_hyb1_parity_lanes -- --mode ddpnpp --pp-size 1 --preset parity
run bench_pp1_mi355x_serial -- --mode pp --preset mi355x --lanes 1
run bench_pp1_mi355x_lanes -- --mode pp --preset mi355x
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -o lanes -- python3 $R/bench.py --mode pp --preset parity --steps 4 --warmup 2 > $O/prof_lanes.out 2> $O/prof_lanes.err || { tail -5 $O/prof_lanes.err; exit 1; }
f=$(find $O -name "lanes_kernel_trace.csv" | head -1)
python3 $R/bench/concurrency_report.py $f | tee $O/concurrency_lanes_2x2.txt
gzip -f $f
cd $R && timeout -k 10 120 python bench/lane_timeline_probe.py --lanes 2 --fwd-lanes 2 > $O/lane_timeline_2x2.txt 2>&1; cat $O/lane_timeline_2x2.txt | tail -12
