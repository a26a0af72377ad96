#!/bin/bash
# round-2 evidence bundle: headline bench, forced-collectives DDP chain (host issue, grad-sync A/B, bucket sweep),
# rocprofv3 kernel profile, kernel microbench, host-issue probe, 2-rank pipeline traces (graph vs eager).
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ev
mkdir -p $O
step() { echo "== $1"; }
step "default bench" &&
timeout -k 10 200 python bench.py --steps 40 --warmup 10 > $O/bench_default_b30.json 2> $O/bench_default.err &&
tail -1 $O/bench_default_b30.json | cut -c1-220 &&
step "forced collectives" &&
DDL_FORCE_COLLECTIVES=1 timeout -k 10 300 python bench.py --steps 40 --warmup 10 --profile --ab on > $O/bench_forced_collectives_b30.json 2> $O/bench_forced.err &&
tail -1 $O/bench_forced_collectives_b30.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'host', d['host_issue_ms_per_step'], d['host_bound'], d.get('grad_sync_ab'), d.get('bucket_mb_sweep'), d.get('profile'))" &&
step "batch sweep" &&
for b in 128 512; do timeout -k 10 300 python bench.py --steps 12 --warmup 4 --batch-per-gpu $b > $O/bench_default_b$b.json 2>> $O/bench_default.err || exit 1; tail -1 $O/bench_default_b$b.json | cut -c1-120; done &&
step "host issue probe" &&
timeout -k 10 200 python bench/host_issue_probe.py > $O/host_issue_probe.txt 2>&1 && tail -6 $O/host_issue_probe.txt &&
step "kernel microbench" &&
timeout -k 10 300 python bench/kernel_bench.py > $O/kernel_microbench_b30.txt 2>&1 && tail -40 $O/kernel_microbench_b30.txt &&
step "pp2 traces" &&
for g in graph nograph; do
  extra=""; [ $g = nograph ] && extra="--no-graph"
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29621 bench.py --gpus 2 --mode pp --backend gloo --steps 10 --warmup 4 --profile --trace-dir $O/trace_pp2_$g $extra > $O/bench_pp2_1f1b_$g.json 2> $O/bench_pp2_$g.err || { tail -20 $O/bench_pp2_$g.err; exit 1; }
  tail -1 $O/bench_pp2_1f1b_$g.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$g', d['ms_per_step'], d.get('profile'))"
  python bench/trace_overlap.py $O/trace_pp2_$g | tee $O/trace_pp2_$g/overlap.txt
  gzip -9 $O/trace_pp2_$g/trace_rank*.json; ls -la $O/trace_pp2_$g
done &&
step "kernel profile" &&
cd /tmp && export TMPDIR=/tmp &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o trace -- python3 $R/bench.py --steps 6 --warmup 3 > $O/prof_bench.out 2> $O/prof_bench.err &&
cd $R && python bench/prof_summary.py $(find $O/prof -name "*kernel_trace.csv" | head -1) > $O/kernel_profile_b30.txt && head -30 $O/kernel_profile_b30.txt
find $O/prof -name "*kernel_trace.csv" -exec gzip -9 {} \;
du -sh $R/gpurun_out
