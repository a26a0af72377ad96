#!/bin/bash
# 2 data-parallel ranks sharing the one GPU over gloo: graphed vs eager, with traces + overlap analysis
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2y
mkdir -p $O
for g in graph nograph; do
  extra=""; [ $g = nograph ] && extra="--no-graph"
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 2 --mode ddp --backend gloo --steps 10 --warmup 4 --ab off --trace-dir $O/trace_ddp2_$g $extra > $O/bench_ddp2_$g.json 2> $O/bench_ddp2_$g.err || { tail -20 $O/bench_ddp2_$g.err; exit 1; }
  grep "^{" $O/bench_ddp2_$g.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$g', d['ms_per_step'], d.get('profile'))"
  python bench/trace_overlap.py $O/trace_ddp2_$g | tee $O/trace_ddp2_$g/overlap.txt
  head -25 $O/trace_ddp2_$g/kernels_rank0.txt | cut -c1-160
  gzip -9 $O/trace_ddp2_$g/trace_rank*.json
done
