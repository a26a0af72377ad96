#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2p
mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_native.py tests/test_gpu_dist.py -m gpu -q -x -s --timeout 400 --timeout-method thread -k "partial_dense or auto_split or pipeline_two_stages or hybrid_graphed" > $O/pytest.log 2>&1
rc=$?; grep -E "auto split|reference split|passed|failed|Error" $O/pytest.log | tail -12
[ $rc -ne 0 ] && { tail -30 $O/pytest.log; exit $rc; }
for preset in parity mi355x; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --mode pp --preset $preset --steps 10 --warmup 3 > $O/bench_pp1_$preset.json 2> $O/bench_pp1_$preset.err || { tail -20 $O/bench_pp1_$preset.err; exit 1; }
  grep '^{' $O/bench_pp1_$preset.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config'].get('preset'), d['ms_per_step'], d['value'], d['host_issue_ms_per_step'])"
done
# two / four pipeline ranks sharing the one GPU (gloo): parity vs mi355x preset, 1f1b
for cfg in "2 parity" "2 mi355x" "4 mi355x"; do
  set -- $cfg
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus $1 --mode pp --preset $2 --backend gloo --steps 8 --warmup 3 --profile > $O/bench_pp$1_$2.json 2> $O/bench_pp$1_$2.err || { tail -20 $O/bench_pp$1_$2.err; exit 1; }
  grep '^{' $O/bench_pp$1_$2.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pp$1', d['config'].get('preset'), d['ms_per_step'], d['value'], d['config'].get('split_points'), {k:v for k,v in d.get('profile',{}).items() if 'p2p' in k or 'compute' in k})"
done
