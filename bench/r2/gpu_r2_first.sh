#!/bin/bash
# round 2, first GPU session: the whole GPU suite (incl. the new real-shape kernel tests), the default bench, the
# forced-collectives DDP chain (RCCL, world of one) with the grad_sync A/B extras, and the pipeline presets on one stage.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2a
mkdir -p $O
timeout -k 10 1000 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread > $O/pytest_gpu.log 2>&1
rc=$?; tail -25 $O/pytest_gpu.log
[ $rc -eq 124 -o $rc -eq 137 -o $rc -eq 134 -o $rc -eq 139 ] && exit $rc
timeout -k 10 200 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
cut -c1-400 $O/bench_default.json
DDL_FORCE_COLLECTIVES=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --profile --ab on > $O/bench_forced.json 2> $O/bench_forced.err || { tail -20 $O/bench_forced.err; exit 1; }
grep '^{' $O/bench_forced.json | cut -c1-1500
for preset in parity mi355x; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --mode pp --preset $preset --steps 10 --warmup 3 > $O/bench_pp1_$preset.json 2> $O/bench_pp1_$preset.err || { tail -20 $O/bench_pp1_$preset.err; exit 1; }
  grep '^{' $O/bench_pp1_$preset.json | cut -c1-600
done
