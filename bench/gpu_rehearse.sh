#!/bin/bash
# multi-rank rehearsal on ONE GPU (ranks share cuda:0, gloo host-staged transport): exercises the
# bench.py distributed modes end to end; absolute numbers are NOT representative of xGMI/RCCL.
set -o pipefail
mkdir -p gpurun_out/rehearse
run() {  # name nproc args...
  name=$1; n=$2; shift 2
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
    bench.py --gpus $n --steps 10 --warmup 5 --backend gloo "$@" > gpurun_out/rehearse/$name.json 2> gpurun_out/rehearse/$name.err
  rc=$?; echo "$name rc=$rc $(tail -1 gpurun_out/rehearse/$name.json | cut -c1-160)"
  if [ $rc -ne 0 ]; then tail -30 gpurun_out/rehearse/$name.err; exit $rc; fi
}
run ddp2 2 --mode ddp &&
run ddp4 4 --mode ddp &&
run ddp2_rsag 2 --mode ddp --grad-sync rs_ag &&
run ddp2_zero1 2 --mode ddp --zero1 &&
run ddp2_nograph 2 --mode ddp --no-graph &&
run pp2_1f1b 2 --mode pp &&
run pp2_1f1b_nograph 2 --mode pp --no-graph &&
run pp2_gpipe 2 --mode pp --schedule gpipe &&
run pp4 4 --mode pp &&
run ddpnpp4 4 --mode ddpnpp --pp-size 2 --batch-per-gpu 10
