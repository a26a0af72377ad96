#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3h
mkdir -p $O
cd $R
export DDL_FORCE_COLLECTIVES=1 DDL_P2P_ALLOW_SELF=1
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench/fence_probe.py steps > $O/fence_probe.txt 2>&1 || { tail -40 $O/fence_probe.txt; exit 1; }
grep "^FENCE\|^steps" $O/fence_probe.txt
