#!/bin/bash
# round 3, first GPU run: the suite, the headline, the first-contact ladder on RCCL (world of one, forced collectives +
# self transfers), and the reference-shaped pipeline presets on ONE stage (the serial baseline the lanes must beat)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3a
mkdir -p $O
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest_gpu.txt 2>&1 || { tail -60 $O/pytest_gpu.txt; exit 1; }
tail -3 $O/pytest_gpu.txt
timeout -k 10 200 python bench.py > $O/bench_default_b30.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
cat $O/bench_default_b30.json
DDL_FORCE_COLLECTIVES=1 DDL_P2P_ALLOW_SELF=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 30 --warmup 5 --grad-sync auto > $O/bench_forced_ddp.json 2> $O/bench_forced_ddp.err || { tail -30 $O/bench_forced_ddp.err; exit 1; }
cat $O/bench_forced_ddp.json; grep "stage '" $O/bench_forced_ddp.err | tail -30
DDL_FORCE_COLLECTIVES=1 DDL_P2P_ALLOW_SELF=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --mode pp --preset parity --steps 20 --warmup 5 > $O/bench_forced_pp1.json 2> $O/bench_forced_pp1.err || { tail -30 $O/bench_forced_pp1.err; exit 1; }
cat $O/bench_forced_pp1.json; grep "stage '" $O/bench_forced_pp1.err | tail -30
timeout -k 10 200 python bench.py --mode pp --preset parity --steps 20 --warmup 5 > $O/bench_pp1_parity.json 2> $O/bench_pp1.err || { tail -20 $O/bench_pp1.err; exit 1; }
cat $O/bench_pp1_parity.json
timeout -k 10 200 python bench.py --mode pp --preset parity --schedule gpipe --steps 20 --warmup 5 > $O/bench_pp1_parity_gpipe.json 2> $O/bench_pp1g.err || { tail -20 $O/bench_pp1g.err; exit 1; }
cat $O/bench_pp1_parity_gpipe.json
timeout -k 10 200 python bench.py --mode ddpnpp --pp-size 1 --preset parity --steps 20 --warmup 5 > $O/bench_hyb1_parity.json 2> $O/bench_hyb1.err || { tail -20 $O/bench_hyb1.err; exit 1; }
cat $O/bench_hyb1_parity.json
for L in 2 3; do
  timeout -k 10 200 python bench.py --mode pp --preset parity --lanes $L --steps 20 --warmup 5 > $O/bench_pp1_parity_lanes$L.json 2> $O/bench_pp1_l$L.err || { tail -20 $O/bench_pp1_l$L.err; exit 1; }
  cat $O/bench_pp1_parity_lanes$L.json
done
timeout -k 10 200 python bench.py --mode pp --preset parity --schedule gpipe --lanes 2 --steps 20 --warmup 5 > $O/bench_pp1_parity_gpipe_lanes2.json 2> $O/bench_pp1g2.err || { tail -20 $O/bench_pp1g2.err; exit 1; }
cat $O/bench_pp1_parity_gpipe_lanes2.json
timeout -k 10 200 python bench.py --mode ddpnpp --pp-size 1 --preset parity --lanes 2 --steps 20 --warmup 5 > $O/bench_hyb1_parity_lanes2.json 2> $O/bench_hyb1l2.err || { tail -20 $O/bench_hyb1l2.err; exit 1; }
cat $O/bench_hyb1_parity_lanes2.json
