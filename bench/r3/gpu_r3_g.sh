#!/bin/bash
# owned RCCL communicators + fence tests: the probe itself (all parts, output kept), then the new tests, the RCCL-touching
# tests of the old suite, and the forced-collective benches for both transports (host issue time of a DDP step)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3g
mkdir -p $O
cd $R
export DDL_FORCE_COLLECTIVES=1 DDL_P2P_ALLOW_SELF=1
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench/fence_probe.py all > $O/fence_probe.txt 2>&1 || { tail -40 $O/fence_probe.txt; exit 1; }
grep "^FENCE" $O/fence_probe.txt
unset DDL_FORCE_COLLECTIVES DDL_P2P_ALLOW_SELF
timeout -k 10 900 python -u -m pytest tests/test_gpu_fences.py tests/test_gpu_dist.py -m gpu -x -q --timeout 400 --timeout-method thread > $O/pytest_fences_dist.txt 2>&1 || { tail -60 $O/pytest_fences_dist.txt; exit 1; }
tail -3 $O/pytest_fences_dist.txt
export DDL_FORCE_COLLECTIVES=1 DDL_P2P_ALLOW_SELF=1
for T in direct c10d; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 1 --steps 30 --warmup 5 --rccl $T > $O/bench_forced_ddp_$T.json 2> $O/bench_forced_ddp_$T.err || { tail -30 $O/bench_forced_ddp_$T.err; exit 1; }
  python -c "import json; d=json.load(open('$O/bench_forced_ddp_$T.json')); print('ddp forced $T: %.3f ms host %.2f ms transport %s ladder %s' % (d['ms_per_step'], d['host_issue_ms_per_step'], d['config']['rccl_transport'], len(d.get('first_contact_ladder', []))))"
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29543 bench.py --gpus 1 --mode pp --preset parity --steps 20 --warmup 5 --rccl $T > $O/bench_forced_pp1_$T.json 2> $O/bench_forced_pp1_$T.err || { tail -30 $O/bench_forced_pp1_$T.err; exit 1; }
  python -c "import json; d=json.load(open('$O/bench_forced_pp1_$T.json')); print('pp1 forced $T: %.3f ms host %.2f ms transport %s' % (d['ms_per_step'], d['host_issue_ms_per_step'], d['config']['rccl_transport']))"
done
grep "stage '" $O/bench_forced_pp1_direct.err | tail -20
