#!/bin/bash
export DDL_EXPERIMENTAL=1   # experiment switches are ignored without it (ddl/utils/knobs.py)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2c
mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_kernels_real.py tests/test_gpu_native.py -m gpu -q -x --timeout 300 --timeout-method thread -k "wgrad or dense_block or full_model or transition or bn_bwd" > $O/pytest.log 2>&1
rc=$?; tail -8 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for sub in 0 1 3 4; do
  echo "== sub $sub"
  timeout -k 10 120 python bench/kernel_bench.py --only wgrad1x1 --sub $sub 2>&1 | grep wgrad
done | tee $O/wgrad_sub.txt
echo "== deferred (slab kernels alone)"
timeout -k 10 120 python bench/kernel_bench.py --only wgrad --defer 2>&1 | grep wgrad | tee $O/wgrad_defer.txt
timeout -k 10 200 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
cut -c1-330 $O/bench_default.json
DDL_DEFER_SLABS=0 timeout -k 10 200 python bench.py > $O/bench_nodefer.json 2> $O/bench_nodefer.err || { tail -20 $O/bench_nodefer.err; exit 1; }
cut -c1-330 $O/bench_nodefer.json
