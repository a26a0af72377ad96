#!/bin/bash
export DDL_EXPERIMENTAL=1   # experiment switches are ignored without it (ddl/utils/knobs.py)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2g
mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_native.py tests/test_gpu_kernels_real.py -m gpu -q -x --timeout 300 --timeout-method thread -k "dense_block or full_model or block_transition or head or trajectory or loss_curve or odd_batches" > $O/pytest.log 2>&1
rc=$?; tail -8 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for g in 1 auto 4 8 12 24; do
  DDL_WGRAD_GROUP=$g timeout -k 10 200 python bench.py > $O/bench_group_$g.json 2> $O/bench_group_$g.err || { tail -20 $O/bench_group_$g.err; exit 1; }
  echo "group=$g: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_group_$g.json | head -1) $(grep -o '"host_issue_ms_per_step": [0-9.]*' $O/bench_group_$g.json | head -1)"
done
