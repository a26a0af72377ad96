#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2o
mkdir -p $O
timeout -k 10 1100 python -u -m pytest tests -m gpu -q --timeout 400 --timeout-method thread > $O/pytest_gpu.log 2>&1
rc=$?; tail -6 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python bench.py > $O/bench_default.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cut -c1-420 $O/bench_default.json
