#!/bin/bash
# full GPU suite + smoke + default bench (what the driver runs at round end)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/full
mkdir -p $O
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > $O/pytest_gpu.log 2>&1
rc=$?; tail -4 $O/pytest_gpu.log; [ $rc -ne 0 ] && { grep -n "FAILED\|Error" $O/pytest_gpu.log | head; tail -40 $O/pytest_gpu.log; exit $rc; }
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 200 python bench.py > $O/bench_default.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
tail -1 $O/bench_default.json | cut -c1-250
