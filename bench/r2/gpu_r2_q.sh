#!/bin/bash
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r2q
mkdir -p $O
export DDL_SPLIT_DEBUG=1
timeout -k 10 120 python -u bench/split_probe.py > $O/a.log 2>&1 || { echo A failed; tail -25 $O/a.log; exit 1; }
tail -2 $O/a.log
timeout -k 10 120 python -u bench/split_probe.py --points features.denseblock2.denselayer7 > $O/b.log 2>&1 || { echo B failed; tail -25 $O/b.log; exit 1; }
tail -2 $O/b.log
timeout -k 10 120 python -u bench/split_probe.py --units-first > $O/c.log 2>&1 || { echo C failed; tail -25 $O/c.log; exit 1; }
tail -2 $O/c.log
timeout -k 10 120 python -u bench/split_probe.py --auto 2 > $O/d.log 2>&1 || { echo D failed; tail -25 $O/d.log; exit 1; }
tail -2 $O/d.log
