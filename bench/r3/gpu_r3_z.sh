#!/bin/bash
# whole-step sweep of the grouped weight-gradient launch size after the kernel restructure
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3z
mkdir -p $O
cd $R
export DDL_EXPERIMENTAL=1
for w in 384 512 640 768 1024 1536; do
  DDL_WGRAD_GROUP_WGS=$w timeout -k 10 300 python bench.py --steps 30 --warmup 8 > $O/bench_b30_gw$w.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
  python -c "import json; d=json.load(open('$O/bench_b30_gw$w.json')); print('group wgs $w: %.3f ms %.0f samples/s' % (d['ms_per_step'], d['value']))"
done
for l in 81920 65536; do
  DDL_WGRAD_GROUP_LDS9=$l timeout -k 10 300 python bench.py --steps 30 --warmup 8 > $O/bench_b30_lds9_$l.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
  python -c "import json; d=json.load(open('$O/bench_b30_lds9_$l.json')); print('group lds9 $l: %.3f ms %.0f samples/s' % (d['ms_per_step'], d['value']))"
done
