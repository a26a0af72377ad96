#!/bin/bash
# bf16 mode after the bf16 weight gradients: batch 512 number, kernel profile at batch 30 / 128
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3v
mkdir -p $O
cd $R
export TMPDIR=/tmp
[ -n "$SKIP512" ] || timeout -k 10 300 python bench.py --steps 20 --warmup 5 --batch-per-gpu 512 --dtype bf16 > $O/bench_b512_bf16.json 2> $O/bench_b512_bf16.err || { tail -20 $O/bench_b512_bf16.err; exit 1; }
[ -n "$SKIP512" ] || python -c "import json; d=json.load(open('$O/bench_b512_bf16.json')); print('b512 bf16 %.3f ms %.0f samples/s' % (d['ms_per_step'], d['value']))"
for b in 30 128; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b$b -o p -- python3 $R/bench.py --steps 10 --warmup 4 --batch-per-gpu $b --dtype bf16 > $O/prof_b$b.log 2>&1 || { tail -20 $O/prof_b$b.log; exit 1; }
  f=$(find $O/prof_b$b -name '*kernel_trace.csv' | head -1)
  python bench/prof_summary.py $f > $O/kernel_stats_bf16_b$b.txt 2>&1 || cp $f $O/kernel_stats_bf16_b$b.csv
  head -30 $O/kernel_stats_bf16_b$b.txt
  rm -rf $O/prof_b$b
done
