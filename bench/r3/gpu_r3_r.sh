#!/bin/bash
# batch sweep + inference numbers on the round-3 build
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3r
mkdir -p $O
cd $R
for B in 128 512; do
  timeout -k 10 400 python bench.py --batch-per-gpu $B --steps 10 --warmup 3 > $O/bench_default_b$B.json 2> $O/b$B.err || { tail $O/b$B.err; exit 1; }
  python -c "import json; d=json.load(open('$O/bench_default_b$B.json')); print('fp32 B=$B %.1f samples/s %.2f ms' % (d['value'], d['ms_per_step']))"
done
timeout -k 10 400 python bench.py --mode pp --preset mi355x --steps 10 --warmup 3 > $O/bench_pp1_mi355x_lanes.json 2> $O/pp.err || { tail $O/pp.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench_pp1_mi355x_lanes.json')); print('pp1 mi355x lanes %.1f samples/s %.2f ms' % (d['value'], d['ms_per_step']))"
timeout -k 10 400 python bench.py --mode pp --preset mi355x-large --steps 6 --warmup 3 > $O/bench_pp1_mi355x_large_lanes.json 2> $O/ppl.err || { tail $O/ppl.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench_pp1_mi355x_large_lanes.json')); print('pp1 mi355x-large lanes %.1f samples/s %.2f ms' % (d['value'], d['ms_per_step']))"
timeout -k 10 400 python bench/eval_bench.py > $O/eval_bench.jsonl 2> $O/eval.err || { tail $O/eval.err; exit 1; }
tail -12 $O/eval_bench.jsonl | cut -c1-220
