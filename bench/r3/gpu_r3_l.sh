#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3l
mkdir -p $O
cd $R
for i in 1 2; do
timeout -k 10 200 python bench.py --steps 40 --warmup 8 > $O/base_$i.json 2> $O/base_$i.err || { tail $O/base_$i.err; exit 1; }
DDL_EXPERIMENTAL=1 DDL_CFG_RULES="2:9:16000:32000:24" timeout -k 10 200 python bench.py --steps 40 --warmup 8 > $O/rule_$i.json 2> $O/rule_$i.err || { tail $O/rule_$i.err; exit 1; }
python -c "import json; a=json.load(open('$O/base_$i.json')); b=json.load(open('$O/rule_$i.json')); print('base %.3f ms   96-row dgrad3x3 rule %.3f ms' % (a['ms_per_step'], b['ms_per_step']))"
done
