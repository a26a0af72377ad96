#!/bin/bash
# unguarded MFMA stream (GUARD = false instantiations): kernel microbench + in-step A/B (DDL_FORCE_GUARD=1 = the old form)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3n
mkdir -p $O
cd $R
timeout -k 10 300 python bench/kernel_bench.py > $O/kbench_unguarded.txt 2>&1 || { tail -30 $O/kbench_unguarded.txt; exit 1; }
DDL_EXPERIMENTAL=1 DDL_FORCE_GUARD=1 timeout -k 10 300 python bench/kernel_bench.py > $O/kbench_guarded.txt 2>&1 || { tail -30 $O/kbench_guarded.txt; exit 1; }
paste <(grep -E "fwd|dgrad" $O/kbench_guarded.txt | awk '{print $1, $2}') <(grep -E "fwd|dgrad" $O/kbench_unguarded.txt | awk '{print $2}')
for i in 1 2; do
DDL_EXPERIMENTAL=1 DDL_FORCE_GUARD=1 timeout -k 10 200 python bench.py --steps 40 --warmup 8 > $O/guard_$i.json 2> $O/guard_$i.err || { tail $O/guard_$i.err; exit 1; }
timeout -k 10 200 python bench.py --steps 40 --warmup 8 > $O/noguard_$i.json 2> $O/noguard_$i.err || { tail $O/noguard_$i.err; exit 1; }
python -c "import json; a=json.load(open('$O/guard_$i.json')); b=json.load(open('$O/noguard_$i.json')); print('guarded %.3f ms   unguarded %.3f ms' % (a['ms_per_step'], b['ms_per_step']))"
done
timeout -k 10 600 python -u -m pytest tests/test_gpu_kernels_real.py tests/test_gpu_native.py -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest_kernels.txt 2>&1 || { tail -40 $O/pytest_kernels.txt; exit 1; }
tail -2 $O/pytest_kernels.txt
