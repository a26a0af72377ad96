#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
export PMC_CASES="b3_wgrad1x1_C640 b3_wgrad3x3 b4_wgrad1x1_C768 b3_dgrad1x1_N640 b3_dgrad3x3 b3_fwd3x3 b2_wgrad1x1_C320"
bash bench/gpu_pmc.sh 2>&1 | tail -30
cd $R && python bench/pmc_summary.py gpurun_out/pmc2 > gpurun_out/pmc2/summary.txt; cat gpurun_out/pmc2/summary.txt
