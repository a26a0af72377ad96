#!/bin/bash
# end-to-end input pipeline check on a real PNG dataset (APTOS layout, synthetic content):
# host DataLoader (decode every epoch, like the reference) vs the HBM-resident dataset cache
set -o pipefail
D=/tmp/aptos_png
python - <<PY || exit 1
import sys; sys.path.insert(0, ".")
from ddl.data import make_png_dataset
make_png_dataset("$D", 1200, 240, image_size=224)
print("dataset written")
PY
for mode in off auto; do
  L=/tmp/logs_$mode
  DDL_DATA_DIR=$D DDL_LOG_DIR=$L DDL_CKPT_DIR=/tmp/ckpt TORCHX_JOB_ID=job-$mode timeout -k 10 400 \
    python single.py --epochs 4 --num-workers 8 --cache-dataset $mode > gpurun_out/data_e2e_$mode.log 2>&1 || { tail -30 gpurun_out/data_e2e_$mode.log; exit 1; }
  echo "== cache-dataset $mode: epoch_time.csv"; cat $L/by_job_id/job-$mode/epoch_time.csv
  echo "== val_accuracy"; cat $L/by_job_id/job-$mode/val_accuracy.csv 2>/dev/null | tail -2
done
