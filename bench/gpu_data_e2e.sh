#!/bin/bash
# end-to-end input pipeline check on a real PNG dataset (APTOS layout and size: 2400 train / 360 eval, synthetic
# content): torch DataLoader + PIL workers (decode every epoch, like the reference), the C++ thread-pool loader
# (csrc/png_loader.cpp, decode every epoch), and the HBM-resident dataset cache filled by that pool (decode once).
# Epoch time = training loop only, as the reference logs it (single.py:172-176).
set -o pipefail
D=/tmp/aptos_png
mkdir -p gpurun_out
python - <<PY || exit 1
import sys; sys.path.insert(0, ".")
from ddl.data import make_png_dataset
make_png_dataset("$D", 2400, 360, image_size=224)
print("dataset written")
PY
run() {   # name, extra flags
  local name=$1; shift
  local L=/tmp/logs_$name
  DDL_DATA_DIR=$D DDL_LOG_DIR=$L DDL_CKPT_DIR=/tmp/ckpt TORCHX_JOB_ID=job-$name timeout -k 10 400 \
    python single.py --epochs 4 --num-workers 8 "$@" > gpurun_out/data_e2e_$name.log 2>&1 || { tail -30 gpurun_out/data_e2e_$name.log; exit 1; }
  echo "== $name ($*): epoch_time.csv"; cat $L/by_job_id/job-$name/epoch_time.csv
  echo "== val_accuracy"; tail -1 $L/by_job_id/job-$name/val_accuracy.csv 2>/dev/null
}
run dataloader --cache-dataset off --native-loader off &&
run native --cache-dataset off &&
run resident --cache-dataset auto
