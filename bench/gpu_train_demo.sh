#!/bin/bash
# End-to-end demo on one MI355X: single.py on an APTOS-layout PNG dataset of the reference's size (2400 train /
# 360 eval images, 224x224, synthetic content), 4 epochs + snapshot, then a DCP resume for one more epoch, then
# the notebook's tables.  Output: gpurun_out/train_demo/.
set -o pipefail
O=gpurun_out/train_demo; mkdir -p $O/csv
D=/tmp/aptos_png_full
python - <<PY || exit 1
import sys, time; sys.path.insert(0, ".")
from ddl.data import make_png_dataset
t = time.time(); make_png_dataset("$D", 2400, 360, image_size=224); print("dataset written in %.1f s" % (time.time() - t))
PY
export DDL_DATA_DIR=$D DDL_LOG_DIR=/tmp/demo_logs DDL_CKPT_DIR=/tmp/demo_ckpt TORCHX_JOB_ID=single-mi355xdemo
timeout -k 10 600 python single.py --epochs 4 --save-every 4 --num-workers 8 > $O/single_demo.log 2>&1 || { tail -30 $O/single_demo.log; exit 1; }
TORCHX_JOB_ID=single-mi355xresume timeout -k 10 600 python single.py --epochs 5 --num-workers 8 \
  --snapshot-job-id single-mi355xdemo --snapshot-epoch 3 > $O/single_resume.log 2>&1 || { tail -30 $O/single_resume.log; exit 1; }
cp /tmp/demo_logs/by_job_id/single-mi355xdemo/*.csv $O/csv/
python -m ddl.analysis --log-dir /tmp/demo_logs > $O/analysis_demo.txt 2>&1 || tail -5 $O/analysis_demo.txt
echo "== epoch_time.csv"; cat $O/csv/epoch_time.csv; echo "== val_accuracy.csv"; cat $O/csv/val_accuracy.csv
grep -E "Resum|resum|epoch 4|Epoch 4 \| Loss" $O/single_resume.log | head -5
