"""Results aggregation over the CSV logs (reference: ipynb/main.ipynb, the L5 "analysis" layer).

The reference's only analysis artefact is a notebook that reads the header-less CSV files the
trainers append to and prints three tables:

  1. mean epoch time per job                      (ipynb/main.ipynb cells 2-3)
  2. epoch-29 quality metrics, mean per strategy  (cells 4-5)
  3. P2P ping statistics, first iteration excluded (cells 6-9)

The same three tables as functions (so ``ipynb/main.ipynb`` in this repo is five lines per
cell and the numbers are testable), plus what the reference never reported and BASELINE.json
asks for: samples/s and the compute / RCCL split per step, and the message-size sweep table.

    python -m ddl.analysis --log-dir $DDL_LOG_DIR [--epoch 29] [--samples-per-epoch 2400]
"""
from __future__ import annotations

import argparse
import glob
import os
from functools import reduce
from typing import Dict, Iterable, List, Optional, Sequence

import pandas as pd

METRIC_COLUMNS = ["recorded_at", "job_id", "global_rank", "local_rank", "model_job_id", "epoch"]
QUALITY_METRICS = ["loss", "train_accuracy", "val_loss", "val_accuracy", "weighted_f1"]
SWEEP_COLUMNS = ["job_id", "op", "bytes", "ms_median", "ms_min", "gbps", "busbw_gbps", "world"]


def strategy_of(job_id: str) -> str:
    """Job ids are ``{script stem}-{hash}`` (torchx naming: ``ddp-mdgx1trqq0h4gd``); ``local`` runs
    carry no prefix."""
    return job_id.rsplit("-", 1)[0] if "-" in job_id else job_id


def jobs_in(log_dir: str) -> List[str]:
    root = os.path.join(log_dir, "by_job_id")
    return sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d))) if os.path.isdir(root) else []


def load_metric(log_dir: str, metric: str, jobs: Optional[Iterable[str]] = None) -> pd.DataFrame:
    """All rows of ``{log_dir}/by_job_id/{job}/{metric}.csv`` (schema: ddp.py:306-308) for ``jobs``
    (default: every job found).  A top-level ``{log_dir}/{metric}.csv`` - the form the reference
    notebook reads ``epoch_time.csv`` in - is included when present."""
    frames = []
    names = METRIC_COLUMNS + [metric]
    wanted = set(jobs) if jobs is not None else None
    for job in jobs_in(log_dir):
        if wanted is not None and job not in wanted:
            continue
        path = os.path.join(log_dir, "by_job_id", job, f"{metric}.csv")
        if os.path.exists(path):
            frames.append(pd.read_csv(path, names=names))
    top = os.path.join(log_dir, f"{metric}.csv")
    if os.path.exists(top):
        df = pd.read_csv(top, names=names)
        frames.append(df[df["job_id"].isin(wanted)] if wanted is not None else df)
    if not frames:
        return pd.DataFrame(columns=names)
    out = pd.concat(frames, ignore_index=True).drop_duplicates()
    out["strategy"] = out["job_id"].map(strategy_of)
    return out


def epoch_time_by_job(log_dir: str, jobs: Optional[Iterable[str]] = None,
                      samples_per_epoch: Optional[int] = None) -> pd.DataFrame:
    """Table 1: mean seconds per training epoch per job (+ samples/s when the epoch size is known;
    the reference's 2400 is derived in SURVEY 6.2)."""
    df = load_metric(log_dir, "epoch_time", jobs)
    out = df.groupby("job_id")["epoch_time"].agg(["mean", "min", "count"]).rename(
        columns={"mean": "time", "min": "best", "count": "epochs"}).reset_index()
    out["strategy"] = out["job_id"].map(strategy_of)
    if samples_per_epoch:
        out["samples_per_s"] = samples_per_epoch / out["time"]
    return out


def final_metrics_by_strategy(log_dir: str, epoch: int = 29,
                              jobs_by_strategy: Optional[Dict[str, Sequence[str]]] = None,
                              metrics: Sequence[str] = QUALITY_METRICS) -> pd.DataFrame:
    """Table 2: mean of each quality metric at ``epoch`` over the jobs of each strategy."""
    per_metric = []
    for metric in metrics:
        if jobs_by_strategy is not None:
            frames = []
            for strat, jobs in jobs_by_strategy.items():
                f = load_metric(log_dir, metric, jobs)
                f = f[f["epoch"] == epoch].copy()
                f["strategy"] = strat
                frames.append(f)
            df = pd.concat(frames, ignore_index=True) if frames else pd.DataFrame(columns=["strategy", metric])
        else:
            df = load_metric(log_dir, metric)
            df = df[df["epoch"] == epoch]
        per_metric.append(df.groupby("strategy")[metric].mean().reset_index())
    return reduce(lambda a, b: pd.merge(a, b, on="strategy", how="inner"), per_metric)


def communication_summary(log_dir: str, jobs: Optional[Iterable[str]] = None) -> pd.DataFrame:
    """Table 3: per job, mean / median / p99 ping (ms) over iterations != 0 and the first
    iteration (communicator set-up) separately (communication_time.py:56-59 rows ``job,i,ms``)."""
    rows = []
    for path in [os.path.join(log_dir, "communication_time.csv")] + \
            sorted(glob.glob(os.path.join(log_dir, "by_job_id", "*", "communication_time.csv"))):
        if os.path.exists(path):
            rows.append(pd.read_csv(path, names=["job_id", "iteration", "elapsed_time"]))
    if not rows:
        return pd.DataFrame(columns=["job_id", "elapsed_time", "median", "p99", "first_iteration", "gb_per_s"])
    df = pd.concat(rows, ignore_index=True)
    if jobs is not None:
        df = df[df["job_id"].isin(set(jobs))]
    steady = df[df["iteration"] != 0].groupby("job_id")["elapsed_time"]
    out = steady.mean().reset_index()
    out["median"] = steady.median().values
    out["p99"] = steady.quantile(0.99).values
    first = df[df["iteration"] == 0].groupby("job_id")["elapsed_time"].first()
    out["first_iteration"] = out["job_id"].map(first)
    out["gb_per_s"] = 4 * 2 ** 20 / (out["elapsed_time"] * 1e-3) / 1e9        # 4 MiB payload one way
    return out


def sweep_table(log_dir: str) -> pd.DataFrame:
    """Message-size sweep rows written by ``communication_time.py --sweep``."""
    paths = [os.path.join(log_dir, "communication_sweep.csv")] + \
        sorted(glob.glob(os.path.join(log_dir, "by_job_id", "*", "communication_sweep.csv")))
    frames = [pd.read_csv(p, names=SWEEP_COLUMNS) for p in paths if os.path.exists(p)]
    return pd.concat(frames, ignore_index=True) if frames else pd.DataFrame(columns=SWEEP_COLUMNS)


def speedup_vs_single(epoch_times: pd.DataFrame) -> pd.DataFrame:
    """Best mean epoch time per strategy and its speed-up over the best single-device job
    (the reference's headline: ddp 1.48x, pp 0.82x, ddpnpp 0.67x - BASELINE.md section 2)."""
    best = epoch_times.groupby("strategy")["time"].min().reset_index()
    single = best.loc[best["strategy"] == "single", "time"]
    if len(single):
        best["speedup_vs_single"] = float(single.iloc[0]) / best["time"]
    return best


def main(argv=None) -> None:
    p = argparse.ArgumentParser(prog="ddl.analysis")
    p.add_argument("--log-dir", default=os.environ.get("DDL_LOG_DIR", "/mnt/dcornelius/training_logs"))
    p.add_argument("--epoch", type=int, default=29)
    p.add_argument("--samples-per-epoch", type=int, default=2400)
    a = p.parse_args(argv)
    pd.set_option("display.width", 160)
    et = epoch_time_by_job(a.log_dir, samples_per_epoch=a.samples_per_epoch)
    print("# Training time (mean s/epoch per job)\n", et.to_string(index=False))
    if len(et):
        print("\n# Best per strategy\n", speedup_vs_single(et).to_string(index=False))
    try:
        print(f"\n# Quality at epoch {a.epoch} (mean per strategy)\n",
              final_metrics_by_strategy(a.log_dir, a.epoch).to_string(index=False))
    except (KeyError, TypeError):
        print(f"\n# no quality metrics logged at epoch {a.epoch}")
    cs = communication_summary(a.log_dir)
    if len(cs):
        print("\n# Communication time (ms, first iteration excluded)\n", cs.to_string(index=False))
    sw = sweep_table(a.log_dir)
    if len(sw):
        print("\n# Message-size sweep\n", sw.to_string(index=False))


if __name__ == "__main__":
    main()
