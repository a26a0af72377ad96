"""Classification metrics from one confusion matrix + the reference's CSV sink.

The reference computes its validation metrics with scikit-learn on CPU numpy
after a D2H copy of all logits (reference: ddp.py:259-290, single.py:226-233).
Here a single k x k confusion matrix (built on whatever device the predictions
live on - one ``bincount``) yields every metric, so only 25 integers cross the
PCIe bus.  Values match ``sklearn.metrics`` (``zero_division`` -> 0 semantics)
and ``cohen_kappa_score(weights="quadratic")``; tests/test_metrics.py pins that.

CSV schema (reference: ddp.py:299-308): one file per metric per job,
``{log_dir}/by_job_id/{job_id}/{metric}.csv``, header-less rows
``[timestamp, job_id, global_rank, local_rank, model_start_job_id, epoch, value]``.
"""
from __future__ import annotations

import csv
import os
from datetime import datetime
from pathlib import Path
from typing import Dict, Optional

import numpy as np
import torch

EVAL_METRICS = (
    "val_accuracy", "macro_f1", "weighted_f1", "macro_precision",
    "weighted_precision", "macro_recall", "weighted_recall", "qwk",
)


def confusion_matrix(targets: torch.Tensor, preds: torch.Tensor, num_classes: int) -> torch.Tensor:
    """cm[t, p] = count, int64, computed on the tensors' device."""
    t = targets.reshape(-1).to(torch.int64)
    p = preds.reshape(-1).to(torch.int64)
    idx = t * num_classes + p
    return torch.bincount(idx, minlength=num_classes * num_classes).reshape(num_classes, num_classes)


def _safe_div(num: np.ndarray, den: np.ndarray) -> np.ndarray:
    out = np.zeros_like(num, dtype=np.float64)
    np.divide(num, den, out=out, where=den != 0)
    return out


def metrics_from_confusion(cm) -> Dict[str, float]:
    """All 8 reference eval metrics from a confusion matrix (rows = truth).

    sklearn semantics: labels = classes present in y_true or y_pred; precision /
    recall / F1 with a zero denominator count as 0; "weighted" weights by support.
    """
    cm = np.asarray(cm.cpu() if isinstance(cm, torch.Tensor) else cm, dtype=np.float64)
    n = cm.sum()
    tp = np.diag(cm)
    support = cm.sum(axis=1)       # true count per class
    predicted = cm.sum(axis=0)     # predicted count per class
    present = (support + predicted) > 0
    precision = _safe_div(tp, predicted)
    recall = _safe_div(tp, support)
    f1 = _safe_div(2 * precision * recall, precision + recall)

    def macro(v):
        return float(v[present].mean()) if present.any() else 0.0

    def weighted(v):
        s = support[present]
        return float((v[present] * s).sum() / s.sum()) if s.sum() > 0 else 0.0

    # quadratic weighted kappa, sklearn.metrics.cohen_kappa_score(weights="quadratic"):
    # computed over the present labels only (sklearn builds the matrix from unique labels).
    idx = np.nonzero(present)[0]
    sub = cm[np.ix_(idx, idx)]
    k = len(idx)
    if k <= 1 or n == 0:
        qwk = float("nan")
    else:
        expected = np.outer(sub.sum(axis=1), sub.sum(axis=0)) / sub.sum()
        w = (np.arange(k)[:, None] - np.arange(k)[None, :]) ** 2.0
        den = (w * expected).sum()
        qwk = float(1.0 - (w * sub).sum() / den) if den != 0 else float("nan")

    return {
        "val_accuracy": float(tp.sum() / n) if n else 0.0,
        "macro_f1": macro(f1),
        "weighted_f1": weighted(f1),
        "macro_precision": macro(precision),
        "weighted_precision": weighted(precision),
        "macro_recall": macro(recall),
        "weighted_recall": weighted(recall),
        "qwk": qwk,
    }


def accuracy(targets: torch.Tensor, preds: torch.Tensor) -> float:
    if targets.numel() == 0:
        return 0.0
    return float((targets.reshape(-1) == preds.reshape(-1)).double().mean().item())


class CsvMetricLogger:
    """Append-only per-metric CSV files in the reference's schema."""

    def __init__(self, log_dir: str, job_id: str, global_rank: int = 0, local_rank: int = 0,
                 model_start_job_id: Optional[str] = None, enabled: bool = True) -> None:
        self.root = Path(log_dir)
        self.job_id = job_id
        self.global_rank = global_rank
        self.local_rank = local_rank
        self.model_start_job_id = model_start_job_id or job_id
        self.enabled = enabled

    @property
    def job_dir(self) -> Path:
        return self.root / "by_job_id" / self.job_id

    def log(self, metric: str, value, epoch: int) -> None:
        if not self.enabled:
            return
        os.makedirs(self.job_dir, exist_ok=True)
        with open(self.job_dir / f"{metric}.csv", "a", newline="") as f:
            now = datetime.now().strftime("%Y-%m-%d %H:%M:%S")
            csv.writer(f).writerow([now, self.job_id, self.global_rank, self.local_rank,
                                    self.model_start_job_id, epoch, value])

    def log_row(self, filename: str, row) -> None:
        """Free-form row into ``{log_dir}/{filename}`` (communication_time.csv, gradient.csv)."""
        if not self.enabled:
            return
        os.makedirs(self.root, exist_ok=True)
        with open(self.root / filename, "a", newline="") as f:
            csv.writer(f).writerow(row)


def gradient_stats(named_params) -> Dict[str, Dict[str, float]]:
    """Per-parameter |grad| statistics - the reference's opt-in debug dump
    (reference: ddp.py:310-326: min, mean, max, p25, median, p75, std)."""
    out = {}
    for name, p in named_params:
        if p.grad is None:
            continue
        g = p.grad.detach().abs().float().reshape(-1)
        if g.numel() == 0:
            continue
        q = torch.quantile(g[: 1 << 24], torch.tensor([0.25, 0.5, 0.75], device=g.device))
        out[name] = dict(min=g.min().item(), mean=g.mean().item(), max=g.max().item(),
                         p25=q[0].item(), median=q[1].item(), p75=q[2].item(),
                         std=g.std().item() if g.numel() > 1 else 0.0)
    return out
