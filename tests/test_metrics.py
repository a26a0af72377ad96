import csv

import numpy as np
import pytest
import torch

from ddl import metrics


def _sk(y, p):
    from sklearn.metrics import (accuracy_score, cohen_kappa_score, f1_score, precision_score,
                                 recall_score)
    kw = dict(zero_division=0)
    return {
        "val_accuracy": accuracy_score(y, p),
        "macro_f1": f1_score(y, p, average="macro", **kw),
        "weighted_f1": f1_score(y, p, average="weighted", **kw),
        "macro_precision": precision_score(y, p, average="macro", **kw),
        "weighted_precision": precision_score(y, p, average="weighted", **kw),
        "macro_recall": recall_score(y, p, average="macro", **kw),
        "weighted_recall": recall_score(y, p, average="weighted", **kw),
        "qwk": cohen_kappa_score(y, p, weights="quadratic"),
    }


@pytest.mark.parametrize("seed,classes_used", [(0, 5), (1, 5), (2, 3), (3, 4)])
def test_metrics_match_sklearn(seed, classes_used):
    """The reference computes these with sklearn (ddp.py:265-272); one confusion matrix must
    reproduce them, including when some classes never occur."""
    rng = np.random.default_rng(seed)
    y = rng.integers(0, classes_used, 360)
    p = np.where(rng.random(360) < 0.6, y, rng.integers(0, classes_used, 360))
    cm = metrics.confusion_matrix(torch.tensor(y), torch.tensor(p), 5)
    got = metrics.metrics_from_confusion(cm)
    want = _sk(y, p)
    for k, v in want.items():
        assert got[k] == pytest.approx(v, abs=1e-12), k


def test_predicted_only_class_counts_in_macro():
    y = np.array([0, 0, 1, 1, 1, 0])
    p = np.array([0, 2, 1, 1, 0, 0])      # class 2 predicted, never true
    got = metrics.metrics_from_confusion(metrics.confusion_matrix(torch.tensor(y), torch.tensor(p), 5))
    want = _sk(y, p)
    for k, v in want.items():
        assert got[k] == pytest.approx(v, abs=1e-12), k


def test_csv_schema(tmp_path):
    """Row = [timestamp, job_id, global_rank, local_rank, model_start_job_id, epoch, value], no
    header, one file per metric under by_job_id/{job} (ddp.py:299-308)."""
    lg = metrics.CsvMetricLogger(str(tmp_path), "ddp-abc", 3, 1, model_start_job_id="ddp-old")
    lg.log("loss", 0.5, 7)
    lg.log("loss", 0.25, 8)
    rows = list(csv.reader(open(tmp_path / "by_job_id" / "ddp-abc" / "loss.csv")))
    assert len(rows) == 2 and len(rows[0]) == 7
    assert rows[0][1:6] == ["ddp-abc", "3", "1", "ddp-old", "7"] and float(rows[1][6]) == 0.25
    off = metrics.CsvMetricLogger(str(tmp_path), "x", enabled=False)
    off.log("loss", 1, 0)
    assert not (tmp_path / "by_job_id" / "x").exists()


def test_gradient_stats():
    lin = torch.nn.Linear(4, 3)
    lin(torch.ones(2, 4)).sum().backward()
    st = metrics.gradient_stats(lin.named_parameters())
    assert set(st) == {"weight", "bias"} and st["bias"]["max"] == pytest.approx(2.0)
