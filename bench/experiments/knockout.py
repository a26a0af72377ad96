#!/usr/bin/env python3
"""Knock-out timing experiments: leave a kernel CLASS out of the training step and time what is left.

RESULTS ARE WRONG BY DESIGN - only the step time is of interest (README "What the batch-30 step is made of").
This used to be an environment switch inside the engine (``DDL_KNOCKOUT`` in models/native_densenet.py); it now
lives here, outside the package, as a proxy around the extension module that the engine calls - the production
module has no path that drops a gradient kernel, and bench.py / the trainers refuse to start with
``DDL_KNOCKOUT`` set.

    python bench/experiments/knockout.py --knock wgrad          # no dense-block weight gradients
    python bench/experiments/knockout.py --knock bnprep         # no bn_bwd_prep launches (ungrouped tail only)
    python bench/experiments/knockout.py --knock none           # the control on the same box
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch  # noqa: E402


class _NoQueue:
    def flush(self):
        pass


class _Proxy:
    """The extension module with one kernel class replaced by no-ops."""

    def __init__(self, ext, knock: str) -> None:
        self._ext, self._knock = ext, knock

    def __getattr__(self, name):
        k = self._knock
        if k == "wgrad" and name in ("conv_wgrad_group", "conv_wgrad"):
            return lambda *a, **kw: None
        if k == "bnprep" and name == "bn_bwd_prep":
            return lambda *a, **kw: None
        return getattr(self._ext, name)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--knock", default="none", choices=["none", "wgrad", "bnprep"])
    ap.add_argument("--batch", type=int, default=30)
    ap.add_argument("--steps", type=int, default=30)
    a = ap.parse_args()
    from ddl import comm, ops
    from ddl.config import TrainConfig
    from ddl.models import native_densenet
    from ddl.trainer import build_strategy
    if a.knock != "none":
        proxy = _Proxy(ops.require_ext(), a.knock)
        native_densenet._ext = lambda: proxy
    dev = torch.device("cuda", 0)
    ctx = comm.CommContext(0, 1, 0, dev, "none")
    strat = build_strategy(TrainConfig(strategy="single", batch_size=a.batch), ctx)
    strat.model.train()
    x = torch.randint(0, 256, (a.batch, 3, 224, 224), dtype=torch.uint8, device=dev)
    y = torch.randint(0, 5, (a.batch,), device=dev)
    for _ in range(6):
        strat.train_step(x, y, a.batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        strat.train_step(x, y, a.batch)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    print(json.dumps({"experiment": "knockout", "knock": a.knock, "batch": a.batch, "ms_per_step": round(ms, 3),
                      "results_valid": a.knock == "none"}))


if __name__ == "__main__":
    main()
