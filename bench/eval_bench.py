#!/usr/bin/env python3
"""Inference (evaluation-loop) throughput and latency of the native DenseNet-121 engine on one GPU.

The reference evaluates after every epoch with the training model in eval mode (single.py:157-180: forward,
argmax, gather to rank 0); this measures that forward - uint8 batch in, logits out - as the Trainer runs it:
`GraphedEvalStep`, one hipGraph per batch shape.  One JSON line per (dtype, batch):

    python bench/eval_bench.py [--batches 1 8 30 128 512] [--dtypes fp32 bf16] [--steps 50]

Not the headline metric (bench.py measures that); the numbers land in profiles/ as serving-side evidence.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, nargs="+", default=[1, 8, 30, 128, 512])
    ap.add_argument("--dtypes", nargs="+", default=["fp32", "bf16"], choices=["fp32", "bf16"])
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--no-graph", action="store_true")
    a = ap.parse_args()

    from ddl.models import create_stage
    from ddl.models.densenet import UNIT_NAMES
    from ddl.utils.graph import GraphedEvalStep

    dev = torch.device("cuda", 0)
    for dtype in a.dtypes:
        model = create_stage(UNIT_NAMES, dev, 5, dtype, pretrained=False)
        model.eval()
        step = GraphedEvalStep(model, enabled=not a.no_graph)
        for B in a.batches:
            g = torch.Generator().manual_seed(B)
            xs = [torch.randint(0, 256, (B, 3, 224, 224), dtype=torch.uint8, generator=g).to(dev) for _ in range(2)]
            for i in range(4):                       # eager pass, capture, two replays
                step(xs[i % 2])
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(a.steps):
                out = step(xs[i % 2])
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t0) / a.steps
            # latency of ONE batch into an idle GPU (what a serving request sees), median of 9
            lat = []
            for i in range(9):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                out = step(xs[i % 2])
                torch.cuda.synchronize(dev)
                lat.append(time.perf_counter() - t0)
            assert torch.isfinite(out.float()).all()
            print(json.dumps({"metric": "DenseNet-121 inference forward (eval loop)", "dtype": dtype, "batch": B,
                              "ms_per_batch": round(dt * 1e3, 3), "images_per_s": round(B / dt, 1),
                              "latency_ms_idle_gpu": round(sorted(lat)[4] * 1e3, 3),
                              "launch": "eager" if a.no_graph else "hipGraph", "impl": type(model).__name__}),
                  flush=True)
        del step, model
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
