#!/usr/bin/env python3
"""How much HOST time does one replay of the captured training step cost?

bench.py's host_issue_ms_per_step is the wall time the host spends issuing back-to-back steps; when the queue to
the GPU is full that includes waiting for the GPU (back-pressure), so it is an upper bound.  This probe launches ONE
step graph into an idle GPU (device synchronised first) and reports the wall and CPU time of the launch call alone,
plus the node count of the graph, at the benchmark batch and at a tiny batch (where the GPU is faster than the host)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ddl import comm
from ddl.config import TrainConfig
from ddl.trainer import SingleStrategy

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
for B in (30, 2):
    strat = SingleStrategy(TrainConfig(strategy="single", batch_size=B), comm.CommContext(0, 1, 0, dev, "none"))
    x = torch.randint(0, 256, (B, 3, 224, 224), dtype=torch.uint8, device=dev)
    y = torch.randint(0, 5, (B,), device=dev)
    strat.model.train()
    for _ in range(6):
        strat.train_step(x, y, B)
    torch.cuda.synchronize()
    cap = next(iter(strat._step._graphs.values()))
    wall, cpu, gpu = [], [], []
    for _ in range(12):
        torch.cuda.synchronize()
        t0, c0 = time.perf_counter(), time.thread_time()
        cap.graph.replay()
        t1, c1 = time.perf_counter(), time.thread_time()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        wall.append((t1 - t0) * 1e3), cpu.append((c1 - c0) * 1e3), gpu.append((t2 - t0) * 1e3)
    # back-to-back: 20 launches without a sync in between
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        cap.graph.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    med = lambda v: sorted(v)[len(v) // 2]
    print(f"B={B}: one launch into an idle GPU: host wall {med(wall):.3f} ms, host cpu {med(cpu):.3f} ms, "
          f"launch+drain {med(gpu):.3f} ms | 20 back-to-back: issue {(t1 - t0) / 20 * 1e3:.3f} ms/step, "
          f"total {(t2 - t0) / 20 * 1e3:.3f} ms/step", flush=True)
