#!/usr/bin/env python3
"""Host and device timeline of ONE pipeline step with micro-batch lanes (one stage, the reference's 5 x 6 shape): for
every graph replay the host time at which it was issued / returned and the device time at which it started / ended
(events on the replaying stream).  Answers "is the host blocked?" and "which graphs overlap on the device?"."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lanes", type=int, default=2)
ap.add_argument("--fwd-lanes", type=int, default=1)
ap.add_argument("--schedule", default="1f1b")
ap.add_argument("--batch", type=int, default=30)
ap.add_argument("--microbatches", type=int, default=5)
a = ap.parse_args()

from ddl import comm  # noqa: E402
from ddl.config import TrainConfig  # noqa: E402
from ddl.trainer import build_strategy  # noqa: E402

dev = torch.device("cuda", 0)
ctx = comm.CommContext(0, 1, 0, dev, "none")
cfg = TrainConfig(strategy="pp", batch_size=a.batch, num_microbatches=a.microbatches, pp_lanes=a.lanes, pp_fwd_lanes=a.fwd_lanes, split_points=[],
                  schedule=a.schedule)
strat = build_strategy(cfg, ctx)
strat.model.train()
x = torch.randint(0, 256, (a.batch, 3, 224, 224), dtype=torch.uint8, device=dev)
y = torch.randint(0, 5, (a.batch,), device=dev)
for _ in range(6):
    strat.train_step(x, y, a.batch)
torch.cuda.synchronize()

log = []
orig = torch.cuda.CUDAGraph.replay


def replay(self):
    s = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(s)
    orig(self)
    e1.record(s)
    log.append((t0, time.perf_counter(), e0, e1, s.stream_id))


torch.cuda.CUDAGraph.replay = replay
base = torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
h0 = time.perf_counter()
base.record(torch.cuda.current_stream())
strat.train_step(x, y, a.batch)
h1 = time.perf_counter()
torch.cuda.synchronize()
h2 = time.perf_counter()
torch.cuda.CUDAGraph.replay = orig
print(f"lanes {a.lanes} {a.schedule}: host issue {1e3 * (h1 - h0):.2f} ms, step wall {1e3 * (h2 - h0):.2f} ms; order {strat.schedule.order}")
print("  #  stream        host issue -> return (ms)     device start -> end (ms)")
ids = {}
for k, (t0, t1, e0, e1, sid) in enumerate(log):
    ids.setdefault(sid, len(ids))
    print(f" {k:2d}  s{ids[sid]}  {1e3 * (t0 - h0):8.2f} -> {1e3 * (t1 - h0):6.2f}      {base.elapsed_time(e0):8.2f} -> {base.elapsed_time(e1):6.2f}"
          f"   ({strat.schedule.order[k] if k < len(strat.schedule.order) else ''})")
