"""Dump the captured training-step hipGraph as DOT (hipGraphDebugDotPrint) and summarise its topology."""
import collections
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ddl import comm
from ddl.config import TrainConfig
from ddl.trainer import SingleStrategy
from ddl.utils import graph as G

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/graph.dot"
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
B = 30
# enable debug mode on every CUDAGraph the step creates
_orig = torch.cuda.CUDAGraph.__new__


class DebugGraph(torch.cuda.CUDAGraph):
    def __new__(cls, *a, **k):
        g = super().__new__(cls, *a, **k)
        g.enable_debug_mode()
        return g


G.torch.cuda.CUDAGraph = DebugGraph
strat = SingleStrategy(TrainConfig(strategy="single", batch_size=B), comm.CommContext(0, 1, 0, dev, "none"))
x = torch.randint(0, 256, (B, 3, 224, 224), dtype=torch.uint8, device=dev)
y = torch.randint(0, 5, (B,), device=dev)
strat.model.train()
for _ in range(5):
    strat.train_step(x, y, B)
torch.cuda.synchronize()
cap = next(iter(strat._step._graphs.values()))
cap.graph.debug_dump(out)
txt = open(out).read()
nodes = re.findall(r'^\s*"?(\w+)"?\s*\[.*?label="([^"]*)"', txt, re.M)
edges = re.findall(r'^\s*"?(\w+)"?\s*->\s*"?(\w+)"?', txt, re.M)
print("nodes", len(nodes), "edges", len(edges))
indeg, outdeg = collections.Counter(), collections.Counter()
for a, b in edges:
    outdeg[a] += 1
    indeg[b] += 1
ids = [n for n, _ in nodes]
print("roots", sum(1 for n in ids if indeg[n] == 0), "leaves", sum(1 for n in ids if outdeg[n] == 0),
      "forks (out>1)", sum(1 for n in ids if outdeg[n] > 1), "joins (in>1)", sum(1 for n in ids if indeg[n] > 1))
kinds = collections.Counter(re.sub(r"[^A-Za-z_].*", "", lab.split("\\n")[0])[:40] for _, lab in nodes)
print(kinds.most_common(12))
