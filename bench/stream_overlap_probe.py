#!/usr/bin/env python3
"""Do two HIP streams of this process really run side by side?

(1) pairs of streams from torch's pool, a chain of small kernels on each, eager launches: wall time of both chains
    issued together against one chain alone (1.0 = perfectly concurrent, 2.0 = serialised);
(2) the same with each chain captured into a hipGraph and replayed on its stream;
(3) graph chains of the training engine's shape: replay order A B A B on two streams.
Motivation: the micro-batch lanes of the pipeline stage (parallel/pp.py) showed NO overlap between the forward stream
and the backward streams in a kernel trace although they sat on different hardware queues."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
N = 200


def chain(x):
    for _ in range(N):
        x = x * 1.0001 + 0.5            # one small elementwise kernel per op (2 launches)
    return x


def wall(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, host * 1e3


xs = [torch.randn(4096, device=dev) for _ in range(8)]
streams = [torch.cuda.Stream(dev) for _ in range(8)]


def eager_on(idx):
    def run():
        for i in idx:
            with torch.cuda.stream(streams[i]):
                chain(xs[i])
    return run


for _ in range(3):
    wall(eager_on([0]))
one, h1 = wall(eager_on([0]))
print(f"eager chain alone: {one:.2f} ms (host {h1:.2f})")
for j in range(1, 8):
    both, h = wall(eager_on([0, j]))
    print(f"eager  streams 0+{j}: {both:.2f} ms  ratio {both / one:.2f}  (host {h:.2f})")

graphs = []
for i in range(8):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = chain(xs[i])
    graphs.append((g, y))


def graph_on(idx, order=None):
    def run():
        for i in (order or idx):
            with torch.cuda.stream(streams[i]):
                graphs[i][0].replay()
    return run


for _ in range(3):
    wall(graph_on([0]))
one, h1 = wall(graph_on([0]))
print(f"graph chain alone: {one:.2f} ms (host {h1:.2f})")
for j in range(1, 8):
    both, h = wall(graph_on([0, j]))
    print(f"graph  streams 0+{j}: {both:.2f} ms  ratio {both / one:.2f}  (host {h:.2f})")
quad, h = wall(graph_on([0, 1, 2, 3]))
print(f"graph  streams 0+1+2+3: {quad:.2f} ms  ratio {quad / one:.2f}  (host {h:.2f})")
# same two graphs replayed alternately several times (A B A B ...): the lanes' pattern
alt, h = wall(graph_on(None, order=[0, 1, 0, 1, 0, 1]))
print(f"graph  A B A B A B on streams 0/1: {alt:.2f} ms  ratio vs 3 chains {alt / (3 * one):.2f}  (host {h:.2f})")
# graphs launched from the SAME stream object vs the default stream
with torch.cuda.stream(streams[1]):
    pass
both, h = wall(lambda: (graphs[0][0].replay(), graph_on([1])()))
print(f"graph  default stream + stream 1: {both:.2f} ms  ratio {both / one:.2f}")
