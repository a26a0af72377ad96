"""Device-side cost of one dependent kernel node: chains of N trivial kernels, eager and as a hipGraph."""
import sys
import time

import torch

sys.path.insert(0, ".")
from ddl import ops  # noqa: E402,F401  (loads the extension: same process set-up as the engine)

dev = torch.device("cuda", 0)
x = torch.zeros(256, device=dev)
big = torch.zeros(64 * 2 ** 20, device=dev)       # 256 MB: a kernel that really touches memory


def chain(n, t):
    for _ in range(n):
        t.add_(1.0)


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


N = 1000
for name, t in (("256 floats", x), ("4 MB", big[: 2 ** 20])):
    eager = timeit(lambda: chain(N, t))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chain(N, t)
    graph = timeit(g.replay)
    print(f"add_ on {name}: eager {eager / N * 1e3:.2f} us/kernel, hipGraph {graph / N * 1e3:.2f} us/node", flush=True)

# two independent chains on two streams inside one graph (the step's compute + weight-gradient chains)
s2 = torch.cuda.Stream()
y = torch.zeros(256, device=dev)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cur = torch.cuda.current_stream()
    s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        chain(N, y)
    chain(N, x)
    cur.wait_stream(s2)
t = timeit(g.replay)
print(f"two parallel chains of {N}: {t / N * 1e3:.2f} us per pair of nodes", flush=True)

# cross-stream hand-offs: main kernel -> side kernel -> main kernel ... (fork/join per node)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cur = torch.cuda.current_stream()
    for _ in range(200):
        x.add_(1.0)
        s2.wait_stream(cur)
        with torch.cuda.stream(s2):
            y.add_(1.0)
        cur.wait_stream(s2)
t = timeit(g.replay)
print(f"ping-pong between two streams: {t / 200 * 1e3:.2f} us per (main, side) pair", flush=True)
