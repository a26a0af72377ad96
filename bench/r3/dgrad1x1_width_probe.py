"""1x1 data gradient (dBuf[:, :N] += mask * sc * (dh W1), BN-backward statistics) as a function of the column count N:
what do the partly filled last column tiles (N % 128 != 0) cost?  Captured-graph timing, real kernel, synthetic tensors.

    python bench/r3/dgrad1x1_width_probe.py [--batch 30]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from ddl import ops

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=30)
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--bm", type=int, default=0)
a = ap.parse_args()
ext = ops.require_ext()
dev = torch.device("cuda", 0)
torch.manual_seed(0)


def graph_time(fn):
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(4):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(a.iters):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters / 4 * 1e3


for blk, H, Cf, widths in (("b1", 56, 256, (64, 96, 128, 160, 192, 224, 256)),
                           ("b2", 28, 512, (128, 160, 192, 224, 256, 288, 320, 384, 448, 480, 512))):
    M = a.batch * H * H
    buf, dbuf = torch.randn(M, Cf, device=dev), torch.randn(M, Cf, device=dev)
    sc, sh = torch.rand(Cf, device=dev) + 0.5, torch.randn(Cf, device=dev)
    h, g2 = torch.randn(M, 128, device=dev), torch.randn(M, 128, device=dev)
    k = torch.randn(3, 128, device=dev) * 0.1
    for N in widths:
        w1 = torch.randn(128, N, device=dev) * 0.05

        def f(N=N, w1=w1):
            ext.conv_gemm(g2, h, k[0], k[1], k[2], False, M, H, H, 1, False, w1, True, dbuf[:, :N], True, True,
                          buf[:, :N], sc, sh, 2, a.bm, 0)
        us = graph_time(f)
        gb = (2 * M * 128 + 3 * M * N) * 4 / 1e9
        print(f"{blk}_dgrad1x1 N={N:4d}: {us:6.1f} us   {2.0 * M * N * 128 / us / 1e6:6.1f} TFLOP/s   {gb / us * 1e3:5.2f} TB/s of minimal traffic "
              f"({(N + 127) // 128} column tiles, last {N - (N - 1) // 128 * 128} wide)", flush=True)
