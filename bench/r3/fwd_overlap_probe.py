"""Upper bound of overlapping a dense layer's 3x3 convolution with the prefix part of the NEXT layer's 1x1 convolution.

In a dense block layer l+1's 1x1 convolution reads channels [0, C_l + 32); only the last 32 come from layer l's 3x3
convolution.  At batch 30 the block 3 / 4 kernels are 10-20 us launches on a mostly idle chip, so the prefix part
(K = C_l) could run BESIDE layer l's 3x3 on a second stream, followed by a K = 32 remainder.  This probe times both
forms as captured graphs with the real kernels on synthetic tensors (the remainder's accumulate-into-output epilogue
does not exist yet: it is timed as a plain K = 32 convolution, i.e. slightly optimistic).

    python bench/r3/fwd_overlap_probe.py [--batch 30]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from ddl import ops

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=30)
ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
ext = ops.require_ext()
dev = torch.device("cuda", 0)
torch.manual_seed(0)


def block(H, C0, L):
    M, Cf = a.batch * H * H, C0 + 32 * L
    buf = torch.randn(M, Cf, device=dev)
    sc, sh = torch.rand(Cf, device=dev) + 0.5, torch.randn(Cf, device=dev)
    h = torch.randn(M, 128, device=dev)
    out = [torch.empty(M, 128, device=dev) for _ in range(2)]
    w1 = [torch.randn(128, C0 + 32 * l, device=dev) * 0.05 for l in range(L)]
    w1p = [torch.randn(128, max(32, C0 + 32 * (l - 1)), device=dev) * 0.05 for l in range(L)]     # prefix weights
    w1r = torch.randn(128, 32, device=dev) * 0.05
    w2 = torch.randn(32, 9 * 128, device=dev) * 0.05

    def c1(l, o):
        C = C0 + 32 * l
        ext.conv_gemm(buf[:, :C], None, sc, None, sh, True, M, H, H, 1, False, w1[l], False, o, False, False,
                      None, None, None, 1, 0, 0)

    def c1_prefix(l, o):
        C = C0 + 32 * (l - 1)
        ext.conv_gemm(buf[:, :C], None, sc, None, sh, True, M, H, H, 1, False, w1p[l], False, o, False, False,
                      None, None, None, 1, 0, 0)

    def c1_rest(l, o):
        C = C0 + 32 * (l - 1)
        ext.conv_gemm(buf[:, C:C + 32], None, sc[C:C + 32].contiguous(), None, sh[C:C + 32].contiguous(), True, M, H, H, 1, False,
                      w1r, False, o, False, False, None, None, None, 1, 0, 0)

    def c3(l):
        C = C0 + 32 * l
        ext.conv_gemm(h, None, sc, None, sh, True, M, H, H, 9, False, w2, False, buf[:, C:C + 32], False, False,
                      None, None, None, 1, 0, 0)

    def sequential():
        for l in range(L):
            c1(l, out[l & 1])
            c3(l)

    side = torch.cuda.Stream(dev)

    def overlapped():
        main = torch.cuda.current_stream(dev)
        c1(0, out[0])
        for l in range(L):
            if l + 1 < L:
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    c1_prefix(l + 1, out[(l + 1) & 1])
                    done = torch.cuda.Event()
                    done.record(side)
            c3(l)
            if l + 1 < L:
                main.wait_event(done)
                c1_rest(l + 1, out[(l + 1) & 1])

    res = {}
    for name, fn in (("sequential", sequential), ("overlapped", overlapped)):
        s = torch.cuda.Stream(dev)
        with torch.cuda.stream(s):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                fn()
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(a.iters):
                g.replay()
            e1.record(s)
            torch.cuda.synchronize()
            res[name] = e0.elapsed_time(e1) / a.iters * 1e3
    return res


for name, H, C0, L in (("block 2", 28, 128, 12), ("block 3", 14, 256, 24), ("block 4", 7, 512, 16)):
    r = block(H, C0, L)
    print(f"{name}: {L} layers, forward 1x1 + 3x3 chain: sequential {r['sequential']:.0f} us, "
          f"3x3 beside the next layer's prefix 1x1 (+ K = 32 remainder) {r['overlapped']:.0f} us "
          f"({r['overlapped'] / r['sequential']:.2f}x)", flush=True)
