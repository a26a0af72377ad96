"""Numerics of every conv_gemm tile configuration (explicit ids, csrc/conv_gemm.hip launch_by_id) on shapes with
partial row / column tiles, against fp64 references.  Prints one line per (kind, config) as it goes (stdout only:
a faulting kernel takes the box and its files with it)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ddl import ops

ext = ops.require_ext()
dev = torch.device("cuda", 0)
NCFG = int(os.environ.get("NCFG", "22"))
only = [int(v) for v in os.environ.get("CFGS", "").split(",") if v]
cfgs = only or list(range(NCFG))


def rnd(*shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev)


def report(kind, cfg, got, ref, extra=""):
    err = float((got.double() - ref).norm() / ref.norm())
    bad = not (err < 2e-5)
    print(f"kind {kind} cfg {cfg:2d} rel-l2 {err:.2e}{' BAD' if bad else ''} {extra}", flush=True)
    return bad


bad = 0
for (M, K, N, Cf) in [(1000, 128, 320, 512), (23520, 128, 320, 512)]:
    # ---- kind 3: 1x1 data gradient, two-source prologue, masked accumulate, statistics ----
    g2, h = rnd(M, K, seed=1), rnd(M, K, seed=2)
    k = rnd(3, K, seed=3, scale=0.1)
    w = rnd(K, N, seed=4, scale=0.05)                 # forward weight [128][Cl], read k-major
    x = rnd(M, Cf, seed=5)
    sc, sh = rnd(Cf, seed=6).abs() + 0.5, rnd(Cf, seed=7)
    d0 = rnd(M, Cf, seed=8)
    a = k[0].double() * g2.double() + k[1].double() * h.double() + k[2].double()
    gfull = a @ w.double()
    keep = (x[:, :N].double() * sc[:N].double() + sh[:N].double()) > 0
    ref = d0[:, :N].double() + sc[:N].double() * torch.where(keep, gfull, torch.zeros_like(gfull))
    for cfg in [-1] + cfgs:
        d = d0.clone()
        print(f"  launching kind 3 M={M} cfg {cfg}", flush=True)
        ext.conv_gemm(g2, h, k[0], k[1], k[2], False, M, 1, 1, 1, False, w, True, d[:, :N], True, True,
                      x[:, :N], sc, sh, 2, 0 if cfg < 0 else 1000 + cfg)
        torch.cuda.synchronize()
        bad += report(3, cfg, d[:, :N], ref, f"M={M}")
        assert torch.equal(d[:, N:], d0[:, N:])
    # ---- kind 0: forward 1x1 with BN+ReLU prologue (K = 320 channels of the prefix) ----
    wf = rnd(128, N, seed=9, scale=0.05)
    ref0 = torch.relu(x[:, :N].double() * sc[:N].double() + sh[:N].double()) @ wf.double().t()
    for cfg in [-1] + cfgs:
        out = torch.full((M, 128), float("nan"), device=dev)
        print(f"  launching kind 0 M={M} cfg {cfg}", flush=True)
        ext.conv_gemm(x[:, :N], None, sc, None, sh, True, M, 1, 1, 1, False, wf, False, out, False, False,
                      None, None, None, 1, 0 if cfg < 0 else 1000 + cfg)
        torch.cuda.synchronize()
        bad += report(0, cfg, out, ref0, f"M={M}")
print("BAD CONFIGS:", bad, flush=True)
sys.exit(1 if bad else 0)
