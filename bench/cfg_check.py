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
# ---- kind 2: 3x3 data gradient (two sources, k-major flipped taps, mask) on 5 x 28 x 28 and 30 x 14 x 14 ----
import torch.nn.functional as F  # noqa: E402
from ddl.models.native_densenet import NativeDenseNetStage  # noqa: E402
for (Bn, Hh) in [(5, 28), (30, 14)]:
    Mm = Bn * Hh * Hh
    hh = rnd(Mm, 128, seed=21)
    sc2, sh2 = rnd(128, seed=22).abs() + 0.5, rnd(128, seed=23)
    w3 = rnd(32, 128, 3, 3, seed=24, scale=0.05)
    wn = NativeDenseNetStage._to_native(w3)
    dbuf, xbuf = rnd(Mm, 256, seed=25), rnd(Mm, 256, seed=26)
    d, x = dbuf[:, 64:96], xbuf[:, 64:96]
    q, pp = rnd(32, seed=27, scale=0.1), rnd(32, seed=28, scale=0.1)
    ones = torch.ones(32, device=dev)
    dy64 = (d.double() + q.double() * x.double() + pp.double()).view(Bn, Hh, Hh, 32).permute(0, 3, 1, 2)
    gref = F.conv_transpose2d(dy64, w3.double(), padding=1).permute(0, 2, 3, 1).reshape(Mm, 128)
    pre = hh.double() * sc2.double() + sh2.double()
    gref = torch.where(pre > 0, gref, torch.zeros_like(gref))
    sure = (pre.abs() > 1e-5).double()
    for cfg in [-1] + cfgs:
        g2o = torch.full((Mm, 128), float("nan"), device=dev)
        print(f"  launching kind 2 M={Mm} cfg {cfg}", flush=True)
        ext.conv_gemm(d, x, ones, q, pp, False, Mm, Hh, Hh, 9, True, wn, True, g2o, True, False, hh, sc2, sh2, 2,
                      0 if cfg < 0 else 1000 + cfg)
        torch.cuda.synchronize()
        bad += report(2, cfg, g2o.double() * sure, gref * sure, f"M={Mm}")
print("BAD CONFIGS:", bad, flush=True)
sys.exit(1 if bad else 0)
