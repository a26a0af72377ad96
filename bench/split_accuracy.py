#!/usr/bin/env python3
"""Accuracy of the three matrix-core paths of conv_gemm against an fp64 evaluation of the same operation:
prec 0 = v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate), 1 = bf16 inputs, 3 = fp32 operands split into three
bf16 pieces, six bf16 MFMAs per product (csrc/conv_gemm.hip).  Prints max / rms error relative to the rms of the
exact result, for the forward 1x1 (BN+ReLU prologue, K = 160 / 640), the forward 3x3 (K = 1152) and both data
gradients.  The claim under test: path 3 is fp32-equivalent (its error is of the size of path 0's)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from ddl import ops

ext = ops.require_ext()
dev = torch.device("cuda", 0)
torch.manual_seed(0)


def err(got, ref):
    d = (got.double() - ref)
    scale = ref.pow(2).mean().sqrt()
    return float(d.abs().max() / scale), float(d.pow(2).mean().sqrt() / scale)


def show(name, fn, ref):
    out = {}
    for prec in (0, 3, 1):
        out[prec] = err(fn(prec), ref)
    print(f"{name:34s} " + "  ".join(f"prec {p}: max {out[p][0]:.2e} rms {out[p][1]:.2e}" for p in (0, 3, 1))
          + f"   rms(3)/rms(0) = {out[3][1] / out[0][1]:.2f}", flush=True)
    return out


for (B, H, C) in ((30, 56, 160), (30, 14, 640), (8, 28, 320)):
    M = B * H * H
    x = torch.randn(M, C, device=dev) * 1.7 + 0.3
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.3
    w1 = torch.randn(128, C, device=dev) * 0.05
    a = torch.relu(sc.double() * x.double() + sh.double())
    ref = a @ w1.double().t()

    def fwd1(prec):
        out = torch.empty(M, 128, device=dev)
        ext.conv_gemm(x, None, sc, None, sh, True, M, H, H, 1, False, w1, False, out, False, False, None, None, None,
                      0, 0, prec)
        return out
    show(f"fwd1x1 M={M} K={C}", fwd1, ref)

    # forward 3x3 (K = 1152): prologue BN+ReLU on h, zero padding after it
    h = torch.randn(M, 128, device=dev)
    sc2, sh2 = torch.rand(128, device=dev) + 0.5, torch.randn(128, device=dev) * 0.3
    w2 = torch.randn(32, 9 * 128, device=dev) * 0.03
    a2 = torch.relu(sc2.double() * h.double() + sh2.double()).view(B, H, H, 128).permute(0, 3, 1, 2)
    wt = w2.double().view(32, 3, 3, 128).permute(0, 3, 1, 2)
    ref3 = F.conv2d(a2, wt, padding=1).permute(0, 2, 3, 1).reshape(M, 32)

    def fwd3(prec):
        out = torch.empty(M, 32, device=dev)
        ext.conv_gemm(h, None, sc2, None, sh2, True, M, H, H, 9, False, w2, False, out, False, False, None, None, None,
                      0, 1000 + 9 if prec == 0 else 0, prec)          # prec 0 on the conv_gemm tiles too (not the halo kernel)
        return out
    show(f"fwd3x3 M={M} K=1152", fwd3, ref3)

    # 1x1 data gradient: dBuf += sc * mask(dh W1), dh = k0 g2 + k1 h + k2
    g2 = torch.randn(M, 128, device=dev) * 0.01
    k = torch.randn(3, 128, device=dev) * 0.1
    k[0] = torch.rand(128, device=dev) + 0.5
    dbuf0 = torch.randn(M, C, device=dev) * 0.01
    dh = k[0].double() * g2.double() + k[1].double() * h.double() + k[2].double()
    z = sc.double() * x.double() + sh.double()
    refd = dbuf0.double() + sc.double() * torch.where(z > 0, dh @ w1.double(), torch.zeros((), dtype=torch.float64, device=dev))

    def dg1(prec):
        dbuf = dbuf0.clone()
        ext.conv_gemm(g2, h, k[0], k[1], k[2], False, M, H, H, 1, False, w1, True, dbuf, True, True, x, sc, sh, 0, 0, prec)
        return dbuf
    show(f"dgrad1x1 M={M} N={C}", dg1, refd)
print("SPLIT ACCURACY DONE")
