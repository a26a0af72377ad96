#!/usr/bin/env python3
"""Micro-benchmark of single native kernels on representative DenseNet-121 shapes (B=30 by default):
event-timed, reports achieved TFLOP/s.  `--only NAME` restricts to one case (for rocprofv3 --pmc)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ddl import ops

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=30)
ap.add_argument("--only", default=None)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--bm", type=int, default=0)
ap.add_argument("--bf16", action="store_true", help="bf16 MFMA inputs for the conv_gemm cases")
ap.add_argument("--prec", type=int, default=None, help="matrix-core path of the conv_gemm cases: 0 fp32 MFMA, 1 bf16, 3 fp32 as "
                                                     "three bf16 pieces (6 bf16 MFMAs per product)")
ap.add_argument("--min-rows", type=int, default=64, help="wgrad: minimum pixel rows per workgroup")
ap.add_argument("--sub", type=int, default=0, help="wgrad 1x1: force the sub-tile count (1, 3, 4) of the channel tiles")
ap.add_argument("--defer", action="store_true", help="wgrad: queue the slab reductions (timed: the slab kernels alone)")
ap.add_argument("--dump", default="", help="--probe: directory for per-workgroup (start, end, placement) CSVs")
ap.add_argument("--probe", action="store_true", help="phase timestamps of the first / last workgroup (needs a "
                                                   "DDL_BUILD_DEFINES=-DDDL_PHASE_PROBE build of the extension)")
ap.add_argument("--sweep", action="store_true", help="time every conv_gemm tile config per case")
ap.add_argument("--cfg", type=int, default=-1, help="explicit conv_gemm tile config id (csrc/conv_gemm.hip)")
a = ap.parse_args()
if a.cfg >= 0:
    a.bm = 1000 + a.cfg
a.bf16 = int(a.bf16) if a.prec is None else a.prec
WPREC = 1 if a.bf16 == 1 else 0          # weight gradients: bf16 MFMA only in the bf16 mode
ext = ops.require_ext()
dev = torch.device("cuda", 0)
B = a.batch
torch.manual_seed(0)


def timeit(fn, iters=a.iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3      # us


cases = []
NCFG = 22
WQ = dict(queue=ext.SlabQueue()) if a.defer else {}


def case(name, flops):
    def deco(fn):
        cases.append((name, flops, fn))
        return fn
    return deco


def mk(H, C, Cf):
    M = B * H * H
    buf = torch.randn(M, Cf, device=dev)
    return M, buf


for (blk, H, Cl, Cf) in [("b1", 56, 160, 256), ("b2", 28, 320, 512), ("b3", 14, 640, 1024), ("b4", 7, 768, 1024)]:
    M, buf = mk(H, Cl, Cf)
    sc, sh = torch.rand(Cf, device=dev) + 0.5, torch.randn(Cf, device=dev)
    w1 = torch.randn(128, Cl, device=dev) * 0.05
    w2 = torch.randn(32, 9 * 128, device=dev) * 0.05
    h = torch.randn(M, 128, device=dev)
    g2 = torch.randn(M, 128, device=dev)
    out128 = torch.empty(M, 128, device=dev)
    dbuf = torch.randn(M, Cf, device=dev)
    ones = torch.ones(1024, device=dev)
    k = torch.randn(3, 128, device=dev) * 0.1
    gw1 = torch.zeros(128, Cl, device=dev)
    gw2 = torch.zeros(32, 9 * 128, device=dev)

    def f1(M=M, buf=buf, Cl=Cl, sc=sc, sh=sh, w1=w1, out=out128, H=H):
        ext.conv_gemm(buf[:, :Cl], None, sc, None, sh, True, M, H, H, 1, False, w1, False, out, False, False,
                      None, None, None, 1, a.bm, a.bf16)
    cases.append((f"{blk}_fwd1x1_K{Cl}", 2.0 * M * Cl * 128, f1))

    def f2(M=M, h=h, sc=sc, sh=sh, w2=w2, buf=buf, Cl=Cl, H=H):
        ext.conv_gemm(h, None, sc, None, sh, True, M, H, H, 9, False, w2, False, buf[:, Cl:Cl + 32], False, False,
                      None, None, None, 1, a.bm, a.bf16)
    cases.append((f"{blk}_fwd3x3", 2.0 * M * 1152 * 32, f2))

    acc_in = (torch.rand(8, 2, Cf, device=dev) * 2 ** 25 * M).long()
    acc_in[:, 1] += acc_in[:, 0]                 # sum of squares >= sum: a valid (mean, var) pair
    acc_out = torch.zeros(8, 2, 128, dtype=torch.int64, device=dev)
    tabs = torch.rand(8, Cf, device=dev) + 0.5

    def f1f(M=M, buf=buf, Cl=Cl, sc=sc, sh=sh, w1=w1, out=out128, H=H, acc_in=acc_in, acc_out=acc_out, tabs=tabs):
        ext.conv_fwd_fused(buf[:, :Cl], acc_in, Cl - 32, sc, sh, tabs[0], tabs[1], tabs[2], tabs[3], tabs[4], tabs[5],
                           tabs[6], float(M), 0.1, 1e-5, M, H, H, 1, w1, out, acc_out, a.bm, a.bf16)
    cases.append((f"{blk}_fwd1x1_fused", 2.0 * M * Cl * 128, f1f))

    def f2f(M=M, h=h, sc=sc, sh=sh, w2=w2, buf=buf, Cl=Cl, H=H, acc_in=acc_in, acc_out=acc_out, tabs=tabs):
        ext.conv_fwd_fused(h, acc_in[:, :, :128], 0, sc, sh, tabs[0], tabs[1], tabs[2], tabs[3], tabs[4], tabs[5],
                           tabs[6], float(M), 0.1, 1e-5, M, H, H, 9, w2, buf[:, Cl:Cl + 32], acc_out[:, :, :32],
                           a.bm, a.bf16)
    cases.append((f"{blk}_fwd3x3_fused", 2.0 * M * 1152 * 32, f2f))

    def f2a(M=M, h=h, sc=sc, sh=sh, w2=w2, buf=buf, Cl=Cl, H=H, acc_out=acc_out):      # tables in, accumulators out
        ext.conv_gemm(h, None, sc, None, sh, True, M, H, H, 9, False, w2, False, buf[:, Cl:Cl + 32], False, False,
                      None, None, None, 1, a.bm, a.bf16, acc_out=acc_out[:, :, :32])
    cases.append((f"{blk}_fwd3x3_accout", 2.0 * M * 1152 * 32, f2a))

    def f3(M=M, dbuf=dbuf, buf=buf, Cl=Cl, ones=ones, sc=sc, sh=sh, w2=w2, g2=g2, h=h, H=H):
        ext.conv_gemm(dbuf[:, Cl:Cl + 32], buf[:, Cl:Cl + 32], ones[:32], sc[:32], sh[:32], False, M, H, H, 9, True,
                      w2, True, g2, True, False, h, sc, sh, 2, a.bm, a.bf16)
    cases.append((f"{blk}_dgrad3x3", 2.0 * M * 1152 * 32, f3))

    def f4(M=M, g2=g2, h=h, k=k, w1=w1, dbuf=dbuf, buf=buf, Cl=Cl, sc=sc, sh=sh, H=H):
        ext.conv_gemm(g2, h, k[0], k[1], k[2], False, M, H, H, 1, False, w1, True, dbuf[:, :Cl], True, True,
                      buf[:, :Cl], sc, sh, 2, a.bm, a.bf16)
    cases.append((f"{blk}_dgrad1x1_N{Cl}", 2.0 * M * Cl * 128, f4))

    def f5(M=M, g2=g2, h=h, k=k, buf=buf, Cl=Cl, sc=sc, sh=sh, gw1=gw1, H=H):
        ext.conv_wgrad(g2, h, k[0], k[1], k[2], buf[:, :Cl], sc, sh, True, M, H, H, 1, gw1, a.min_rows, force_sub=a.sub, prec=WPREC, **WQ)
    cases.append((f"{blk}_wgrad1x1_C{Cl}", 2.0 * M * Cl * 128, f5))

    def f6(M=M, dbuf=dbuf, buf=buf, Cl=Cl, ones=ones, sc=sc, sh=sh, h=h, gw2=gw2, H=H):
        ext.conv_wgrad(dbuf[:, Cl:Cl + 32], buf[:, Cl:Cl + 32], ones[:32], sc[:32], sh[:32], h, sc, sh, True, M, H, H,
                       9, gw2, a.min_rows, prec=WPREC, **WQ)
    cases.append((f"{blk}_wgrad3x3", 2.0 * M * 1152 * 32, f6))

for (blk, H, Cl, Cf) in [("b1", 56, 64, 256), ("b1", 56, 96, 256), ("b1", 56, 224, 256), ("b2", 28, 160, 512), ("b2", 28, 480, 512),
                         ("b3", 14, 288, 1024), ("b3", 14, 992, 1024), ("b4", 7, 544, 1024), ("b4", 7, 992, 1024)]:
    M, buf = mk(H, Cl, Cf)
    sc, sh = torch.rand(Cf, device=dev) + 0.5, torch.randn(Cf, device=dev)
    h, g2 = torch.randn(M, 128, device=dev), torch.randn(M, 128, device=dev)
    k = torch.randn(3, 128, device=dev) * 0.1
    gw1 = torch.zeros(128, Cl, device=dev)

    def f5x(M=M, g2=g2, h=h, k=k, buf=buf, Cl=Cl, sc=sc, sh=sh, gw1=gw1, H=H):
        ext.conv_wgrad(g2, h, k[0], k[1], k[2], buf[:, :Cl], sc, sh, True, M, H, H, 1, gw1, a.min_rows, force_sub=a.sub, prec=WPREC, **WQ)
    cases.append((f"{blk}_wgrad1x1_C{Cl}", 2.0 * M * Cl * 128, f5x))

PHASES = ["entry", "loads issued + tables", "first slab in LDS", "K loop", "K-split combine", "epilogue stores issued",
          "statistics", "vmcnt(0) + barrier", "end"]


def probe(name, fn):
    dbg = torch.zeros(32 + 3 * 2048, dtype=torch.int64, device=dev)
    ext.set_phase_probe(dbg)
    fn()
    torch.cuda.synchronize()
    ext.set_phase_probe(None)
    t = dbg.cpu().tolist()
    if "wgrad" in name:      # per-phase sums over the pixel steps (conv_wgrad.hip, probe builds)
        names = ["prologue", "load issue", "LDS reads + MFMA issue", "exposed load wait", "prologue math + LDS writes", "barrier"]
        for label, off in (("WG 0", 0), ("WG 264", 16)):
            v = t[off:off + 10]
            if v[6] == 0:
                continue
            print(f"    {label}: {v[6]} steps, {(v[8] - v[7]) / 100:.2f} us at {v[9] / max(1, v[8] - v[7]) / 10:.2f} GHz shader clock | "
                  + " | ".join(f"{n} {x / 100:.2f}" for n, x in zip(names, v[:6])), flush=True)
        wg = [(t[32 + 3 * i], t[33 + 3 * i], t[34 + 3 * i]) for i in range(2048) if t[32 + 3 * i]]
        if wg:
            t0 = min(w[0] for w in wg)
            if a.dump:
                with open(f"{a.dump}/{name}.csv", "w") as f:
                    f.write("bid,start_us,end_us,xcc,hw_id\n")
                    for i in range(2048):
                        if t[32 + 3 * i]:
                            f.write(f"{i},{(t[32 + 3 * i] - t0) / 100:.2f},{(t[33 + 3 * i] - t0) / 100:.2f},"
                                    f"{(t[34 + 3 * i] >> 32) & 0xf},{t[34 + 3 * i] & 0xffffffff}\n")
            starts = sorted((w[0] - t0) / 100 for w in wg)
            ends = sorted((w[1] - t0) / 100 for w in wg)
            life = sorted((w[1] - w[0]) / 100 for w in wg)
            q = lambda v, f: v[min(len(v) - 1, int(f * len(v)))]
            print(f"    {len(wg)} workgroups: start p50 {q(starts, .5):.1f} p90 {q(starts, .9):.1f} max {starts[-1]:.1f} us | "
                  f"end p10 {q(ends, .1):.1f} p50 {q(ends, .5):.1f} max {ends[-1]:.1f} us | life p10 {q(life, .1):.1f} p50 {q(life, .5):.1f} "
                  f"p90 {q(life, .9):.1f} max {life[-1]:.1f} us", flush=True)
            import collections
            per_cu = collections.Counter()
            for s0, e0, idw in wg:
                xcc, hw = idw >> 32, idw & 0xffffffff
                per_cu[(xcc & 0xf, (hw >> 8) & 0xff)] += 1       # (XCC, SE / SH / CU bits of HW_ID)
            hist = collections.Counter(per_cu.values())
            print(f"    placement: {len(per_cu)} distinct (XCC, SE, CU); workgroups per CU histogram {dict(sorted(hist.items()))}", flush=True)
            late = [w for w in wg if (w[0] - t0) / 100 > 5.0]
            print(f"    {len(late)} workgroups start more than 5 us after the first", flush=True)
        return
    for label, off in (("first WG", 0), ("last WG", 16)):
        v = t[off:off + 9]
        if v[0] == 0:
            continue
        parts, prev = [], v[0]
        for i in range(1, 9):
            if v[i]:
                parts.append(f"{PHASES[i]} +{(v[i] - prev) / 100:.2f}")
                prev = v[i]
        print(f"    {label}: " + " | ".join(parts) + f" | total {(prev - v[0]) / 100:.2f} us"
              + (f" | starts {(t[16] - t[0]) / 100:+.2f} us after the first WG" if off == 16 else ""), flush=True)


for name, flops, fn in cases:
    if WQ:
        WQ["queue"].flush()
    if a.only and a.only not in name:
        continue
    if a.probe:
        us = timeit(fn)
        print(f"{name:28s} {us:7.1f} us", flush=True)
        probe(name, fn)
        continue
    if a.sweep:
        if "wgrad" in name:
            continue
        res = []
        for cfg in [-1] + list(range(NCFG)):
            a.bm = 0 if cfg < 0 else 1000 + cfg
            try:
                res.append(timeit(fn, 10))
            except RuntimeError:          # a tile configuration without an fp32x3 form (LDS)
                res.append(float("inf"))
        best = min(range(1, NCFG + 1), key=lambda i: res[i])
        print(f"{name:20s} auto {res[0]:6.1f} | " + " ".join(f"{r:6.1f}" for r in res[1:]) +
              f" | best cfg {best - 1} {res[best]:6.1f} us", flush=True)
        continue
    us = timeit(fn)
    print(f"{name:24s} {us:9.1f} us  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)
