"""DenseNet-121 on hand-written gfx950 kernels: the MI355X-native engine.

Same architecture, parameter names and numerics (fp32) as ``ddl.models.densenet`` /
``torchvision.models.densenet121`` (reference: single.py:297-299), executed very differently:

  layout      NHWC; a dense block owns ONE pre-allocated buffer ``[B,H,W,C_final]``.  Layer i
              reads the channel prefix ``[:C_i]`` and its 3x3 conv writes channels
              ``[C_i:C_i+32]`` in place - ``torch.cat`` (O(L^2) copies) does not exist.
  BN          batch statistics of a channel are computed once, in the epilogue of the conv that
              produced it, and shared by all later ``norm1`` layers of the block (they normalise
              the same data); each layer only derives ``scale = gamma*rstd, shift = beta-mean*scale``.
              BN+ReLU is applied while the following conv stages its A operand: the normalised
              tensor never exists in HBM (SURVEY K3 "best fused as prologue of the 1x1 conv").
  convs       fp32 MFMA implicit GEMM (csrc/conv_gemm.hip, conv_wgrad.hip).
  backward    hand-derived.  ``dBuf`` (gradient of the block buffer) is accumulated in place by
              the 1x1 data-gradient epilogue (ReLU mask, BN scale, per-channel sums fused).  The
              two "mean-subtraction" terms of every BN backward are per-channel affine functions
              of the buffer itself, so they are *deferred*: accumulated as two vectors (NQ, NP)
              and applied on the fly when a channel slice is finally consumed
              (``dY = dBuf + NQ*x + NP`` in the 3x3 kernels' operand prologue).  Per layer the wide
              prefix is touched 3 times (mask read, dBuf read-modify-write) instead of 5+.
  params      one flat arena (``FlatArena``): gradients are written straight into it (no
              per-parameter autograd accumulation kernels); it doubles as the DP bucket space.
  units       every unit (stem, block, transition, head-pool) is ONE autograd node, so pipeline
              stages and the data-parallel reducer see ~10 nodes instead of ~600.

Nothing in the training step goes through MIOpen / rocBLAS: the stem (7x7/2 conv as an MFMA
implicit GEMM over an NHWC4 repack, BN + ReLU + max-pool fused) and the 1024->5 classifier are
native kernels too (csrc/stem_kernels.hip).

Weight layouts (converted at the checkpoint boundary by ``ddl_export_tensor`` /
``ddl_import_tensor``, so checkpoints stay in the torchvision layout - SURVEY 5.4):
  1x1 conv  [N, C]           (= torch [N, C, 1, 1])
  3x3 conv  [N, 9*C] (tap-major, channel-minor = torch.permute(0, 2, 3, 1))
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from .. import ops
from ..utils.arena import FlatArena, FlatBuffers, reverse_named_parameters
from ..utils.knobs import knob
from ..utils.streams import new_stream
from .densenet import (BLOCK_CONFIG, BN_SIZE, GROWTH, UNIT_NAMES, DenseNetStage, block_channels, parse_unit)

BN_MOMENTUM = 0.1
BN_EPS = 1e-5
MID = BN_SIZE * GROWTH      # 128 bottleneck channels


def _ext():
    return ops.require_ext()


# ======================================================================================
# dense block
# ======================================================================================
class _BlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_in: torch.Tensor, anchor: torch.Tensor, unit: "BlockUnit"):
        ext = _ext()
        train = unit.stage.training
        bf = unit.stage.bf16
        B, H, W, C0 = x_in.shape
        L, Cf = unit.num_layers, unit.c_out
        M = B * H * W
        dev = x_in.device
        f32 = dict(dtype=torch.float32, device=dev)
        buf = torch.empty(B, H, W, Cf, **f32)
        buf2d = buf.view(M, Cf)
        stats = torch.empty(3, Cf, **f32)                   # chan_mean / chan_var / chan_rstd
        sc1 = torch.empty(L, Cf, **f32)
        sh1 = torch.empty(L, Cf, **f32)
        bn2 = torch.empty(L, 5, MID, **f32)                 # mean2, var2, rstd2, sc2, sh2
        h = torch.empty(L, M, MID, **f32)                   # bottleneck activations (pre-BN2)
        x2d = x_in.reshape(M, C0)
        if train and unit.stage.fused_stats:
            # Fused statistics: channel sums live in int64 fixed-point accumulators (integer atomics from the
            # producing kernel's epilogue - the copy of the incoming channels included); each conv derives its BN
            # scale / shift from them in its prologue and its workgroup (0,0) writes the tables backward needs +
            # the running statistics.  No bn_prep launches between the kernels of the block.
            R = ext.STAT_REPLICAS
            acc = torch.zeros(R * 2 * (Cf + L * MID), dtype=torch.int64, device=dev)
            acc_b, acc_h = acc[:R * 2 * Cf].view(R, 2, Cf), acc[R * 2 * Cf:].view(R, 2, L * MID)
            ext.rows_op(0, M, x2d, None, None, None, buf2d[:, :C0], H, W, False, acc_out=acc_b)
            for l, lay in enumerate(unit.layers):
                C = C0 + GROWTH * l
                hs = slice(l * MID, (l + 1) * MID)
                ext.conv_fwd_fused(buf2d[:, :C], acc_b, 0 if l == 0 else C - GROWTH, lay.g1, lay.b1, lay.rm1, lay.rv1,
                                   sc1[l], sh1[l], stats[0], stats[1], stats[2], float(M), BN_MOMENTUM, BN_EPS,
                                   M, H, W, 1, lay.w1, h[l], acc_h[:, :, hs], 0, bf)
                ext.conv_fwd_fused(h[l], acc_h[:, :, hs], 0, lay.g2, lay.b2, lay.rm2, lay.rv2, bn2[l, 3], bn2[l, 4],
                                   bn2[l, 0], bn2[l, 1], bn2[l, 2], float(M), BN_MOMENTUM, BN_EPS, M, H, W, 9,
                                   lay.w2, buf2d[:, C:C + GROWTH], acc_b[:, :, C:C + GROWTH], 0, bf)
            if unit.defer_finalize:     # the following transition / head finalises the last slice in its bn_prep
                unit.pending_acc = (acc_b, Cf - GROWTH, GROWTH)
            else:
                unit.pending_acc = None
                ext.bn_acc_finalize(acc_b, Cf - GROWTH, GROWTH, float(M), BN_EPS, stats[0], stats[1], stats[2])
            unit.out_stats = stats
            ctx.unit = unit
            ctx.geom = (B, H, W, C0)
            ctx.save_for_backward(buf, stats, sc1, sh1, bn2, h)
            return buf
        part = ext.rows_op(0, M, x2d, None, None, None, buf2d[:, :C0], H, W, train)
        new0, n_new = 0, C0
        for l, lay in enumerate(unit.layers):
            C = C0 + GROWTH * l
            ext.bn_prep(part if train else None, new0, n_new, float(M), stats[0], stats[1], stats[2],
                        lay.g1, lay.b1, lay.rm1, lay.rv1, sc1[l], sh1[l], C, BN_MOMENTUM, BN_EPS, train)
            st = ext.conv_gemm(buf2d[:, :C], None, sc1[l], None, sh1[l], True, M, H, W, 1, False,
                               lay.w1, False, h[l], False, False, None, None, None, 1 if train else 0, 0, bf)
            ext.bn_prep(st if train else None, 0, MID, float(M), bn2[l, 0], bn2[l, 1], bn2[l, 2],
                        lay.g2, lay.b2, lay.rm2, lay.rv2, bn2[l, 3], bn2[l, 4], MID, BN_MOMENTUM, BN_EPS, train)
            part = ext.conv_gemm(h[l], None, bn2[l, 3], None, bn2[l, 4], True, M, H, W, 9, False,
                                 lay.w2, False, buf2d[:, C:C + GROWTH], False, False, None, None, None,
                                 1 if train else 0, 0, bf)
            new0, n_new = C, GROWTH
        if train:   # statistics of the last slice (consumed by the following transition / head)
            ext.bn_prep(part, new0, n_new, float(M), stats[0], stats[1], stats[2], None, None, None, None,
                        None, None, Cf, BN_MOMENTUM, BN_EPS, True)
        unit.out_stats = stats if train else None
        unit.pending_acc = None
        ctx.unit = unit
        ctx.geom = (B, H, W, C0)
        ctx.save_for_backward(buf, stats, sc1, sh1, bn2, h)
        return buf

    @staticmethod
    def backward(ctx, dbuf: torch.Tensor):
        ext = _ext()
        unit = ctx.unit
        buf, stats, sc1, sh1, bn2, h = ctx.saved_tensors
        B, H, W, C0 = ctx.geom
        L, Cf = unit.num_layers, unit.c_out
        M = B * H * W
        f32 = dict(dtype=torch.float32, device=buf.device)
        # The 1x1 data-gradient epilogues accumulate into dBuf IN PLACE.  That is only legitimate on a tensor this
        # engine produced itself for exactly this consumer (the following transition / head backward); any other
        # incoming gradient - a pipeline receive slot, a user hook's tensor - is copied first (autograd contract).
        owned = dbuf.is_contiguous() and dbuf.data_ptr() == unit.stage._owned_grad_ptr
        dbuf = dbuf if owned else dbuf.clone(memory_format=torch.contiguous_format)
        unit.stage._owned_grad_ptr = 0
        d2d = dbuf.view(M, Cf)
        buf2d = buf.view(M, Cf)
        # deferred terms dX = dBuf + nq * x + np, and (grouped backward) the statistics accumulators + arrival
        # tickets of the data-gradient kernels' tail bookkeeping: ONE zero-fill for all of them
        fused = unit.stage.fused_bwd
        G = unit.stage.wgrad_group_size(M)
        grouped = G > 1 and not fused
        R = ext.STAT_REPLICAS
        n_acc = L * R * 2 * (MID + Cf) if (grouped and unit.stage.tail_prep) else 0
        n_tick = L * 2 * 32 if n_acc else 0
        zero = torch.zeros(8 * n_acc + 4 * (2 * Cf + n_tick), dtype=torch.uint8, device=buf.device)
        acc = zero[:8 * n_acc].view(torch.int64).view(L, R, 2, MID + Cf) if n_acc else None
        fz = zero[8 * n_acc:].view(torch.float32)
        nq, np_ = fz[:Cf], fz[Cf:2 * Cf]
        # The transition / head behind this block left its own BN backward as (sc*g in dBuf, k1, k2): exactly the
        # form of the deferred terms, so its k1 / k2 vectors BECOME nq / np (no dX = dBuf + k1*x + k2 pass over the
        # whole block buffer: 47 us behind block 1 at batch 30).
        pend = unit.stage.take_pending_affine(unit)
        if pend is not None:
            nq, np_ = pend[0], pend[1]
        tickets = fz[2 * Cf:].view(torch.int32).view(L, 2, 32) if n_acc else None
        ones = unit.stage.ones
        # Weight gradients feed nothing downstream, so they run on a side stream beside the
        # data-gradient chain (parallel branches once the step is captured into a hipGraph).
        # g2 / k are double-buffered by layer parity so a layer's 1x1 wgrad may still be reading
        # them while the next layer's 3x3 dgrad runs.  (Measured at batch 30: 1 buffer 11.65 ms/step,
        # 2: 9.31, 3: 9.53, 4: 9.68, 6: 10.03 - a weight-gradient stream that lags further reads
        # operands that have left the L2.)
        main = torch.cuda.current_stream(buf.device)
        side = unit.stage.fork(main) if unit.stage.overlap_wgrad else None
        NB = unit.stage.g2_bufs
        k = torch.empty(NB, 3, MID, **f32)
        g2 = torch.empty(NB, M, MID, **f32)
        done = [None] * NB
        min_rows = unit.stage.wgrad_min_rows
        bf = unit.stage.bf16
        wbf = unit.stage.wgrad_prec       # weight gradients: bf16 MFMA in bf16 mode, fp32 MFMA otherwise (fp32x3 too)
        if fused:     # (sum g2, sum g2*h) accumulators: BN2-backward coefficients are derived by their consumers
            acc_g = torch.zeros(L, ext.STAT_REPLICAS, 2, MID, dtype=torch.int64, device=buf.device)
        # every layer's weight-gradient kernels leave partial slabs; ONE launch at the end of the block adds them
        # into the gradient arena (2 slab_reduce launches per layer otherwise: 116 of the step's 660 graph nodes)
        defer = unit.stage.defer_slab_reduce          # "block" | "layer" | "" (reduce behind every kernel)
        wq = dict(queue=ext.SlabQueue()) if defer else {}
        if grouped:
            _backward_grouped(ext, unit, G, main, side, d2d, buf2d, nq, np_, stats, sc1, sh1, bn2, h,
                              M, H, W, C0, L, bf, acc, tickets)
            if side is not None:
                unit.stage.join(main, side)
            dx = torch.empty(B, H, W, C0, **f32)
            ext.rows_op(4, M, d2d[:, :C0], buf2d[:, :C0], nq, np_, dx.view(M, C0), H, W, False)
            unit.stage._unit_grads_ready(unit)
            return dx, None, None
        for l in range(L - 1, -1, -1):
            lay = unit.layers[l]
            C = C0 + GROWTH * l
            par = l % NB
            sl = slice(C, C + GROWTH)
            dy, xs = d2d[:, sl], buf2d[:, sl]
            sc2, sh2 = bn2[l, 3], bn2[l, 4]
            if side is not None:
                if done[par] is not None:
                    main.wait_event(done[par])
                ev0 = torch.cuda.Event()
                ev0.record(main)
            # 3x3 data gradient (+ ReLU/BN2 mask, sum g2, sum g2*h)
            st = ext.conv_gemm(dy, xs, ones[:GROWTH], nq[sl], np_[sl], False, M, H, W, 9, True,
                               lay.w2, True, g2[par], True, False, h[l], sc2, sh2, 2, 0, bf,
                               acc_out=acc_g[l] if fused else None)
            # 3x3 weight gradient
            if side is not None:
                side.wait_event(ev0)
                with torch.cuda.stream(side):
                    ext.conv_wgrad(dy, xs, ones[:GROWTH], nq[sl], np_[sl], h[l], sc2, sh2, True, M, H, W, 9,
                                   lay.gw2, min_rows, **wq, prec=wbf)
            else:
                ext.conv_wgrad(dy, xs, ones[:GROWTH], nq[sl], np_[sl], h[l], sc2, sh2, True, M, H, W, 9,
                               lay.gw2, min_rows, **wq, prec=wbf)
            if fused:
                kd = (None, None, None)
                kf = dict(k_acc=acc_g[l], k_mean=bn2[l, 0], k_rstd=bn2[l, 2], k_scale=sc2, k_count=float(M))
                kg = dict(k_dgamma=lay.gg2, k_dbeta=lay.gb2)
            else:
                ext.bn_bwd_prep(st, float(M), bn2[l, 0], bn2[l, 2], sc2, lay.gg2, lay.gb2,
                                k[par, 0], k[par, 1], k[par, 2], MID, 0)
                kd, kf, kg = (k[par, 0], k[par, 1], k[par, 2]), {}, {}
            # 1x1 weight gradient (activation re-normalised on the fly)
            if side is not None:
                ev1 = torch.cuda.Event()
                ev1.record(main)
                side.wait_event(ev1)
                with torch.cuda.stream(side):
                    ext.conv_wgrad(g2[par], h[l], *kd, buf2d[:, :C], sc1[l], sh1[l],
                                   True, M, H, W, 1, lay.gw1, min_rows, **kf, **wq, prec=wbf)
                    if defer == "layer":
                        wq["queue"].flush()       # the layer's 3x3 + 1x1 slabs in one launch, still cache-resident
                    done[par] = torch.cuda.Event()
                    done[par].record(side)
            else:
                ext.conv_wgrad(g2[par], h[l], *kd, buf2d[:, :C], sc1[l], sh1[l],
                               True, M, H, W, 1, lay.gw1, min_rows, **kf, **wq, prec=wbf)
                if defer == "layer":
                    wq["queue"].flush()
            # 1x1 data gradient: dh = k0*g2 + k1*h + k2 on the fly; epilogue: mask, dBuf += sc1*g1, sums
            st1 = ext.conv_gemm(g2[par], h[l], *kd, False, M, H, W, 1, False,
                                lay.w1, True, d2d[:, :C], True, True, buf2d[:, :C], sc1[l], sh1[l], 2, 0, bf,
                                **kf, **kg)
            ext.bn_bwd_prep(st1, float(M), stats[0], stats[2], sc1[l], lay.gg1, lay.gb1, None, nq, np_, C, 1)
        if wq:
            if side is not None:
                with torch.cuda.stream(side):
                    wq["queue"].flush()
            else:
                wq["queue"].flush()
        if side is not None:
            unit.stage.join(main, side)
        dx = torch.empty(B, H, W, C0, **f32)
        ext.rows_op(4, M, d2d[:, :C0], buf2d[:, :C0], nq, np_, dx.view(M, C0), H, W, False)
        unit.stage._unit_grads_ready(unit)
        return dx, None, None


def _backward_grouped(ext, unit, G, main, side, d2d, buf2d, nq, np_, stats, sc1, sh1, bn2, h, M, H, W, C0, L, bf,
                      acc, tickets):
    """Dense-block backward with LAYER-GROUPED weight gradients (csrc/conv_wgrad.hip: wgrad_group_kernel).

    The data-gradient chain runs layer by layer as always; the weight gradients of G consecutive layers are issued
    as two launches (all 3x3, all 1x1) + one slab reduction on the side stream once the chain has produced the
    group's g2 / k - everything else they read stays valid to the end of the block: layer l's slice of dBuf and of
    the deferred-term vectors nq / np is final when its turn comes and later (lower) layers only touch channels
    below it.  g2 / k live in two sets of G buffers, so the side stream may work on one group while the chain fills
    the other."""
    f32 = dict(dtype=torch.float32, device=buf2d.device)
    ones = unit.stage.ones
    k = torch.empty(2, G, 3, MID, **f32)
    g2 = torch.empty(2, G, M, MID, **f32)
    done = [None, None]
    grp, gi = [], 0
    # BN-backward bookkeeping in the tail of the data-gradient kernels (csrc/conv_params.h TailPrep): statistics go
    # to fixed-point accumulators (`acc`, zeroed by the caller together with nq / np), the last-arriving workgroup
    # of a column tile derives dgamma / dbeta and the coefficients - no bn_bwd_prep launch between two kernels.
    tail = acc is not None

    def launch(group, set_):
        q = ext.SlabQueue()
        sls = [slice(C0 + GROWTH * l, C0 + GROWTH * (l + 1)) for l in group]
        lays = [unit.layers[l] for l in group]
        n = len(group)
        ext.conv_wgrad_group(9, [d2d[:, s] for s in sls], [buf2d[:, s] for s in sls], [ones[:GROWTH]] * n,
                             [nq[s] for s in sls], [np_[s] for s in sls], [h[l] for l in group],
                             [bn2[l, 3] for l in group], [bn2[l, 4] for l in group], [a.gw2 for a in lays], M, H, W, q,
                             unit.stage.wgrad_prec)
        ext.conv_wgrad_group(1, [g2[set_, i] for i in range(n)], [h[l] for l in group], [k[set_, i, 0] for i in range(n)],
                             [k[set_, i, 1] for i in range(n)], [k[set_, i, 2] for i in range(n)],
                             [buf2d[:, :C0 + GROWTH * l] for l in group], [sc1[l] for l in group], [sh1[l] for l in group],
                             [a.gw1 for a in lays], M, H, W, q, unit.stage.wgrad_prec)
        q.flush()

    for l in range(L - 1, -1, -1):
        lay = unit.layers[l]
        C = C0 + GROWTH * l
        sl = slice(C, C + GROWTH)
        set_, slot = gi % 2, len(grp)
        if slot == 0 and side is not None and done[set_] is not None:
            main.wait_event(done[set_])              # the side stream has finished with this buffer set
        sc2, sh2 = bn2[l, 3], bn2[l, 4]
        t3 = t1 = {}
        if tail:
            t3 = dict(acc_out=acc[l, :, :, :MID], tail_counters=tickets[l, 0], tail_mode=0, tail_count=float(M),
                      tail_mean=bn2[l, 0], tail_rstd=bn2[l, 2], tail_scale=sc2, tail_dgamma=lay.gg2, tail_dbeta=lay.gb2,
                      tail_k0=k[set_, slot, 0], tail_k1=k[set_, slot, 1], tail_k2=k[set_, slot, 2])
            t1 = dict(acc_out=acc[l, :, :, MID:MID + C], tail_counters=tickets[l, 1], tail_mode=1, tail_count=float(M),
                      tail_mean=stats[0], tail_rstd=stats[2], tail_scale=sc1[l], tail_dgamma=lay.gg1, tail_dbeta=lay.gb1,
                      tail_k1=nq, tail_k2=np_)
        st = ext.conv_gemm(d2d[:, sl], buf2d[:, sl], ones[:GROWTH], nq[sl], np_[sl], False, M, H, W, 9, True,
                           lay.w2, True, g2[set_, slot], True, False, h[l], sc2, sh2, 2, 0, bf, **t3)
        if not tail:
            ext.bn_bwd_prep(st, float(M), bn2[l, 0], bn2[l, 2], sc2, lay.gg2, lay.gb2,
                            k[set_, slot, 0], k[set_, slot, 1], k[set_, slot, 2], MID, 0)
        st1 = ext.conv_gemm(g2[set_, slot], h[l], k[set_, slot, 0], k[set_, slot, 1], k[set_, slot, 2], False, M, H, W, 1,
                            False, lay.w1, True, d2d[:, :C], True, True, buf2d[:, :C], sc1[l], sh1[l], 2, 0, bf, **t1)
        if not tail:
            ext.bn_bwd_prep(st1, float(M), stats[0], stats[2], sc1[l], lay.gg1, lay.gb1, None, nq, np_, C, 1)
        grp.append(l)
        if len(grp) == G or l == 0:
            if side is not None:
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    launch(grp, set_)
                    done[set_] = torch.cuda.Event()
                    done[set_].record(side)
            else:
                launch(grp, set_)
            grp, gi = [], gi + 1


class _Layer:
    """Parameter / gradient views of one dense layer (all views into the flat arena)."""
    __slots__ = ("g1", "b1", "rm1", "rv1", "w1", "g2", "b2", "rm2", "rv2", "w2",
                 "gg1", "gb1", "gw1", "gg2", "gb2", "gw2")


class BlockUnit:
    def __init__(self, stage: "NativeDenseNetStage", name: str) -> None:
        # ``name`` may carry a layer range (``denseblock3[8:24]``: a pipeline stage boundary inside the block);
        # the unit then owns layers l0..l1-1 and its input is the block's buffer prefix of c_in channels
        self.stage, self.name = stage, name
        self.block, self.l0, self.l1 = parse_unit(name)
        b = int(self.block[-1]) - 1
        c0 = block_channels()[b][0]
        self.num_layers = self.l1 - self.l0
        self.c_in, self.c_out = c0 + GROWTH * self.l0, c0 + GROWTH * self.l1
        self.layers: List[_Layer] = []
        self.out_stats: Optional[torch.Tensor] = None
        self.pending_acc = None             # (accumulators, first channel, count) the consumer still has to finalise
        self.defer_finalize = False         # set by the stage: the next unit is a transition / head of this stage
        self.param_names: List[str] = []

    def bind(self) -> None:
        st = self.stage
        self.layers = []
        prefixes = []
        for i in range(self.l0, self.l1):
            pre = f"features.{self.block}.denselayer{i + 1}."
            prefixes.append(pre)
            lay = _Layer()
            lay.g1, lay.gg1 = st.pv(pre + "norm1.weight")
            lay.b1, lay.gb1 = st.pv(pre + "norm1.bias")
            lay.w1, lay.gw1 = st.pv(pre + "conv1.weight")
            lay.g2, lay.gg2 = st.pv(pre + "norm2.weight")
            lay.b2, lay.gb2 = st.pv(pre + "norm2.bias")
            lay.w2, lay.gw2 = st.pv(pre + "conv2.weight")
            lay.rm1, lay.rv1 = st.bv(pre + "norm1.running_mean"), st.bv(pre + "norm1.running_var")
            lay.rm2, lay.rv2 = st.bv(pre + "norm2.running_mean"), st.bv(pre + "norm2.running_var")
            self.layers.append(lay)
        self.param_names = [n for n in st.param_index if n.startswith(tuple(prefixes))]

    def __call__(self, x: torch.Tensor, in_stats) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        y = _BlockFn.apply(x, self.stage.anchor, self)
        if self.pending_acc is not None:
            return y, (self.out_stats, self.pending_acc)
        return y, self.out_stats


# ======================================================================================
# transition: BN+ReLU -> 2x2 avg-pool -> 1x1 conv  (pool and 1x1 conv commute: 4x fewer GEMM rows)
# ======================================================================================
def _input_stats(ext, buf2d: torch.Tensor, M: int, C: int, H: int, W: int) -> torch.Tensor:
    """Channel statistics of a tensor that did not come with any (a stage that starts here)."""
    stats = torch.empty(3, C, dtype=torch.float32, device=buf2d.device)
    scratch = torch.empty(M, C, dtype=torch.float32, device=buf2d.device)
    part = ext.rows_op(0, M, buf2d, None, None, None, scratch, H, W, True)
    ext.bn_prep(part, 0, C, float(M), stats[0], stats[1], stats[2], None, None, None, None, None, None, C,
                BN_MOMENTUM, BN_EPS, True)
    return stats


class _TransitionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, buf: torch.Tensor, anchor: torch.Tensor, unit: "TransitionUnit", in_stats):
        ext = _ext()
        train = unit.stage.training
        B, H, W, C = buf.shape
        M, Mp, Co = B * H * W, B * (H // 2) * (W // 2), C // 2
        f32 = dict(dtype=torch.float32, device=buf.device)
        buf = buf.contiguous()
        buf2d = buf.view(M, C)
        pend = {}
        if isinstance(in_stats, tuple):      # a dense block that left its last slice in the accumulators
            in_stats, (pacc, pc0, pn) = in_stats
            if train:
                pend = dict(acc=pacc, acc_c0=pc0, acc_n=pn)
        if train and in_stats is None:
            in_stats = _input_stats(ext, buf2d, M, C, H, W)
        stats = in_stats if train else torch.empty(3, C, **f32)
        ss = torch.empty(2, C, **f32)
        ext.bn_prep(None, 0, 0, float(M), stats[0], stats[1], stats[2], unit.g, unit.b, unit.rm, unit.rv,
                    ss[0], ss[1], C, BN_MOMENTUM, BN_EPS, train, **pend)
        pooled = torch.empty(Mp, C, **f32)
        ext.rows_op(1, Mp, pooled, buf2d, ss[0], ss[1], pooled, H, W, False)
        y = torch.empty(B, H // 2, W // 2, Co, **f32)
        ext.conv_gemm(pooled, None, None, None, None, False, Mp, H // 2, W // 2, 1, False, unit.w, False,
                      y.view(Mp, Co), False, False, None, None, None, 0, 0, unit.stage.bf16)
        ctx.unit = unit
        ctx.geom = (B, H, W, C)
        ctx.save_for_backward(buf, stats, ss, pooled)
        return y

    @staticmethod
    def backward(ctx, dy: torch.Tensor):
        ext = _ext()
        unit = ctx.unit
        buf, stats, ss, pooled = ctx.saved_tensors
        B, H, W, C = ctx.geom
        M, Mp, Co = B * H * W, B * (H // 2) * (W // 2), C // 2
        f32 = dict(dtype=torch.float32, device=buf.device)
        dy2d = dy.contiguous().view(Mp, Co)
        buf2d = buf.view(M, C)
        ext.conv_wgrad(dy2d, None, None, None, None, pooled, None, None, False, Mp, H // 2, W // 2, 1,
                       unit.gw, unit.stage.wgrad_min_rows, prec=unit.stage.wgrad_prec)
        dpool = torch.empty(Mp, C, **f32)
        ext.conv_gemm(dy2d, None, None, None, None, False, Mp, H // 2, W // 2, 1, False, unit.w, True,
                      dpool, False, False, None, None, None, 0, 0, unit.stage.bf16)
        dbuf = torch.empty(B, H, W, C, **f32)
        d2d = dbuf.view(M, C)
        st = ext.rows_op(2, M, dpool, buf2d, ss[0], ss[1], d2d, H, W, True)
        k = torch.empty(3, C, **f32)
        ext.bn_bwd_prep(st, float(M), stats[0], stats[2], ss[0], unit.gg, unit.gb, k[0], k[1], k[2], C, 0)
        if unit.defer_affine:       # the dense block in front adopts k1 / k2 as its deferred terms
            unit.stage.set_pending_affine(unit, k[1], k[2])
        else:
            ext.rows_op(4, M, d2d, buf2d, k[1], k[2], d2d, H, W, False)     # dX = sc*g + k1*x + k2
        unit.stage._unit_grads_ready(unit)
        unit.stage._owned_grad_ptr = dbuf.data_ptr()       # fresh tensor, single consumer: may be updated in place
        return dbuf, None, None, None


class TransitionUnit:
    def __init__(self, stage: "NativeDenseNetStage", name: str) -> None:
        self.stage, self.name = stage, name
        self.defer_affine = False           # set by the stage: the unit in front is a dense block of this stage
        self.param_names: List[str] = []

    def bind(self) -> None:
        st, pre = self.stage, f"features.{self.name}."
        self.g, self.gg = st.pv(pre + "norm.weight")
        self.b, self.gb = st.pv(pre + "norm.bias")
        self.w, self.gw = st.pv(pre + "conv.weight")
        self.rm, self.rv = st.bv(pre + "norm.running_mean"), st.bv(pre + "norm.running_var")
        self.param_names = [n for n in st.param_index if n.startswith(pre)]

    def __call__(self, x, in_stats):
        return _TransitionFn.apply(x, self.stage.anchor, self, in_stats), None


# ======================================================================================
# head: norm5 + ReLU + global average pool + Linear(1024 -> classes), all native, ONE autograd node
# ======================================================================================
class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, buf: torch.Tensor, anchor: torch.Tensor, unit: "HeadUnit", in_stats):
        ext = _ext()
        train = unit.stage.training
        B, H, W, C = buf.shape
        M = B * H * W
        f32 = dict(dtype=torch.float32, device=buf.device)
        buf = buf.contiguous()
        buf2d = buf.view(M, C)
        pend = {}
        if isinstance(in_stats, tuple):      # a dense block that left its last slice in the accumulators
            in_stats, (pacc, pc0, pn) = in_stats
            if train:
                pend = dict(acc=pacc, acc_c0=pc0, acc_n=pn)
        if train and in_stats is None:
            in_stats = _input_stats(ext, buf2d, M, C, H, W)
        stats = in_stats if train else torch.empty(3, C, **f32)
        ss = torch.empty(2, C, **f32)
        ext.bn_prep(None, 0, 0, float(M), stats[0], stats[1], stats[2], unit.g, unit.b, unit.rm, unit.rv,
                    ss[0], ss[1], C, BN_MOMENTUM, BN_EPS, train, **pend)
        pooled = ext.bnrelu_gap_fwd(buf2d, B, H * W, ss[0], ss[1])
        logits = ext.linear_fwd(pooled, unit.cw, unit.cb)
        ctx.unit = unit
        ctx.geom = (B, H, W, C)
        ctx.save_for_backward(buf, stats, ss, pooled)
        return logits

    @staticmethod
    def backward(ctx, dlogits: torch.Tensor):
        ext = _ext()
        unit = ctx.unit
        buf, stats, ss, pooled = ctx.saved_tensors
        B, H, W, C = ctx.geom
        M = B * H * W
        f32 = dict(dtype=torch.float32, device=buf.device)
        buf2d = buf.view(M, C)
        dpooled = ext.linear_bwd(dlogits.contiguous(), pooled, unit.cw, unit.gcw, unit.gcb)
        dbuf = torch.empty(B, H, W, C, **f32)
        d2d = dbuf.view(M, C)
        st = ext.rows_op(3, M, dpooled, buf2d, ss[0], ss[1], d2d, H, W, True)
        k = torch.empty(3, C, **f32)
        ext.bn_bwd_prep(st, float(M), stats[0], stats[2], ss[0], unit.gg, unit.gb, k[0], k[1], k[2], C, 0)
        if unit.defer_affine:
            unit.stage.set_pending_affine(unit, k[1], k[2])
        else:
            ext.rows_op(4, M, d2d, buf2d, k[1], k[2], d2d, H, W, False)
        unit.stage._unit_grads_ready(unit)
        unit.stage._owned_grad_ptr = dbuf.data_ptr()
        return dbuf, None, None, None


class HeadUnit:
    def __init__(self, stage: "NativeDenseNetStage") -> None:
        self.stage, self.name = stage, "head"
        self.defer_affine = False
        self.param_names: List[str] = []

    def bind(self) -> None:
        st = self.stage
        self.g, self.gg = st.pv("features.norm5.weight")
        self.b, self.gb = st.pv("features.norm5.bias")
        self.cw, self.gcw = st.pv("classifier.weight")
        self.cb, self.gcb = st.pv("classifier.bias")
        self.rm, self.rv = st.bv("features.norm5.running_mean"), st.bv("features.norm5.running_var")
        self.param_names = ["features.norm5.weight", "features.norm5.bias", "classifier.weight", "classifier.bias"]

    def __call__(self, x, in_stats):
        return _HeadFn.apply(x, self.stage.anchor, self, in_stats), None


# ======================================================================================
# stem: conv0 7x7/2 (fp32 MFMA implicit GEMM over an NHWC4 repack of the image batch, norm0
# statistics in its epilogue) -> BN + ReLU + max-pool 3x3/2 in one pass; backward = pool/ReLU/BN
# masks in one pass + conv0 weight gradient (csrc/stem_kernels.hip).  ONE autograd node.
# ======================================================================================
class _StemFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: torch.Tensor, anchor: torch.Tensor, unit: "StemUnit"):
        ext = _ext()
        train = unit.stage.training
        pooled, xp, y0, stats, ss, idx = ext.stem_fwd(x.contiguous(), unit.w, unit.g, unit.b, unit.rm, unit.rv,
                                                      train, BN_MOMENTUM, BN_EPS)
        ctx.unit = unit
        ctx.save_for_backward(xp, y0, stats, ss, idx)
        return pooled

    @staticmethod
    def backward(ctx, dp: torch.Tensor):
        unit = ctx.unit
        xp, y0, stats, ss, idx = ctx.saved_tensors
        _ext().stem_bwd(dp.contiguous(), xp, y0, stats, ss, idx, unit.gw, unit.gg, unit.gb)
        unit.stage._unit_grads_ready(unit)
        return None, None, None


class StemUnit:
    def __init__(self, stage: "NativeDenseNetStage") -> None:
        self.stage, self.name = stage, "stem"
        self.param_names: List[str] = []

    def bind(self) -> None:
        st = self.stage
        self.w, self.gw = st.pv("features.conv0.weight")          # torch layout [64,3,7,7]
        self.g, self.gg = st.pv("features.norm0.weight")
        self.b, self.gb = st.pv("features.norm0.bias")
        self.rm, self.rv = st.bv("features.norm0.running_mean"), st.bv("features.norm0.running_var")
        self.param_names = ["features.conv0.weight", "features.norm0.weight", "features.norm0.bias"]

    def __call__(self, x, in_stats):
        if x.dtype not in (torch.uint8, torch.float32):
            x = x.float()
        return _StemFn.apply(x, self.stage.anchor, self), None


# ======================================================================================
# the stage module
# ======================================================================================
class NativeDenseNetStage(nn.Module):
    """Drop-in for ``DenseNetStage`` on a GPU.  Input: uint8/float NCHW images (first stage) or an
    NHWC activation (later stages); output: NHWC activation, or logits on the last stage."""

    def __init__(self, units: Sequence[str] = UNIT_NAMES, num_classes: int = 5, device="cuda",
                 dtype: str = "fp32", bucket_mb: float = 8.0, dp_world: int = 1,
                 wgrad_min_rows: int = 64, grad_phases: Sequence[str] = ()) -> None:
        super().__init__()
        if dtype not in ("fp32", "bf16", "fp32x3"):
            raise ValueError(f"dtype must be fp32, bf16 or fp32x3, got {dtype!r}")
        # bf16: the dense-block / transition forward and data-gradient GEMMs feed the bf16 MFMA (operands
        # rounded at the LDS write, fp32 accumulate); storage, BN statistics, weight gradients, stem, head
        # and Adam stay fp32.  Not the reference's numerics -- never the default, never the headline.
        # fp32x3 (experiment, README "dead ends"): fp32 operands split into three bf16 pieces, six bf16 MFMAs per product
        # in the same GEMMs - fp32-level accuracy (bench/split_accuracy.py), but the staging work triples and the
        # kernels are staging-bound, so it only wins on the block-1 1x1 forward.
        # `bf16` is the matrix-core path selector handed to the kernels: 0 fp32 MFMA, 1 bf16, 3 split.
        self.bf16 = {"fp32": 0, "bf16": 1, "fp32x3": 3}[dtype]
        # weight gradients: bf16 MFMA (transposing LDS reads, csrc/conv_wgrad.hip) in bf16 mode only; DDL_WGRAD_PREC is the
        # A/B switch of the experiment "how much of the fp32 step do the weight gradients cost" (profiles/r3/bf16_wgrad)
        self.wgrad_prec = int(knob("DDL_WGRAD_PREC", "1" if self.bf16 == 1 else "0"))
        # Fused BN statistics (csrc/conv_params.h FusedBn / FusedK): forward on by default (2 bn_prep launches
        # per dense layer leave the single-stream forward chain: 10.41 -> 9.92 ms/step at batch 30).  The
        # backward variant (BN2 coefficients derived inside the 1x1 dgrad / wgrad) removes 58 more launches
        # but its prologue work outweighs them once the two backward streams overlap inside a hipGraph
        # (10.23 ms), so it is opt-in: DDL_FUSED_BWD=1 (it does win with eager launches).
        self.fused_stats = knob("DDL_FUSED_STATS", "1") == "1"
        self.fused_bwd = knob("DDL_FUSED_BWD", "0") == "1"
        # g2 / k buffers of the dense-block backward: how many layers the weight-gradient stream may lag
        self.g2_bufs = max(1, int(knob("DDL_G2_BUFS", "2")))
        # slab reductions of the weight gradients (csrc/conv_bindings.cpp SlabQueue): "layer" = one launch per dense
        # layer for its 3x3 + 1x1 slabs (58 fewer graph nodes than one per kernel); "block" = one per dense block
        # (116 fewer nodes, but measured 0.4 ms/step SLOWER at batch 30: the slabs of 24 layers no longer fit the
        # 256 MB Infinity Cache and come back from HBM); "0" = behind every kernel
        d = knob("DDL_DEFER_SLABS", "layer")
        self.defer_slab_reduce = d if d in ("layer", "block") else ""
        ops.require_ext()
        device = torch.device(device)
        if device.type != "cuda":
            raise ValueError("NativeDenseNetStage needs a GPU")
        self.units = tuple(units)
        self.num_classes = num_classes
        self.wgrad_min_rows = wgrad_min_rows
        ref = DenseNetStage(self.units, num_classes)          # same init distribution as the reference
        self.features = ref.features
        if "head" in self.units:
            self.classifier = ref.classifier
        # re-lay-out conv weights for the MFMA kernels
        for name, mod in self.named_modules():
            if isinstance(mod, nn.Conv2d) and name != "features.conv0":
                mod.weight = nn.Parameter(self._to_native(mod.weight.data))
        self.to(device)
        # backward phases: ``grad_phases`` names the units that open a new phase (forward order).
        # Each phase is one contiguous run of units; gradient buckets are cut at phase boundaries.
        self.phase_starts: List[int] = sorted({self.units.index(u) for u in grad_phases if u in self.units} - {0})
        rev = reverse_named_parameters(self)
        breaks = []
        for start in self.phase_starts:      # first parameter (in backward order) of the unit before the cut
            prior = set(self.units[:start])
            for n, _ in rev:
                if self.unit_of_param(n) in prior:
                    breaks.append(n)
                    break
        self.arena = FlatArena(rev, device=device,
                               segment_numel=max(int(bucket_mb * 2 ** 20 / 4), 1),
                               segment_multiple=max(dp_world, 1) * 64, break_before=breaks)
        self.flat_buffers = FlatBuffers(self, device)
        self.param_index: Dict[str, int] = {n: i for i, n in enumerate(self.arena.names)}
        self._buffers_by_name = dict(self.named_buffers())
        self.anchor = torch.zeros(1, device=device, requires_grad=True)   # keeps every unit in the graph
        self.side_stream = new_stream(device)
        self.overlap_wgrad = True
        self._wgrad_group = knob("DDL_WGRAD_GROUP", "auto")
        self.tail_prep = knob("DDL_TAIL_PREP", "1") == "1"      # csrc/conv_params.h TailPrep (grouped backward)
        self._fork_origin = None          # origin stream of the open side-stream fork (see ``fork``)
        self._owned_grad_ptr = 0          # data_ptr of the gradient a transition / head backward just produced
        self.ones = torch.ones(1024, device=device)
        self._grad_ready_cb = None
        self.unit_objs = []
        for u in self.units:
            if u == "stem":
                self.unit_objs.append(StemUnit(self))
            elif u.startswith("denseblock"):
                self.unit_objs.append(BlockUnit(self, u))
            elif u.startswith("transition"):
                self.unit_objs.append(TransitionUnit(self, u))
            else:
                self.unit_objs.append(HeadUnit(self))
        for a, b in zip(self.unit_objs[:-1], self.unit_objs[1:]):
            if isinstance(a, BlockUnit) and isinstance(b, (TransitionUnit, HeadUnit)):
                a.defer_finalize = True         # b's bn_prep finalises a's last channel slice (one launch fewer)
                b.defer_affine = knob("DDL_DEFER_AFFINE", "1") == "1"   # a adopts b's BN-backward k1 / k2
                a.next_unit = b
        self._pending_affine = None
        # gradient lanes (parallel/pp.py "lanes"): lane 0 is the arena's own gradient buffer; further lanes are
        # buffers of the same layout that concurrently running micro-batch backwards accumulate into.  A unit's
        # views are rebound per lane (cached: rebinding happens at graph-capture time and in eager / --debug-sync
        # steps only, never on the replay path).
        self._grad_lanes: List[torch.Tensor] = [self.arena.flat_grad]
        self._grad_lane = 0
        # buffer lanes (forward lanes): lane 0 is the real running statistics; lane 1 + i is micro-batch i's scratch
        # copy - forwards that run concurrently leave `momentum * statistic` there (the kernels' update applied to
        # zeros) and the running statistics are advanced in micro-batch order afterwards (``combine_buffer_lanes``)
        self._buf_lanes: List[torch.Tensor] = [self.flat_buffers.flat_float]
        self._buf_lane = 0
        self.defer_counters = False       # num_batches_tracked advanced by combine_buffer_lanes instead of per forward
        self._mb_forwarded = set()
        for u in self.unit_objs:
            before = dict(vars(u))
            u.bind()
            u._bind_keys = [k for k, v in vars(u).items() if k not in before or before[k] is not v]
            u._lane_cache = {(0, 0): {k: getattr(u, k) for k in u._bind_keys}}

    def unit_of_param(self, fqn: str) -> str:
        """Unit of this stage that owns a parameter (torchvision FQN)."""
        if fqn.startswith("classifier.") or fqn.startswith("features.norm5."):
            return "head"
        parts = fqn.split(".")
        part = parts[1]
        if part in ("conv0", "norm0"):
            return "stem"
        if part.startswith("denseblock"):
            layer = int(parts[2][len("denselayer"):]) - 1
            for u in self.units:
                base, l0, l1 = parse_unit(u)
                if base == part and l0 <= layer < l1:
                    return u
        return part

    # ---- layout conversion (checkpoint boundary) -------------------------------------------
    @staticmethod
    def _to_native(w: torch.Tensor) -> torch.Tensor:
        n, c, kh, kw = w.shape
        return w.permute(0, 2, 3, 1).reshape(n, kh * kw * c).contiguous()

    def _is_native_conv(self, fqn: str) -> bool:
        return fqn.endswith(("conv1.weight", "conv2.weight", "conv.weight")) and "conv0" not in fqn

    def ddl_export_tensor(self, fqn: str, t: torch.Tensor) -> torch.Tensor:
        if not self._is_native_conv(fqn) or t.dim() != 2:
            return t
        k = 3 if fqn.endswith("conv2.weight") else 1
        n, kc = t.shape
        return t.reshape(n, k, k, kc // (k * k)).permute(0, 3, 1, 2).contiguous()

    def ddl_import_tensor(self, fqn: str, t: torch.Tensor) -> torch.Tensor:
        if not self._is_native_conv(fqn) or t.dim() != 4:
            return t
        return self._to_native(t)

    def export_state_dict(self) -> Dict[str, torch.Tensor]:
        """State-dict in the torchvision layout (loadable by ``DenseNetStage``)."""
        out = {}
        for k, v in self.state_dict().items():
            out[k] = self.ddl_export_tensor(k, v.detach()).clone()
        return out

    @torch.no_grad()
    def import_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        for k, p in self.named_parameters():
            p.data.copy_(self.ddl_import_tensor(k, sd[k]))
        for k, b in self.named_buffers():
            b.data.copy_(sd[k])

    # ---- views ---------------------------------------------------------------------------
    def pv(self, fqn: str) -> Tuple[torch.Tensor, torch.Tensor]:
        i = self.param_index[fqn]
        return self.arena.param_view(i), self.arena.view_in(self._grad_lanes[self._grad_lane], i)

    # ---- gradient lanes ------------------------------------------------------------------
    def ensure_grad_lanes(self, n: int) -> None:
        if self._grad_lanes[0] is not self.arena.flat_grad:
            raise RuntimeError("the arena's gradient buffer was replaced after the stage was built")
        while len(self._grad_lanes) < n:
            self._grad_lanes.append(torch.zeros_like(self.arena.flat_grad))

    def set_grad_lane(self, k: int) -> None:
        """Kernels launched (or captured) from now on accumulate parameter gradients into gradient lane ``k``."""
        if not 0 <= k < len(self._grad_lanes):
            raise ValueError(f"gradient lane {k} of {len(self._grad_lanes)}")
        self._set_lanes(k, self._buf_lane)

    def _set_lanes(self, g: int, b: int) -> None:
        if (g, b) == (self._grad_lane, self._buf_lane):
            return
        self._grad_lane, self._buf_lane = g, b
        for u in self.unit_objs:
            cached = u._lane_cache.get((g, b))
            if cached is None:
                u.bind()
                cached = u._lane_cache[(g, b)] = {key: getattr(u, key) for key in u._bind_keys}
            else:
                for key, val in cached.items():
                    setattr(u, key, val)

    def sum_grad_lanes(self) -> None:
        """arena gradient += lane 1 + lane 2 + ... (fixed order), lanes cleared for the next step; back on lane 0."""
        for lane in self._grad_lanes[1:]:
            self.arena.flat_grad.add_(lane)
            lane.zero_()
        self.set_grad_lane(0)

    # ---- buffer lanes (concurrent forwards) ---------------------------------------------------
    def ensure_buffer_lanes(self, n_microbatches: int) -> None:
        while len(self._buf_lanes) < 1 + n_microbatches:
            self._buf_lanes.append(torch.zeros_like(self.flat_buffers.flat_float))
        self.defer_counters = True

    def begin_microbatch_buffers(self, i: int) -> None:
        """Training forward of micro-batch ``i`` is next (eager, or being captured): its BN kernels update scratch
        copy ``i`` (cleared here - inside the graph when capturing) instead of the running statistics."""
        self._buf_lanes[1 + i].zero_()
        self._set_lanes(self._grad_lane, 1 + i)
        self._mb_forwarded.add(i)

    def note_microbatch_forward(self, i: int) -> None:
        """Replay path: micro-batch ``i``'s captured forward was launched (its scratch copy will hold an update)."""
        self._mb_forwarded.add(i)

    def combine_buffer_lanes(self) -> None:
        """running = (1 - momentum) * running + scratch_i, in micro-batch order, for the forwards of this step; the
        num_batches_tracked counters advance by their number.  Runs on the compute stream after the lanes joined."""
        ff = self.flat_buffers.flat_float
        keep = float(torch.tensor(1.0) - torch.tensor(BN_MOMENTUM, dtype=torch.float32))    # the kernels' (1.f - momentum)
        for i in sorted(self._mb_forwarded):
            ff.mul_(keep).add_(self._buf_lanes[1 + i])
        if self._mb_forwarded and self.flat_buffers.flat_int.numel():
            self.flat_buffers.flat_int.add_(len(self._mb_forwarded))
        self._mb_forwarded = set()
        self._set_lanes(self._grad_lane, 0)

    def bv(self, fqn: str) -> torch.Tensor:
        if self._buf_lane == 0:
            return self._buffers_by_name[fqn]
        off, n, shape = self.flat_buffers.float_slots[fqn]
        return self._buf_lanes[self._buf_lane][off:off + n].view(shape)

    # ---- BN backward of a transition / head handed to the dense block in front of it ---------------------------
    def set_pending_affine(self, producer, k1: torch.Tensor, k2: torch.Tensor) -> None:
        if self._pending_affine is not None:
            raise RuntimeError("a deferred BN-backward term was never consumed (backward of a dense block skipped?)")
        self._pending_affine = (producer, k1, k2)

    def take_pending_affine(self, block):
        """(k1, k2) left by the unit behind ``block`` in this backward pass, or None.  The hand-off is by unit
        adjacency: the gradient that reaches ``block`` IS the producer's returned tensor (or a copy of it) - the
        block's output has no other consumer inside a stage."""
        p, self._pending_affine = self._pending_affine, None
        if p is None:
            return None
        if p[0] is not getattr(block, "next_unit", None):
            raise RuntimeError(f"deferred BN-backward term of {p[0].name} reached {block.name}")
        return p[1], p[2]

    def wgrad_group_size(self, M: int) -> int:
        """Dense layers whose weight gradients share one launch (``_backward_grouped``).  ``DDL_WGRAD_GROUP``: an
        integer, or ``auto`` - group on the small feature maps, where a single layer's weight gradient is a
        latency-bound 10-25 us kernel; keep the per-layer kernels where one layer already fills the chip."""
        g = self._wgrad_group
        if g == "auto":
            for rule in knob("DDL_WGRAD_GROUPS", "").split(","):     # "minM:G,..." first match wins (experiments)
                if rule:
                    m, gg = rule.split(":")
                    if M >= int(m):
                        return max(1, min(int(gg), 24))
            return 4
        return max(1, min(int(g), 24))

    # ---- side-stream discipline ------------------------------------------------------------
    # Ending a HIP stream capture whose nested fork rejoins the origin only THROUGH an intermediate stream
    # crashes the runtime (SIGSEGV in capture_end, round 1).  Rule: a side stream is forked from the origin
    # stream only, and joined into the origin directly.  ``fork`` / ``join`` enforce it; ``assert_joined`` is
    # called by the capture code before a capture ends.
    def fork(self, origin: "torch.cuda.Stream") -> "torch.cuda.Stream":
        if self._fork_origin is not None:
            raise RuntimeError("nested stream fork: the weight-gradient side stream must be forked from the origin "
                               "stream and joined into it directly (a nested fork crashes hipStreamEndCapture)")
        if origin == self.side_stream:
            # torch hands streams out of a pool of 32, round-robin: every 32nd stream drawn in a process is the one
            # torch.cuda.graph captures on (its class-level default capture stream), so a stage built late in a
            # long-lived process can find its side stream IS the stream it runs on.  Draw another one.
            self.side_stream = new_stream(origin.device, avoid=[origin])
        self._fork_origin = origin
        return self.side_stream

    def join(self, origin: "torch.cuda.Stream", side: "torch.cuda.Stream") -> None:
        if self._fork_origin is None or origin != self._fork_origin or side != self.side_stream:
            raise RuntimeError("stream join does not match the open fork: join the side stream into its origin directly")
        origin.wait_stream(side)
        self._fork_origin = None

    def assert_joined(self) -> None:
        if self._fork_origin is not None:
            self._fork_origin = None
            raise RuntimeError("a forked side stream was not joined into its origin stream before the capture ended")

    # ---- DP integration ------------------------------------------------------------------
    def set_grad_ready_callback(self, cb) -> None:
        """``cb(slot_index)`` is called when a natively computed parameter gradient is final."""
        self._grad_ready_cb = cb

    def _unit_grads_ready(self, unit) -> None:
        if self._grad_ready_cb is not None:
            for n in unit.param_names:
                self._grad_ready_cb(self.param_index[n])

    # ---- forward -------------------------------------------------------------------------
    @property
    def is_first(self) -> bool:
        return self.units[0] == "stem"

    @property
    def is_last(self) -> bool:
        return self.units[-1] == "head"

    def forward_phased(self, x: torch.Tensor):
        """Forward with the autograd graph cut at the phase boundaries: returns ``(out, cuts)`` where
        ``cuts[i] = (y, y_leaf)`` - ``y`` is the boundary activation as produced, ``y_leaf`` the detached
        leaf the following units consumed.  ``out.backward()`` stops at the last cut; then
        ``y.backward(y_leaf.grad)`` for the cuts in reverse order runs the earlier phases - each call
        can be its own hipGraph, with that phase's gradient buckets handed to RCCL in between."""
        stats, cuts = None, []
        self._pending_affine = None
        for i, u in enumerate(self.unit_objs):
            if i in self.phase_starts:
                leaf = x.detach().requires_grad_(True)
                cuts.append((x, leaf))
                x = leaf
            x, stats = u(x, stats)
        if self.training:
            self._bump_counters()
        return x, cuts

    def phase_param_names(self) -> List[List[str]]:
        """Parameter names per backward phase, in execution order (last units first)."""
        bounds = [0] + self.phase_starts + [len(self.units)]
        out = []
        for a, b in reversed(list(zip(bounds[:-1], bounds[1:]))):
            units = set(self.units[a:b])
            out.append([n for n in self.arena.names if self.unit_of_param(n) in units])
        return out

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        stats = None
        self._pending_affine = None          # a backward pass that died half-way must not leak into the next one
        for u in self.unit_objs:
            x, stats = u(x, stats)
        if self.training:
            # num_batches_tracked of every BN layer (+1 per forward, one launch over the flat buffer)
            self._bump_counters()
        return x

    def _bump_counters(self) -> None:
        fi = self.flat_buffers.flat_int
        if fi.numel() == 0 or (self.defer_counters and self._buf_lane != 0):
            return
        fi.add_(1)

    def output_shape(self, in_shape):
        """NHWC output shape for an NCHW (first stage) / NHWC input shape."""
        if self.is_first:
            s = tuple(in_shape)
        else:
            s = (in_shape[0], in_shape[3], in_shape[1], in_shape[2])
        for u in self.units:
            s = DenseNetStage.unit_output_shape(u, s)
        if self.is_last:
            return (s[0], self.num_classes)
        return (s[0], s[2], s[3], s[1])
