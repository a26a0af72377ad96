"""Direct tests of the hot kernels at DenseNet-121's REAL shapes for the benchmark batch (B=30 @ 224x224):
M = 94080 / 23520 / 5880 / 1470 pixels, C up to 992 (not a multiple of the 128-column tiles), row counts that do
not divide by ``rows_per_wg``.  Every kernel is compared with an fp64 PyTorch evaluation of the same formula.
(the weight-gradient kernels - 30 % of the kernel time - had no direct test, nothing ran at the
benchmark shape.)  Plus: the whole model forward + backward at B=30 @ 224 and a 50-step loss curve against the
PyTorch oracle on the same data."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

MID, GROWTH = 128, 32


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from ddl import ops
    ops.require_ext()
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    return torch.device("cuda", 0)


def _rand(*shape, dev, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev)


def _close(got, ref, l2=2e-6, mx=2e-5):
    got, ref = got.double(), ref.double()
    assert float((got - ref).norm() / ref.norm().clamp_min(1e-30)) < l2
    assert float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)) < mx


# the (M, C, C_final) triples of the 1x1 convolutions of blocks 1-4 at B=30: first, a middle and the last layer
WGRAD1 = [(94080, 64, 256), (94080, 224, 256), (23520, 128, 512), (23520, 480, 512), (5880, 256, 1024),
          (5880, 640, 1024), (5880, 992, 1024), (1470, 512, 1024), (1470, 768, 1024), (1470, 992, 1024)]


@pytest.mark.parametrize("M,C,Cf", WGRAD1)
def test_wgrad1x1_real_shapes(dev, M, C, Cf):
    """dW1[128, C] += dh^T relu(bn1(x[:, :C])) with dh = k0*g2 + k1*h + k2 produced by the two-source prologue and
    the activation re-normalised on the fly from the block buffer's channel prefix."""
    from ddl import ops
    ext = ops.require_ext()
    g2, h = _rand(M, MID, dev=dev, seed=1, scale=0.02), _rand(M, MID, dev=dev, seed=2)
    k0, k1, k2 = (_rand(MID, dev=dev, seed=3).abs() + 0.5, _rand(MID, dev=dev, seed=4, scale=0.01),
                  _rand(MID, dev=dev, seed=5, scale=0.01))
    buf = _rand(M, Cf, dev=dev, seed=6)
    sc, sh = _rand(Cf, dev=dev, seed=7).abs() + 0.5, _rand(Cf, dev=dev, seed=8)
    dW = _rand(MID, C, dev=dev, seed=9, scale=0.1)
    dW0 = dW.clone()
    ext.conv_wgrad(g2, h, k0, k1, k2, buf[:, :C], sc[:C].contiguous(), sh[:C].contiguous(), True, M, 1, 1, 1, dW, 64)
    dh = k0.double() * g2.double() + k1.double() * h.double() + k2.double()
    a = torch.relu(buf[:, :C].double() * sc[:C].double() + sh[:C].double())
    ref = dh.t() @ a
    _close(dW.double() - dW0.double(), ref, l2=3e-6, mx=3e-5)


@pytest.mark.parametrize("M,C,Cf", [(5880, 640, 1024), (1470, 992, 1024), (94080, 96, 256)])
def test_wgrad1x1_fused_k_real_shapes(dev, M, C, Cf):
    """The same weight gradient with the BN2-backward coefficients derived inside the kernel from the (sum g,
    sum g*h) fixed-point accumulators (DDL_FUSED_BWD path) - including upstream gradients scaled by 1e-6, where
    the accumulators need their own fixed-point scale (2^-28 quanta rounded such sums away)."""
    from ddl import ops
    ext = ops.require_ext()
    R, S = ext.STAT_REPLICAS, 2.0 ** 48
    for gscale in (0.02, 2e-8):
        g2, h = _rand(M, MID, dev=dev, seed=1, scale=gscale), _rand(M, MID, dev=dev, seed=2)
        mean, rstd = h.double().mean(0), 1.0 / torch.sqrt(h.double().var(0, unbiased=False) + 1e-5)
        gamma = _rand(MID, dev=dev, seed=3).abs() + 0.5
        scale = (gamma.double() * rstd).float()
        acc = torch.zeros(R, 2, MID, dtype=torch.int64, device=dev)
        rows = torch.arange(M, device=dev) % R
        for r in range(R):
            sel = rows == r
            acc[r, 0] = torch.round(g2.double()[sel].sum(0) * S).long()
            acc[r, 1] = torch.round((g2.double()[sel] * h.double()[sel]).sum(0) * S).long()
        buf = _rand(M, Cf, dev=dev, seed=6)
        sc, sh = _rand(Cf, dev=dev, seed=7).abs() + 0.5, _rand(Cf, dev=dev, seed=8)
        dW = torch.zeros(MID, C, device=dev)
        ext.conv_wgrad(g2, h, None, None, None, buf[:, :C], sc[:C].contiguous(), sh[:C].contiguous(), True, M, 1, 1, 1,
                       dW, 64, k_acc=acc, k_mean=mean.float(), k_rstd=rstd.float(), k_scale=scale, k_count=float(M))
        sg, sgh = g2.double().sum(0), (g2.double() * h.double()).sum(0)
        dgam = rstd.float().double() * (sgh - mean.float().double() * sg)
        k0 = scale.double()
        k1 = -k0 * (dgam / M) * rstd.float().double()
        k2 = -k0 * (sg / M) + k0 * (dgam / M) * rstd.float().double() * mean.float().double()
        dh = k0 * g2.double() + k1 * h.double() + k2
        a = torch.relu(buf[:, :C].double() * sc[:C].double() + sh[:C].double())
        _close(dW, dh.t() @ a, l2=2e-5, mx=2e-4)


def _wgrad3x3_ref(dy64, a64, B, H, W):
    """dW[n, tap, c] = sum_m dy[m, n] * a[m shifted by tap, c] (zero padding), fp64, as 9 GEMMs."""
    N, C = dy64.shape[1], a64.shape[1]
    ap = F.pad(a64.view(B, H, W, C), (0, 0, 1, 1, 1, 1))
    out = torch.empty(N, 9, C, dtype=torch.float64, device=dy64.device)
    for ky in range(3):
        for kx in range(3):
            sh = ap[:, ky:ky + H, kx:kx + W, :].reshape(-1, C)
            out[:, ky * 3 + kx, :] = dy64.t() @ sh
    return out.reshape(N, 9 * C)


@pytest.mark.parametrize("B,H,W", [(30, 56, 56), (30, 28, 28), (30, 14, 14), (30, 7, 7), (7, 14, 14), (6, 7, 7)])
def test_wgrad3x3_real_shapes(dev, B, H, W):
    """dW2[32, 9*128] += dy^T (*) relu(bn2(h)) with the deferred dense-block terms applied on the fly
    (dy = dBuf + nq*x + np on a 32-channel slice of 1024-wide buffers)."""
    from ddl import ops
    ext = ops.require_ext()
    M, Cf = B * H * W, 512
    dbuf, xbuf = _rand(M, Cf, dev=dev, seed=1, scale=0.02), _rand(M, Cf, dev=dev, seed=2)
    sl = slice(224, 256)
    nq, np_ = _rand(GROWTH, dev=dev, seed=3, scale=0.01), _rand(GROWTH, dev=dev, seed=4, scale=0.01)
    ones = torch.ones(GROWTH, device=dev)
    h = _rand(M, MID, dev=dev, seed=5)
    sc2, sh2 = _rand(MID, dev=dev, seed=6).abs() + 0.5, _rand(MID, dev=dev, seed=7)
    dW = _rand(GROWTH, 9 * MID, dev=dev, seed=8, scale=0.1)
    dW0 = dW.clone()
    ext.conv_wgrad(dbuf[:, sl], xbuf[:, sl], ones, nq, np_, h, sc2, sh2, True, M, H, W, 9, dW, 64)
    dy = dbuf[:, sl].double() + nq.double() * xbuf[:, sl].double() + np_.double()
    a = torch.relu(h.double() * sc2.double() + sh2.double())
    _close(dW.double() - dW0.double(), _wgrad3x3_ref(dy, a, B, H, W), l2=3e-6, mx=3e-5)


def test_transition_wgrad_real_shape(dev):
    """transition conv weight gradient: plain operands (no prologue), N = 128 / 512 output channels."""
    from ddl import ops
    ext = ops.require_ext()
    for Mp, C, Co in ((23520, 256, 128), (1470, 1024, 512)):
        dy, pooled = _rand(Mp, Co, dev=dev, seed=1, scale=0.02), _rand(Mp, C, dev=dev, seed=2)
        dW = torch.zeros(Co, C, device=dev)
        ext.conv_wgrad(dy, None, None, None, None, pooled, None, None, False, Mp, 1, 1, 1, dW, 64)
        _close(dW, dy.double().t() @ pooled.double(), l2=3e-6, mx=3e-5)


# ---------------------------------------------------------------- bf16 weight gradients (`--dtype bf16`)
def _bf16(t):
    return t.to(torch.bfloat16).double()


def _close_bf16(got, ref):
    """Operands are rounded to bf16 AFTER the fp32 prologue, products accumulate in fp32.  A prologue value within one
    fp32 ulp of a rounding boundary may round the other way in the kernel (fma contraction): judge the tensor in L2
    and bound single outliers."""
    got, ref = got.double(), ref.double()
    assert float((got - ref).norm() / ref.norm().clamp_min(1e-30)) < 2e-4
    assert float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)) < 5e-3


@pytest.mark.parametrize("M,C,Cf", WGRAD1 + [(700, 160, 256), (37, 96, 256)])
def test_wgrad1x1_bf16_real_shapes(dev, M, C, Cf):
    """The 1x1 weight gradient on the bf16 matrix cores (pixel-major bf16 LDS tiles read through the transposing
    ds_read_b64_tr_b16; every tile width SUB = 1 / 3 / 4 and a ragged last pixel step) against an fp64 GEMM of the
    bf16-rounded operands."""
    from ddl import ops
    ext = ops.require_ext()
    g2, h = _rand(M, MID, dev=dev, seed=1, scale=0.02), _rand(M, MID, dev=dev, seed=2)
    k0, k1, k2 = (_rand(MID, dev=dev, seed=3).abs() + 0.5, _rand(MID, dev=dev, seed=4, scale=0.01),
                  _rand(MID, dev=dev, seed=5, scale=0.01))
    buf = _rand(M, Cf, dev=dev, seed=6)
    sc, sh = _rand(Cf, dev=dev, seed=7).abs() + 0.5, _rand(Cf, dev=dev, seed=8)
    dh = _bf16(k0 * g2 + k1 * h + k2)
    a = _bf16(torch.relu(buf[:, :C] * sc[:C] + sh[:C]))
    ref = dh.t() @ a
    for sub in (0, 1, 3, 4):
        dW = _rand(MID, C, dev=dev, seed=9, scale=0.1)
        dW0 = dW.clone()
        ext.conv_wgrad(g2, h, k0, k1, k2, buf[:, :C], sc[:C].contiguous(), sh[:C].contiguous(), True, M, 1, 1, 1, dW, 64,
                       force_sub=sub, prec=1)
        _close_bf16(dW.double() - dW0.double(), ref)


@pytest.mark.parametrize("B,H,W", [(30, 56, 56), (30, 14, 14), (7, 14, 14), (6, 7, 7), (3, 5, 9)])
def test_wgrad3x3_bf16_real_shapes(dev, B, H, W):
    """3x3 weight gradient, bf16 MFMA: the border masks live in three per-tap copies of the dO tile."""
    from ddl import ops
    ext = ops.require_ext()
    M, Cf = B * H * W, 512
    dbuf, xbuf = _rand(M, Cf, dev=dev, seed=1, scale=0.02), _rand(M, Cf, dev=dev, seed=2)
    sl = slice(224, 256)
    nq, np_ = _rand(GROWTH, dev=dev, seed=3, scale=0.01), _rand(GROWTH, dev=dev, seed=4, scale=0.01)
    ones = torch.ones(GROWTH, device=dev)
    h = _rand(M, MID, dev=dev, seed=5)
    sc2, sh2 = _rand(MID, dev=dev, seed=6).abs() + 0.5, _rand(MID, dev=dev, seed=7)
    dW = torch.zeros(GROWTH, 9 * MID, device=dev)
    ext.conv_wgrad(dbuf[:, sl], xbuf[:, sl], ones, nq, np_, h, sc2, sh2, True, M, H, W, 9, dW, 64, prec=1)
    dy = _bf16(ones * dbuf[:, sl] + nq * xbuf[:, sl] + np_)
    a = _bf16(torch.relu(h * sc2 + sh2))
    _close_bf16(dW, _wgrad3x3_ref(dy, a, B, H, W))


def test_wgrad_group_and_transition_bf16(dev):
    """The layer-grouped launch (dense blocks 3 / 4) and the transition's plain weight gradient on the bf16 MFMA."""
    from ddl import ops
    ext = ops.require_ext()
    B, H, W, C0, n = 6, 7, 7, 512, 5
    M, Cf = B * H * W, C0 + GROWTH * n
    dbuf, xbuf = _rand(M, Cf, dev=dev, seed=1, scale=0.02), _rand(M, Cf, dev=dev, seed=2)
    nq, np_ = _rand(Cf, dev=dev, seed=3, scale=0.01), _rand(Cf, dev=dev, seed=4, scale=0.01)
    ones = torch.ones(GROWTH, device=dev)
    hs = [_rand(M, MID, dev=dev, seed=10 + i) for i in range(n)]
    g2 = [_rand(M, MID, dev=dev, seed=20 + i, scale=0.02) for i in range(n)]
    ks = [[_rand(MID, dev=dev, seed=30 + 3 * i + j, scale=(1.0 if j == 0 else 0.01)) for j in range(3)] for i in range(n)]
    sc2, sh2 = [_rand(MID, dev=dev, seed=50 + i).abs() + 0.5 for i in range(n)], [_rand(MID, dev=dev, seed=60 + i) for i in range(n)]
    sc1, sh1 = [_rand(Cf, dev=dev, seed=70 + i).abs() + 0.5 for i in range(n)], [_rand(Cf, dev=dev, seed=80 + i) for i in range(n)]
    sls = [slice(C0 + GROWTH * i, C0 + GROWTH * (i + 1)) for i in range(n)]
    dW2 = [torch.zeros(GROWTH, 9 * MID, device=dev) for _ in range(n)]
    dW1 = [torch.zeros(MID, C0 + GROWTH * i, device=dev) for i in range(n)]
    q = ext.SlabQueue()
    ext.conv_wgrad_group(9, [dbuf[:, s] for s in sls], [xbuf[:, s] for s in sls], [ones] * n, [nq[s] for s in sls],
                         [np_[s] for s in sls], hs, sc2, sh2, dW2, M, H, W, q, 1)
    ext.conv_wgrad_group(1, g2, hs, [k[0] for k in ks], [k[1] for k in ks], [k[2] for k in ks],
                         [xbuf[:, :C0 + GROWTH * i] for i in range(n)], [sc1[i][:C0 + GROWTH * i].contiguous() for i in range(n)],
                         [sh1[i][:C0 + GROWTH * i].contiguous() for i in range(n)], dW1, M, H, W, q, 1)
    q.flush()
    for i in range(n):
        dy = _bf16(ones * dbuf[:, sls[i]] + nq[sls[i]] * xbuf[:, sls[i]] + np_[sls[i]])
        a = _bf16(torch.relu(hs[i] * sc2[i] + sh2[i]))
        _close_bf16(dW2[i], _wgrad3x3_ref(dy, a, B, H, W))
        C = C0 + GROWTH * i
        dh = _bf16(ks[i][0] * g2[i] + ks[i][1] * hs[i] + ks[i][2])
        a1 = _bf16(torch.relu(xbuf[:, :C] * sc1[i][:C] + sh1[i][:C]))
        _close_bf16(dW1[i], dh.t() @ a1)
    for Mp, C, Co in ((23520, 256, 128), (1470, 1024, 512)):
        dy, pooled = _rand(Mp, Co, dev=dev, seed=1, scale=0.02), _rand(Mp, C, dev=dev, seed=2)
        dW = torch.zeros(Co, C, device=dev)
        ext.conv_wgrad(dy, None, None, None, None, pooled, None, None, False, Mp, 1, 1, 1, dW, 64, prec=1)
        _close_bf16(dW, _bf16(dy).t() @ _bf16(pooled))


# ------------------------------------------------------------------------------------------- rows_op
@pytest.mark.parametrize("M,C,Cf", [(94080, 64, 256), (1470, 512, 1024), (5880, 256, 1024)])
def test_rows_op_copy_stats(dev, M, C, Cf):
    from ddl import ops
    ext = ops.require_ext()
    x = _rand(M, C, dev=dev, seed=1)
    buf = torch.zeros(M, Cf, device=dev)
    st = ext.rows_op(0, M, x, None, None, None, buf[:, :C], 1, 1, True)
    assert torch.equal(buf[:, :C], x) and buf[:, C:].abs().sum() == 0
    _close(st[:, 0].double().sum(0), x.double().sum(0), l2=1e-5, mx=1e-4)
    _close(st[:, 1].double().sum(0), (x.double() ** 2).sum(0), l2=1e-6, mx=1e-5)


@pytest.mark.parametrize("M,C,Cf", [(94080, 64, 256), (1470, 512, 1024), (5880, 96, 1024)])
def test_rows_op_copy_accumulator_stats(dev, M, C, Cf):
    """Block entry of the fused-statistics engine: the copy adds the channel sums to the fixed-point accumulators
    (2^28 scale, 8 replicas) that the first 1x1 conv derives its BN statistics from - no table, no bn_prep launch."""
    from ddl import ops
    ext = ops.require_ext()
    x = _rand(M, C, dev=dev, seed=1)
    buf = torch.zeros(M, Cf, device=dev)
    acc = torch.zeros(ext.STAT_REPLICAS, 2, Cf, dtype=torch.int64, device=dev)
    st = ext.rows_op(0, M, x, None, None, None, buf[:, :C], 1, 1, False, acc_out=acc)
    assert (st is None or st.numel() == 0) and torch.equal(buf[:, :C], x) and buf[:, C:].abs().sum() == 0
    assert int(acc[:, :, C:].abs().sum()) == 0
    got = acc.sum(0).double() / 2.0 ** 28
    _close(got[0, :C], x.double().sum(0), l2=1e-5, mx=1e-4)
    _close(got[1, :C], (x.double() ** 2).sum(0), l2=1e-6, mx=1e-5)
    # integer accumulation: a second launch gives exactly twice the sums, whatever the arrival order
    first = acc.sum(0).clone()
    ext.rows_op(0, M, x, None, None, None, buf[:, :C], 1, 1, False, acc_out=acc)
    assert torch.equal(acc.sum(0), 2 * first)


@pytest.mark.parametrize("B,H,C", [(30, 56, 256), (30, 28, 512), (30, 14, 1024)])
def test_rows_op_pool_fwd_bwd(dev, B, H, C):
    """transition glue: BN+ReLU+2x2 average pool forward (mode 1) and its backward with the ReLU mask, BN scale
    and (sum g, sum g*x) statistics (mode 2)."""
    from ddl import ops
    ext = ops.require_ext()
    W = H
    M, Mp = B * H * W, B * (H // 2) * (W // 2)
    x = _rand(M, C, dev=dev, seed=1)
    sc, sh = _rand(C, dev=dev, seed=2).abs() + 0.5, _rand(C, dev=dev, seed=3)
    pooled = torch.empty(Mp, C, device=dev)
    ext.rows_op(1, Mp, pooled, x, sc, sh, pooled, H, W, False)
    z = torch.relu(x.double() * sc.double() + sh.double()).view(B, H, W, C).permute(0, 3, 1, 2)
    ref = F.avg_pool2d(z, 2).permute(0, 2, 3, 1).reshape(Mp, C)
    _close(pooled, ref, l2=1e-6, mx=1e-5)
    d = _rand(Mp, C, dev=dev, seed=4, scale=0.02)
    out = torch.empty(M, C, device=dev)
    st = ext.rows_op(2, M, d, x, sc, sh, out, H, W, True)
    up = d.double().view(B, H // 2, W // 2, C).repeat_interleave(2, 1).repeat_interleave(2, 2).reshape(M, C) * 0.25
    pre = x.double() * sc.double() + sh.double()
    g = torch.where(pre > 0, up, torch.zeros_like(up))
    sure = pre.abs() > 1e-5
    assert float(((out.double() - sc.double() * g).abs() * sure).max()) < 1e-6
    _close(st[:, 0].double().sum(0), g.sum(0), l2=1e-4, mx=1e-3)
    _close(st[:, 1].double().sum(0), (g * x.double()).sum(0), l2=1e-4, mx=1e-3)


def test_rows_op_gap_bwd_and_affine_add(dev):
    from ddl import ops
    ext = ops.require_ext()
    B, H, W, C = 30, 7, 7, 1024
    M = B * H * W
    x = _rand(M, C, dev=dev, seed=1)
    sc, sh = _rand(C, dev=dev, seed=2).abs() + 0.5, _rand(C, dev=dev, seed=3)
    d = _rand(B, C, dev=dev, seed=4)
    out = torch.empty(M, C, device=dev)
    st = ext.rows_op(3, M, d, x, sc, sh, out, H, W, True)
    pre = x.double() * sc.double() + sh.double()
    up = d.double().repeat_interleave(H * W, 0) / (H * W)
    g = torch.where(pre > 0, up, torch.zeros_like(up))
    sure = pre.abs() > 1e-5
    assert float(((out.double() - sc.double() * g).abs() * sure).max()) < 1e-6
    _close(st[:, 0].double().sum(0), g.sum(0), l2=1e-4, mx=1e-3)
    # mode 4: out = in + sc * x + sh on a channel prefix of wider buffers, in place
    M, C0, Cf = 23520, 128, 512
    dbuf, xbuf = _rand(M, Cf, dev=dev, seed=5), _rand(M, Cf, dev=dev, seed=6)
    q, p = _rand(Cf, dev=dev, seed=7), _rand(Cf, dev=dev, seed=8)
    dx = torch.empty(M, C0, device=dev)
    ext.rows_op(4, M, dbuf[:, :C0], xbuf[:, :C0], q, p, dx, 28, 28, False)
    ref = dbuf[:, :C0].double() + q[:C0].double() * xbuf[:, :C0].double() + p[:C0].double()
    _close(dx, ref, l2=1e-6, mx=1e-5)


@pytest.mark.parametrize("G,C,mode", [(735, 128, 0), (46, 992, 1), (184, 640, 1), (12, 128, 0)])
def test_bn_bwd_prep(dev, G, C, mode):
    """Per-tile (sum g, sum g*x) tables -> dgamma / dbeta (accumulated) and the BN-backward coefficients
    (mode 0: k0, k1, k2 written; mode 1: deferred dense-block terms NQ += k1, NP += k2)."""
    from ddl import ops
    ext = ops.require_ext()
    count = float(G * 128)
    part = _rand(G, 2, C, dev=dev, seed=1, scale=0.05)
    mean, rstd = _rand(C, dev=dev, seed=2), _rand(C, dev=dev, seed=3).abs() + 0.5
    scale = _rand(C, dev=dev, seed=4).abs() + 0.5
    dg, db = _rand(C, dev=dev, seed=5), _rand(C, dev=dev, seed=6)
    dg0, db0 = dg.clone(), db.clone()
    k = _rand(3, C, dev=dev, seed=7, scale=1e-3)       # small running NQ / NP: the increment must stay visible in fp32
    k_before = k.clone()
    ext.bn_bwd_prep(part, count, mean, rstd, scale, dg, db, k[0] if mode == 0 else None, k[1], k[2], C, mode)
    sg, sgx = part[:, 0].double().sum(0), part[:, 1].double().sum(0)
    dgam = rstd.double() * (sgx - mean.double() * sg)
    _close(dg.double() - dg0.double(), dgam, l2=1e-5, mx=1e-4)
    _close(db.double() - db0.double(), sg, l2=1e-5, mx=1e-4)
    k1 = -scale.double() * (dgam / count) * rstd.double()
    k2 = -scale.double() * (sg / count) + scale.double() * (dgam / count) * rstd.double() * mean.double()
    if mode == 0:
        _close(k[0], scale, l2=1e-7, mx=1e-6)
        _close(k[1], k1, l2=1e-5, mx=1e-4)
        _close(k[2], k2, l2=1e-5, mx=1e-4)
    else:
        assert torch.equal(k[0], k_before[0])
        _close(k[1].double() - k_before[1].double(), k1, l2=1e-4, mx=1e-3)
        _close(k[2].double() - k_before[2].double(), k2, l2=1e-4, mx=1e-3)


# ------------------------------------------------------------------------------------------- units / model at B=30
def test_stem_at_benchmark_shape(dev):
    from test_gpu_native import _compare_unit
    _compare_unit(dev, ("stem",), (30, 3, 224, 224))


def test_dense_block3_at_benchmark_shape(dev):
    from test_gpu_native import _compare_unit
    _compare_unit(dev, ("denseblock3",), (30, 256, 14, 14), global_l2=True)


def test_dense_block4_head_at_benchmark_shape(dev):
    from test_gpu_native import _compare_unit
    _compare_unit(dev, ("denseblock4", "head"), (30, 512, 7, 7), global_l2=True)


def test_full_model_at_benchmark_shape(dev):
    """B=30 @ 224x224, forward + backward, native vs PyTorch fp32, both judged against PyTorch fp64 (global L2 of
    the whole gradient vector; input / logits in max-norm)."""
    from ddl.models.densenet import UNIT_NAMES
    from test_gpu_native import _compare_unit
    _compare_unit(dev, UNIT_NAMES, (30, 3, 224, 224), tol=2e-3, slack=10.0, global_l2=True)


def test_loss_curve_50_steps_tracks_the_pytorch_model(dev):
    """50 Adam steps at the benchmark shape (B=30 @ 224, four fixed synthetic batches cycled) - the native engine
    (hipGraph step) against the PyTorch model with torch.optim.Adam from the same initial weights."""
    from ddl import comm
    from ddl.config import TrainConfig
    from ddl.models.densenet import UNIT_NAMES
    from ddl.trainer import SingleStrategy
    from test_gpu_native import _ref_stage
    B = 30
    strat = SingleStrategy(TrainConfig(strategy="single", batch_size=B), comm.CommContext(0, 1, 0, dev, "none"))
    ref = _ref_stage(UNIT_NAMES, dev)
    strat.model.import_state_dict(ref.state_dict())
    ropt = torch.optim.Adam(ref.parameters())
    g = torch.Generator().manual_seed(0)
    data = [(torch.randint(0, 256, (B, 3, 224, 224), dtype=torch.uint8, generator=g).to(dev),
             torch.randint(0, 5, (B,), generator=g).to(dev)) for _ in range(4)]
    ref.train()
    ln, lr = [], []
    for step in range(50):
        x, y = data[step % 4]
        ln.append(float(strat.train_step(x, y, B)[0]))
        ropt.zero_grad()
        rl = F.cross_entropy(ref(x), y)
        rl.backward()
        ropt.step()
        lr.append(float(rl.detach()))
    assert all(v == v for v in ln)
    # identical start, then two fp32 implementations of a chaotic system: the curves must stay together on
    # average and both must learn the four batches
    assert abs(ln[0] - lr[0]) < 1e-3 * max(1.0, abs(lr[0]))
    first = sum(abs(a - b) for a, b in zip(ln[:8], lr[:8])) / 8
    whole = sum(abs(a - b) for a, b in zip(ln, lr)) / len(ln)
    assert first < 0.02 and whole < 0.15, (first, whole, ln[-4:], lr[-4:])
    assert sum(ln[-4:]) < 0.7 * sum(ln[:4]) and sum(lr[-4:]) < 0.7 * sum(lr[:4]), (ln, lr)
