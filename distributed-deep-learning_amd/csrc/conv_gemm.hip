// Implicit-GEMM convolution for gfx950 (MI355X), fp32 in / fp32 accumulate on the matrix cores
// (v_mfma_f32_32x32x2_f32 - exact f32, 64 FLOP/clk/SIMD; gfx950 has no TF32/xf32).
//
//   out[M x N] (op)= prologue(rows)[M x taps*C] * W
//
// One kernel family serves every GEMM-shaped op of DenseNet-121 except the weight gradients
// (conv_wgrad.hip) - reference ops: the cuDNN conv fwd / bwd-data calls behind
// torchvision's _DenseLayer / _Transition:
//   * 1x1 bottleneck conv forward with BN+ReLU fused into the A-operand load (the normalised
//     tensor is never written), reading a channel *prefix* of the dense block's NHWC buffer;
//   * 3x3 conv forward (taps = 9, gathered rows, zero padding applied AFTER BN+ReLU), writing
//     its 32 channels straight into the block buffer slice (concat is free), with per-channel
//     sum / sum-of-squares of the output reduced in the epilogue (BN statistics of the new
//     channels, shared by every later layer of the block);
//   * both data gradients, with ReLU/BN-backward masks, the per-channel BN-backward sums and the
//     "+=" into the block's shared gradient buffer fused into the epilogue.
//
// Tiling (wave64): a workgroup is WM x WN x KS waves.  Each wave owns TM x TN MFMA tiles of
// 32x32 (16 accumulator VGPRs each); the KS wave groups split every K slab between them
// (intra-workgroup split-K, combined through LDS at the end) - DenseNet at batch 30 has layers
// with only 46..184 row tiles, and without the K split those run one wave per CU for 36 dependent
// slabs.  K is walked in slabs of 32*KS (flat index over taps x channels); A and B slabs are
// staged global -> registers -> LDS with the next slab's global loads issued BEFORE the current
// slab's MFMAs and consumed (BN affine + ReLU, LDS write) AFTER them, one barrier per slab.
// LDS images keep K contiguous with a +4 float pad ([rows][32*KS+4]): ds_read_b128 is
// conflict-free for its 16-lane groups and hands a lane 4 consecutive k; MFMA e of a group takes
// element e from both operands, so lane half h contributes k = 8s + 4h + e - the same
// permutation on A and B, hence a valid reduction.  Weights that arrive K-major (the forward
// tensor read transposed by the data-gradient) keep that orientation in LDS ([k][BN+4], b128
// writes) and are read with four conflict-free ds_read_b32 instead of being transposed.
// fp32 MFMA issues once per 64 cycles, so LDS and VALU (the fused prologue) are far from
// limiting; what matters is keeping every SIMD's MFMA queue fed.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "conv_params.h"
#include "ddl_kernels.h"
#include "device_utils.h"
#include "conv_common.h"

namespace ddl {

namespace {

// Operation kinds (what DenseNet needs; anything else is rejected on the host):
//   0 FWD     one source, n-major weights, plain store            (+ column statistics v, v*v)
//   1 TDGRAD  one source, k-major weights, plain store            (transition data gradient)
//   2 DGRAD3  two sources, k-major, masked store                  (+ statistics g, g*aux)
//   3 DGRAD1  two sources, k-major, masked accumulate into out    (+ statistics g, g*aux)
//
// PREC = 1: bf16 matrix-core inputs, fp32 accumulate (`--dtype bf16`).  Storage stays fp32; the
// operands are rounded (RNE, v_cvt_pk_bf16_f32) when they are written to LDS, after the fused
// BN/ReLU prologue, and consumed by v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate).  Both LDS
// images are then K-contiguous bf16 rows of 2*BKT + 16 bytes (conflict-free ds_read_b128); k-major
// weights are transposed while being staged, so the MFMA loop is the same for every kind.
//
// PREC = 3: fp32 arithmetic ON the bf16 matrix cores (`--dtype fp32x3`).  gfx950 runs fp32 MFMA at 1/16 of the
// bf16 rate and has no TF32, so an fp32 operand is split into three bf16 pieces when it is written to LDS
// (x = hi + mid + lo exactly to 2^-25 |x|: each residual x - bf16(x) is exact in fp32) and a product is the sum of
// the six piece products of weight >= 2^-16 (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid - every bf16 x bf16
// product is exact in the fp32 accumulator); the three dropped ones are <= 2^-23 |a*b| together, the size of the
// rounding of ONE fp32 multiply-add.  6 bf16 MFMAs replace 16 fp32 MFMAs of the same cycles each: 2.7x the fp32
// matrix-core roof at fp32-level accuracy (tests/test_gpu_split.py compares both paths with fp64).
// x -> (hi, mid, lo) bf16, RNE at every level (the residuals are exact)
__device__ __forceinline__ void split3(const f32x4 v, bf16x4& h, bf16x4& m, bf16x4& l) {
  h = __builtin_convertvector(v, bf16x4);
  f32x4 r = v - __builtin_convertvector(h, f32x4);
  m = __builtin_convertvector(r, bf16x4);
  r = r - __builtin_convertvector(m, f32x4);
  l = __builtin_convertvector(r, bf16x4);
}

template <int BM, int BN, int WM, int WN, int KS, int NBUF, int PF, int KIND, int PREC, bool RB>
__global__ void __launch_bounds__(64 * WM * WN * KS) conv_gemm_kernel(const ConvGemmParams p) {
  constexpr bool BF = PREC != 0;         // operands live in LDS as bf16 images
  constexpr int NP = PREC == 3 ? 3 : 1;  // images per operand (split pieces)
  constexpr bool TWO = KIND >= 2, KMAJOR = KIND >= 1;
  constexpr int LDH = 32 * KS + 8;       // bf16 image row stride (elements)
  constexpr int NT = 64 * WM * WN * KS;
  constexpr int TM = BM / (32 * WM);
  constexpr int TN = BN / (32 * WN);
  constexpr int BKT = 32 * KS;           // K slab
  constexpr int LDA = BKT + 4;           // LDS row stride of K-contiguous images
  constexpr int LDBK = BN + 4;           // LDS row stride of the K-major weight image
  constexpr int Q = 8 * KS;              // float4 per K-contiguous row
  constexpr int RSTEP = NT / Q;          // rows covered per pass
  constexpr int PA = (BM * Q) / NT;
  constexpr int PB = (BN * Q) / NT;
  constexpr int B_ELEMS = KMAJOR ? BKT * LDBK : BN * LDA;
  constexpr int BQ4 = BN / 4;            // float4 per row of the K-major weight image
  constexpr int KROWS = NT / BQ4;        // K rows covered per pass of the K-major weight load
  static_assert((BM * Q) % NT == 0 && (BN * Q) % NT == 0 && NT % Q == 0, "slab must divide over the workgroup");
  static_assert(NT % BQ4 == 0, "k-major weight slab must divide over the workgroup");
  static_assert(TM >= 1 && TN >= 1, "bad wave layout");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  DDL_PROBE(p, 0);
  // forward kind with a BN prologue: scale / shift of all C channels live in LDS ahead of the staging
  // buffers (copied from the tables, or derived from the fused-statistics accumulators)
  const bool affine = p.a.s0 != nullptr || (KIND == 0 && p.bn.acc != nullptr) || (KIND == 3 && p.fk.acc != nullptr);
  constexpr bool LDS_TABLES = KIND == 0 || KIND == 3;     // per-channel prologue vectors staged in LDS
  const int soff = (LDS_TABLES && affine) ? (KIND == 0 ? 2 : 3) * p.C : 0;
  float* scs = smem;                       // KIND 0: [C] scale, [C] shift;  KIND 3: [C] k0, [C] k2, [C] k1
  float* shs = smem + p.C;
  float* k1s = smem + 2 * p.C;
  float* As = smem + soff;                 // [NBUF][BM][LDA]
  float* Bs = As + NBUF * BM * LDA;        // [NBUF][B_ELEMS]
  __bf16* Ah = reinterpret_cast<__bf16*>(smem + soff);   // BF: [NBUF][NP][BM][LDH]
  __bf16* Bh = Ah + NBUF * NP * BM * LDH;                //     [NBUF][NP][BN][LDH]

  // wave index as a scalar: the wave-uniform sub-tile guards then branch on SCC instead of masking EXEC
  // (the BN = 32 forms have no live guard; scalarising them only moves an index table to scratch)
  const int tid = threadIdx.x, lane = tid & 63, wid = BN == 32 ? tid >> 6 : __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ks = wid / (WM * WN);
  const int wmn = wid - ks * (WM * WN);
  const int wm = wmn / WN, wn = wmn - wm * WN;
  const int li = lane & 31, lh = lane >> 5;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a
  // contiguous run of row tiles (neighbouring tiles share 3x3 halo rows in that XCD's L2)
  int tile_x;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    tile_x = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int m0 = tile_x * BM;
  const int n0 = blockIdx.y * BN;
  const int cchunks = p.C >> 5;            // 32-wide sub-slabs per tap (C % 32 == 0, checked on host)
  const int NS = p.taps * cchunks;         // sub-slabs in all
  const int KT = (NS + KS - 1) / KS;

  // source descriptors start at the first row this tile can touch (its rows, minus the 3x3 halo)
  const int halo_rows = p.taps > 1 ? p.W + 1 : 0;
  constexpr bool rb = RB;      // operand views beyond 2 GiB (host decision, device_utils.h)
  const int mbase = (rb && m0 > halo_rows) ? m0 - halo_rows : 0;
  const rsrc_t rA0 = make_rsrc_rows(p.a.X0, p.a.bytes0, mbase, p.a.ld0, rb);
  const rsrc_t rA1 = make_rsrc_rows(TWO ? p.a.X1 : p.a.X0, TWO ? p.a.bytes1 : p.a.bytes0, mbase, TWO ? p.a.ld1 : p.a.ld0, rb);
  const rsrc_t rW = make_rsrc(p.Wt, p.wbytes);

  // ---- per-thread bookkeeping -----------------------------------------------------------------
  const int aq = tid % Q;                  // this thread's float4 column inside a slab row
  const int a_sub = aq >> 3;               // which 32-wide sub-slab
  const int a_col = 4 * (aq & 7);          // channel offset inside it
  const int a_row0 = tid / Q;
  unsigned a_off0[PA], a_off1[TWO ? PA : 1], a_bits[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int m = m0 + a_row0 + i * RSTEP;
    a_off0[i] = (unsigned)(m - mbase) * (unsigned)(p.a.ld0 * 4);
    if (TWO) a_off1[i] = (unsigned)(m - mbase) * (unsigned)(p.a.ld1 * 4);
    unsigned bits = 0;
    if (m < p.M) {
      bits = 1;
      if (p.taps > 1) {       // bit t: tap t of this pixel lies inside the image (flip: offsets negated)
        const int rem = m % (p.H * p.W);
        const int y = rem / p.W, x = rem - y * p.W;
        const unsigned up = y > 0, down = y < p.H - 1, left = x > 0, right = x < p.W - 1;
        const unsigned yb = p.flip ? (down | 2u | (up << 2)) : (up | 2u | (down << 2));
        const unsigned xb = p.flip ? (right | 2u | (left << 2)) : (left | 2u | (right << 2));
        bits = ((yb & 1u) ? xb : 0u) | ((yb & 2u) ? (xb << 3) : 0u) | ((yb & 4u) ? (xb << 6) : 0u);
      }
    }
    a_bits[i] = bits;
  }
  unsigned b_off[PB];                      // n-major weights: byte offset of this thread's rows
  if (!KMAJOR) {
#pragma unroll
    for (int i = 0; i < PB; ++i) b_off[i] = (unsigned)(n0 + a_row0 + i * RSTEP) * (unsigned)(p.ldb * 4);
  }
  const int bk_row = tid / BQ4;            // k-major weights: K row / output-channel quad of this thread
  const int bk_n = n0 + 4 * (tid - bk_row * BQ4);

  // one slab in flight in registers (PF = 2: two of them)
  struct Stage {
    f32x4 a_v0[PA], a_v1[TWO ? PA : 1];
    f32x4 a_s0, a_s1, a_s2;
    int a_c;
    unsigned a_mask;
    f32x4 b_reg[PB];
  };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // sub-slab index -> (tap, first channel); inv_cc = ceil(2^16 / cchunks), exact for idx <= 320
  auto split = [&](int idx, int& tap, int& c0) {
    tap = (int)(((unsigned)idx * (unsigned)p.inv_cc) >> 16);
    c0 = (idx - tap * cchunks) * 32;
  };
  auto tap_pix = [&](int tap) {            // pixel offset of a 3x3 tap (rows of the NHWC tensor)
    const int ty = (tap * 11) >> 5, tx = tap - 3 * ty;
    const int o = (ty - 1) * p.W + (tx - 1);
    return p.flip ? -o : o;
  };

  auto load_slab = [&](int kt, Stage& S) {
    const int idx = kt * KS + a_sub;
    const bool sub_ok = idx < NS;
    int tap, c0;
    split(idx, tap, c0);
    const int c = c0 + a_col;
    {   // ---- A: rows gathered with this thread's tap offset, raw values only ----
      const int pix = p.taps > 1 ? tap_pix(tap) : 0;
      const unsigned d0 = (unsigned)((pix * p.a.ld0 + c) * 4);
      const unsigned d1 = TWO ? (unsigned)((pix * p.a.ld1 + c) * 4) : 0u;
      S.a_s0 = zero4; S.a_s1 = zero4; S.a_s2 = zero4;
      S.a_c = c;                           // forward kind: scale / shift come from LDS at store time
      if (!LDS_TABLES && affine) {         // c < C always (even for idx >= NS the loads stay in range)
        S.a_s0 = ld4(p.a.s0 + c);
        S.a_s2 = ld4(p.a.s2 + c);
        if (TWO) S.a_s1 = ld4(p.a.s1 + c);
      }
      S.a_mask = 0;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const bool ok = sub_ok && ((a_bits[i] >> tap) & 1u);
        S.a_mask |= ok ? (1u << i) : 0u;
        S.a_v0[i] = bld4(rA0, ok ? a_off0[i] + d0 : OOB);
        if (TWO) S.a_v1[i] = bld4(rA1, ok ? a_off1[i] + d1 : OOB);
      }
    }
    // ---- B ----
    if (!KMAJOR) {                         // same (tap, c) as A: both index the slab row by tid % Q
      const unsigned dcol = (unsigned)((tap * p.C + c) * 4);
#pragma unroll
      for (int i = 0; i < PB; ++i) S.b_reg[i] = bld4(rW, sub_ok ? b_off[i] + dcol : OOB);
    } else {
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int kk = bk_row + i * KROWS;
        const int bidx = kt * KS + (KS == 1 ? 0 : (kk >> 5));
        int btap, bc0;
        split(bidx, btap, bc0);
        const unsigned off = (unsigned)(((bc0 + (kk & 31)) * p.ldb + btap * p.N + bk_n) * 4);
        S.b_reg[i] = bld4(rW, (bidx < NS && bk_n < p.N) ? off : OOB);
      }
    }
  };

  auto store_slab = [&](int buf, const Stage& S) {
    float* Ab = As + buf * BM * LDA;
    float* Bb = Bs + buf * B_ELEMS;
    __bf16* Ahb = Ah + buf * NP * BM * LDH;
    __bf16* Bhb = Bh + buf * NP * BN * LDH;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      f32x4 v = S.a_v0[i];
      if (affine) {          // deferred prologue: BN affine (+ second source) + ReLU, zero where invalid
        if (LDS_TABLES) v = ld4(scs + S.a_c) * v + ld4(shs + S.a_c);
        else v = S.a_s0 * v + S.a_s2;
        if (TWO) v += (KIND == 3 ? ld4(k1s + S.a_c) : S.a_s1) * S.a_v1[i];
        if (p.a.relu) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (!((S.a_mask >> i) & 1u)) v = zero4;
      }
      if (PREC == 3) {
        bf16x4 h0, h1, h2;
        split3(v, h0, h1, h2);
        __bf16* q = Ahb + (a_row0 + i * RSTEP) * LDH + 4 * aq;
        *reinterpret_cast<bf16x4*>(q) = h0;
        *reinterpret_cast<bf16x4*>(q + BM * LDH) = h1;
        *reinterpret_cast<bf16x4*>(q + 2 * BM * LDH) = h2;
      } else if (BF) {
        *reinterpret_cast<bf16x4*>(Ahb + (a_row0 + i * RSTEP) * LDH + 4 * aq) = __builtin_convertvector(v, bf16x4);
      } else {
        *reinterpret_cast<f32x4*>(Ab + (a_row0 + i * RSTEP) * LDA + 4 * aq) = v;
      }
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      if (BF) {
        bf16x4 hp[3];
        if (PREC == 3) split3(S.b_reg[i], hp[0], hp[1], hp[2]);
        else hp[0] = __builtin_convertvector(S.b_reg[i], bf16x4);
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) {
          const bf16x4 h = hp[pc];
          if (!KMAJOR) {
            *reinterpret_cast<bf16x4*>(Bhb + pc * BN * LDH + (a_row0 + i * RSTEP) * LDH + 4 * aq) = h;
          } else {             // transpose on the way in: this thread holds 4 output channels of one k
            __bf16* q = Bhb + pc * BN * LDH + 4 * (tid - bk_row * BQ4) * LDH + (bk_row + i * KROWS);
            q[0] = h[0]; q[LDH] = h[1]; q[2 * LDH] = h[2]; q[3 * LDH] = h[3];
          }
        }
      } else if (!KMAJOR) {
        *reinterpret_cast<f32x4*>(Bb + (a_row0 + i * RSTEP) * LDA + 4 * aq) = S.b_reg[i];
      } else {
        *reinterpret_cast<f32x4*>(Bb + (bk_row + i * KROWS) * LDBK + 4 * (tid - bk_row * BQ4)) = S.b_reg[i];
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int buf) {
    if (PREC == 3) {
      const __bf16* Ab = Ah + buf * NP * BM * LDH + (wm * TM * 32 + li) * LDH + ks * 32 + 8 * lh;
      const __bf16* Bb = Bh + buf * NP * BN * LDH + (wn * TN * 32 + li) * LDH + ks * 32 + 8 * lh;
#pragma unroll
      for (int s = 0; s < 2; ++s) {        // two 16-wide k steps per 32-wide sub-slab, three pieces per operand
        bf16x8 af[3][TM], bf[3][TN];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
          for (int i = 0; i < TM; ++i) af[pc][i] = *reinterpret_cast<const bf16x8*>(Ab + pc * BM * LDH + i * 32 * LDH + 16 * s);
#pragma unroll
          for (int j = 0; j < TN; ++j) bf[pc][j] = *reinterpret_cast<const bf16x8*>(Bb + pc * BN * LDH + j * 32 * LDH + 16 * s);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (n0 + (wn * TN + j) * 32 >= p.N) continue;
#pragma unroll
          for (int i = 0; i < TM; ++i) {     // small terms first; one accumulation chain per tile runs back to back
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], bf[0][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[2][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[1][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[0][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[1][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
          }
        }
      }
      return;
    }
    if (BF) {
      const __bf16* Ab = Ah + buf * BM * LDH + (wm * TM * 32 + li) * LDH + ks * 32 + 8 * lh;
      const __bf16* Bb = Bh + buf * BN * LDH + (wn * TN * 32 + li) * LDH + ks * 32 + 8 * lh;
#pragma unroll
      for (int s = 0; s < 2; ++s) {        // two 16-wide k steps per 32-wide sub-slab
        bf16x8 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(Ab + i * 32 * LDH + 16 * s);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(Bb + j * 32 * LDH + 16 * s);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (n0 + (wn * TN + j) * 32 >= p.N) continue;
#pragma unroll
          for (int i = 0; i < TM; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
      }
      return;
    }
    const float* Ab = As + buf * BM * LDA + (wm * TM * 32 + li) * LDA + ks * 32 + 4 * lh;
    const float* Bb = Bs + buf * B_ELEMS;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = ld4(Ab + i * 32 * LDA + 8 * s);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (!KMAJOR) {
          bf[j] = ld4(Bb + ((wn * TN + j) * 32 + li) * LDA + ks * 32 + 4 * lh + 8 * s);
        } else {
          const float* q = Bb + (ks * 32 + 8 * s + 4 * lh) * LDBK + (wn * TN + j) * 32 + li;
          bf[j].x = q[0];
          bf[j].y = q[LDBK];
          bf[j].z = q[2 * LDBK];
          bf[j].w = q[3 * LDBK];
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (n0 + (wn * TN + j) * 32 >= p.N) continue;      // wave-uniform: sub-tile beyond N (N % 128 != 0)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < TM; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
      }
    }
  };

  // ---- forward BN prologue tables -> LDS (runs while the first slab's global loads are in flight) ----
  auto bn_tables = [&]() { conv_fill_tables<NT, KIND>(p, affine, scs, shs, k1s, tid); };

  // ---- main loop: register-staged; NBUF = 2: one barrier per slab, NBUF = 1: half the LDS
  // (more workgroups per CU hide the extra barrier and every other latency) -------------------
  if (PF == 1) {
    Stage S;
    load_slab(0, S);
    bn_tables();
    DDL_PROBE(p, 1);
    if (LDS_TABLES) __syncthreads();
    store_slab(0, S);
    __syncthreads();
    DDL_PROBE(p, 2);
    for (int kt = 0; kt < KT; ++kt) {
      const int buf = NBUF == 2 ? (kt & 1) : 0;
      if (kt + 1 < KT) load_slab(kt + 1, S);      // global loads fly under this slab's MFMAs
      compute(buf);
      if (NBUF == 1) __syncthreads();
      if (kt + 1 < KT) store_slab(NBUF == 2 ? (buf ^ 1) : 0, S);
      __syncthreads();
    }
  } else {
    // prefetch distance 2 (NBUF == 2): slab kt in LDS, slab kt+1 in registers, slab kt+2 being loaded.
    // (Round 2, measured: the conditional prefetches below make the number of outstanding loads path-dependent, so
    // the compiler drains to vmcnt(0) before the older slab is consumed and the distance is really 1.  Issuing
    // every load unconditionally - out-of-range slabs read zeros - gives the intended vmcnt(5) / vmcnt(15) waits
    // for distance 2 / 4, and changes NOTHING in the run time of any DenseNet shape (bench/kernel_bench.py --sweep:
    // 19.2 / 19.6 vs 19.3 us for the block-3 1x1 forward): the phase probes show the K loop of the small-M tiles at
    // 1.4 us per slab because the two K-split waves of a tile share one SIMD's MFMA pipe (2 x 0.43 us) plus the
    // barrier - it is matrix-core bound per CU, not latency bound.  The simpler form stays.)
    // For the latency-bound small-M layers (a handful of workgroups per CU at most) a slab's loads
    // get two compute phases to land instead of one.
    Stage S0, S1;
    load_slab(0, S0);
    if (KT > 1) load_slab(1, S1);
    bn_tables();
    DDL_PROBE(p, 1);
    if (LDS_TABLES) __syncthreads();
    store_slab(0, S0);
    __syncthreads();
    DDL_PROBE(p, 2);
    for (int kt = 0; kt < KT; kt += 2) {
      if (kt + 2 < KT) load_slab(kt + 2, S0);
      compute(0);
      if (kt + 1 < KT) store_slab(1, S1);
      __syncthreads();
      if (kt + 1 >= KT) break;
      if (kt + 3 < KT) load_slab(kt + 3, S1);
      compute(1);
      if (kt + 2 < KT) store_slab(0, S0);
      __syncthreads();
    }
  }

  DDL_PROBE(p, 3);
  // ---- combine the K-split partial accumulators ---------------------------------------------
  if (KS > 1) {
    float* red = smem;   // [(KS-1)][WM*WN][TM*TN*16][64]
    if (ks > 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            red[((((ks - 1) * (WM * WN) + wmn) * (TM * TN) + i * TN + j) * 16 + r) * 64 + lane] = acc[i][j][r];
    }
    __syncthreads();
    if (ks == 0) {
#pragma unroll
      for (int k2 = 1; k2 < KS; ++k2)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              acc[i][j][r] += red[((((k2 - 1) * (WM * WN) + wmn) * (TM * TN) + i * TN + j) * 16 + r) * 64 + lane];
    }
    __syncthreads();
  }

  DDL_PROBE(p, 4);
  conv_epilogue<BM, BN, WM, WN, NT, KIND, RB>(p, acc, smem, m0, n0, tile_x, tid, ks, wmn);
  DDL_PROBE(p, 8);
}

template <int BM, int BN, int WM, int WN, int KS, int NBUF, int PF, int KIND, int PREC, bool RB>
void launch_inst_rb(const ConvGemmParams& p, hipStream_t s) {
  constexpr bool KMAJOR = KIND >= 1;
  constexpr int BKT = 32 * KS;
  constexpr size_t b_elems = KMAJOR ? (size_t)BKT * (BN + 4) : (size_t)BN * (BKT + 4);
  constexpr size_t stage32 = (NBUF * (size_t)BM * (BKT + 4) + NBUF * b_elems) * sizeof(float);
  constexpr size_t stage3 = (size_t)NBUF * 3 * (BM + BN) * (BKT + 8) * 2;        // three bf16 images per operand
  constexpr size_t stage = PREC == 3 ? stage3 : stage32;
  // the K-split combine and the statistics reduce reuse the staging area
  constexpr size_t red = (size_t)(KS - 1) * (BM / 32) * (BN / 32) * 16 * 64 * sizeof(float);
  constexpr size_t epi = (size_t)WM * WN * 32 * kEpiLd * sizeof(float);      // epilogue transposition images
  constexpr size_t sta = (size_t)2 * WM * 8 * BN * sizeof(float);        // statistics combine
  constexpr size_t m1 = stage > red ? stage : red, m2 = epi > sta ? epi : sta;
  constexpr size_t lds_max = (m1 > m2 ? m1 : m2) + ((KIND == 0 || KIND == 3) ? kMaxAffineC * 3 * sizeof(float) : 0);
  // kinds 0 / 3: per-channel prologue vectors of the C input channels ahead of the staging buffers
  const bool has_tables = (KIND == 0 && (p.a.s0 != nullptr || p.bn.acc != nullptr)) ||
                          (KIND == 3 && (p.a.s0 != nullptr || p.fk.acc != nullptr));
  const size_t tables = has_tables ? (size_t)p.C * (KIND == 0 ? 2 : 3) * sizeof(float) : 0;
  const size_t m1t = (stage + tables) > red ? (stage + tables) : red;
  const size_t lds = m1t > m2 ? m1t : m2;
  static bool attr_set = false;
  if (!attr_set) {
    DDL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<BM, BN, WM, WN, KS, NBUF, PF, KIND, PREC, RB>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
    attr_set = true;
  }
  dim3 grid((unsigned)((p.M + BM - 1) / BM), (unsigned)((p.N + BN - 1) / BN));
  hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, KS, NBUF, PF, KIND, PREC, RB>), grid, dim3(64 * WM * WN * KS), lds, s, p);
}

template <int BM, int BN, int WM, int WN, int KS, int NBUF, int PF, int KIND, int PREC>
void launch_inst(const ConvGemmParams& p, hipStream_t s) {
  // bf16 forms are instantiated rebasing only (not the headline path; halves the kernel count of this file)
  if (PREC != 0 || needs_rebase(p.a.bytes0, p.a.bytes1, p.obytes, p.auxbytes)) launch_inst_rb<BM, BN, WM, WN, KS, NBUF, PF, KIND, PREC, true>(p, s);
  else launch_inst_rb<BM, BN, WM, WN, KS, NBUF, PF, KIND, PREC, PREC != 0>(p, s);       // (PREC, RB) = (0, false)
}

template <int BM, int BN, int WM, int WN, int KS, int NBUF, int PF = 1>
void launch_cfg(const ConvGemmParams& p, hipStream_t s) {
  static_assert(PF == 1 || NBUF == 2, "prefetch distance 2 needs both LDS buffers");
  if (p.bf16 == 3) {
    // the split form needs 1.5x the staging LDS of fp32: configurations beyond the 160 KB of a CU do not exist
    if constexpr ((size_t)NBUF * 3 * (BM + BN) * (32 * KS + 8) * 2 + kMaxAffineC * 3 * sizeof(float) <= 160 * 1024) {
      switch (p.kind) {
        case 0: launch_inst<BM, BN, WM, WN, KS, NBUF, PF, 0, 3>(p, s); break;
        case 1: launch_inst<BM, BN, WM, WN, KS, NBUF, PF, 1, 3>(p, s); break;
        case 2: launch_inst<BM, BN, WM, WN, KS, NBUF, PF, 2, 3>(p, s); break;
        default: launch_inst<BM, BN, WM, WN, KS, NBUF, PF, 3, 3>(p, s); break;
      }
      return;
    } else {
      throw std::runtime_error("conv_gemm: this tile configuration has no fp32x3 form (LDS)");
    }
  }
  if (p.bf16) {
    switch (p.kind) {
      case 0: launch_inst<BM, BN, WM, WN, KS, NBUF, PF, 0, 1>(p, s); break;
      case 1: launch_inst<BM, BN, WM, WN, KS, NBUF, PF, 1, 1>(p, s); break;
      case 2: launch_inst<BM, BN, WM, WN, KS, NBUF, PF, 2, 1>(p, s); break;
      default: launch_inst<BM, BN, WM, WN, KS, NBUF, PF, 3, 1>(p, s); break;
    }
    return;
  }
  switch (p.kind) {
    case 0: launch_inst<BM, BN, WM, WN, KS, NBUF, PF, 0, 0>(p, s); break;
    case 1: launch_inst<BM, BN, WM, WN, KS, NBUF, PF, 1, 0>(p, s); break;
    case 2: launch_inst<BM, BN, WM, WN, KS, NBUF, PF, 2, 0>(p, s); break;
    default: launch_inst<BM, BN, WM, WN, KS, NBUF, PF, 3, 0>(p, s); break;
  }
}

// Tile configurations (BM, BN, waves M x N, K split, LDS buffers).  `force` >= 1000 selects
// config (force - 1000) explicitly (bench/kernel_bench.py --cfg); 32/64/128 force the row-tile
// height of the default family; 0 = heuristic.
struct TileCfg { int bm, bn; };
constexpr int NUM_CFG = 22;
const TileCfg kCfg[NUM_CFG] = {{128, 128}, {64, 128}, {32, 128}, {128, 32}, {64, 32}, {32, 32},
                               {128, 128}, {64, 128}, {32, 128}, {128, 32}, {64, 32}, {32, 32},
                               {128, 128}, {64, 128}, {32, 128}, {128, 32}, {64, 32}, {32, 32},
                               {32, 32}, {32, 32}, {32, 128}, {64, 32}};

void launch_by_id(int id, const ConvGemmParams& p, hipStream_t s) {
  switch (id) {
    case 0: launch_cfg<128, 128, 2, 2, 1, 2>(p, s); break;
    case 1: launch_cfg<64, 128, 2, 2, 1, 2>(p, s); break;
    case 2: launch_cfg<32, 128, 1, 4, 2, 2>(p, s); break;
    case 3: launch_cfg<128, 32, 4, 1, 1, 2>(p, s); break;
    case 4: launch_cfg<64, 32, 2, 1, 2, 2>(p, s); break;
    case 5: launch_cfg<32, 32, 1, 1, 4, 2>(p, s); break;
    case 6: launch_cfg<128, 128, 2, 2, 1, 1>(p, s); break;
    case 7: launch_cfg<64, 128, 2, 2, 1, 1>(p, s); break;
    case 8: launch_cfg<32, 128, 1, 4, 2, 1>(p, s); break;
    case 9: launch_cfg<128, 32, 4, 1, 1, 1>(p, s); break;
    case 10: launch_cfg<64, 32, 2, 1, 2, 1>(p, s); break;
    case 11: launch_cfg<32, 32, 1, 1, 4, 1>(p, s); break;
    case 12: launch_cfg<128, 128, 2, 2, 1, 2, 2>(p, s); break;
    case 13: launch_cfg<64, 128, 2, 2, 1, 2, 2>(p, s); break;
    case 14: launch_cfg<32, 128, 1, 4, 2, 2, 2>(p, s); break;
    case 15: launch_cfg<128, 32, 4, 1, 1, 2, 2>(p, s); break;
    case 16: launch_cfg<64, 32, 2, 1, 2, 2, 2>(p, s); break;
    case 17: launch_cfg<32, 32, 1, 1, 4, 2, 2>(p, s); break;
    case 18: launch_cfg<32, 32, 1, 1, 8, 1>(p, s); break;       // 8 / 16 waves on one 32-row tile: small-M layers
    case 19: launch_cfg<32, 32, 1, 1, 16, 1>(p, s); break;      //   have < 1 workgroup per CU, so the K split supplies
    case 20: launch_cfg<32, 128, 1, 4, 4, 1>(p, s); break;      //   the waves that overlap VALU staging with MFMA
    // (round 3, measured and removed: 96-row tiles - a grid just under one round of resident workgroups, no tail
    // round: 94080 rows = 980 tiles at 4 per CU instead of 1470 tiles of 64 at 5 per CU.  profiles/r3/ksweep_b30_96rows.txt:
    // no better than the 64-row tiles on block 1, 27.0 vs 31.6 us on the block-2 3x3 data gradient - and 8.58 vs 8.56 ms
    // INSIDE the step, so the configurations are gone again.)
    default: launch_cfg<64, 32, 2, 1, 4, 1>(p, s); break;
  }
}

// Heuristic from the measured sweep (bench/kernel_bench.py --sweep, B=30 shapes, MI355X):
//  * few row tiles (block 4 at batch 30: M = 1470): 32-column tiles, so the output-channel axis
//    supplies the workgroups that the pixel axis cannot (fwd 1x1: 20.5 -> 10.7 us);
//  * 128-column family: 64-row tiles beat 128-row tiles up to ~1500 row tiles (smaller tail when
//    tiles are dealt over the resident slots), single LDS buffer (4 workgroups / CU) >= double;
//  * 32-column family (3x3 forward): 128 rows for the big maps, 32 rows x 4-way K split below.
// DDL_CFG_RULES="kind:taps:minM:maxM:cfg,..." overrides the heuristic for matching problems (first match wins).
// Tile choices have to be judged INSIDE the step (next to the weight-gradient chain a configuration that wins the
// microbenchmark can lose - README "dead ends"); this is the hook bench/r2/cfg_rules_ab.sh drives.
struct CfgRule { int kind, taps; long lo, hi; int cfg; };
static const std::vector<CfgRule>& cfg_rules() {
  static const std::vector<CfgRule> rules = [] {
    std::vector<CfgRule> r;
    const char* e = ddl_knob("DDL_CFG_RULES");
    if (e == nullptr) return r;
    std::string s(e);
    size_t pos = 0;
    while (pos < s.size()) {
      size_t end = s.find(',', pos);
      if (end == std::string::npos) end = s.size();
      CfgRule c{};
      if (sscanf(s.substr(pos, end - pos).c_str(), "%d:%d:%ld:%ld:%d", &c.kind, &c.taps, &c.lo, &c.hi, &c.cfg) == 5 &&
          c.cfg >= 0 && c.cfg < NUM_CFG)
        r.push_back(c);
      pos = end + 1;
    }
    return r;
  }();
  return rules;
}

int pick_cfg(const ConvGemmParams& p, int force) {
  const int M = p.M, N = p.N;
  if (force >= 1000 && force < 1000 + NUM_CFG) return force - 1000;
  if (!force) {
    for (const CfgRule& c : cfg_rules())
      if (c.kind == p.kind && c.taps == p.taps && M >= c.lo && M <= c.hi) return c.cfg;
  }
  int bm = (force == 32 || force == 64 || force == 128) ? force : 0;
  if (N > 32) {
    if (!bm && M <= 2048) {
      // 1x1 data gradient of the last block (N = C = 512..1024 columns): 32x128 tiles x 8 waves (11.2 vs 13.9 us)
      if (p.kind == 3 && N >= 512) return 8;
      return ((long)((M + 63) / 64) * ((N + 31) / 32) >= 256) ? 4 : 18;
    }
    if (!bm) {
      const long nt = (N + 127) / 128;
      const long t64 = (long)((M + 63) / 64) * nt;
      // short-K forward 1x1 with one to two 64-row tiles per CU (block 2 at batch 30): 32-row tiles x 8 waves
      // hide the staging latency better (24.6 vs 29.8 us); with K = 640 the 64-row tiles win again
      if (p.kind == 0 && p.taps == 1 && p.C <= 512 && t64 >= 256 && t64 < 512) return 8;
      if (t64 >= 8192) bm = 128;
      else if (t64 >= 256) bm = 64;
      else bm = 32;
    }
    return bm == 128 ? 0 : bm == 64 ? 7 : 2;
  }
  if (!bm) {
    if ((M + 127) / 128 >= 384) bm = 128;
    else if ((M + 63) / 64 >= 512) bm = 64;
    else return (M + 63) / 64 >= 256 ? 11 : 18;     // < 1 workgroup per CU: 8-way K split (2 waves / SIMD)
  }
  return bm == 128 ? 9 : bm == 64 ? 10 : 5;      // single LDS buffer for the bigger tiles (B = 30 and 128 sweeps)
}

}  // namespace

// `force` == 2000 / 2001 sends an eligible 3x3 problem to the big / small halo kernel whatever its size, any other
// non-zero value keeps it on the conv_gemm tiles (tests / bench/kernel_bench.py); DDL_HALO=0 disables both.
static int halo_bm(const ConvGemmParams& p, int force) {
  static const bool enabled = [] { const char* e = ddl_knob("DDL_HALO"); return !(e && e[0] == '0'); }();
  if (!enabled) return 0;
  return conv3x3_halo_bm(p, force);
}

int conv_gemm_grid_x(const ConvGemmParams& p, int force) {
  int bm = halo_bm(p, force);
  if (!bm) bm = kCfg[pick_cfg(p, force >= 2000 ? 0 : force)].bm;
  return (p.M + bm - 1) / bm;
}

void launch_conv_gemm(const ConvGemmParams& p, int force, hipStream_t s) {
  const int bm = halo_bm(p, force);
  if (bm) launch_conv3x3_halo(p, bm, s);
  else launch_by_id(pick_cfg(p, force >= 2000 ? 0 : force), p, s);
}

}  // namespace ddl
