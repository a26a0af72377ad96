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
#include "conv_params.h"
#include "ddl_kernels.h"

namespace ddl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

template <int BM, int BN, int WM, int WN, int KS, bool TWO, bool KMAJOR>
__global__ void __launch_bounds__(64 * WM * WN * KS) conv_gemm_kernel(const ConvGemmParams p) {
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
  static_assert((BM * Q) % NT == 0 && (BN * Q) % NT == 0 && NT % Q == 0, "slab must divide over the workgroup");
  static_assert(TM >= 1 && TN >= 1, "bad wave layout");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                        // [2][BM][LDA]
  float* Bs = smem + 2 * BM * LDA;         // [2][B_ELEMS]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
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
  const long m0 = (long)tile_x * BM;
  const int n0 = blockIdx.y * BN;
  const int cchunks = p.C >> 5;            // 32-wide sub-slabs per tap (C % 32 == 0, checked on host)
  const int NS = p.taps * cchunks;         // sub-slabs in all
  const int KT = (NS + KS - 1) / KS;
  const int HW = p.H * p.W;

  // ---- per-thread bookkeeping -----------------------------------------------------------------
  const int aq = tid % Q;                  // this thread's float4 column inside a slab row
  const int a_sub = aq >> 3;               // which 32-wide sub-slab
  const int a_col = 4 * (aq & 7);          // channel offset inside it
  const int a_row0 = tid / Q;
  long a_m[PA];
  int a_y[PA], a_x[PA];
  bool a_ok[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const long m = m0 + a_row0 + i * RSTEP;
    a_m[i] = m;
    a_ok[i] = m < p.M;
    a_y[i] = 0;
    a_x[i] = 0;
    if (p.taps > 1) {
      const int rem = (int)(m % HW);
      a_y[i] = rem / p.W;
      a_x[i] = rem - a_y[i] * p.W;
    }
  }

  f32x4 a_v0[PA], a_v1[PA];
  f32x4 a_s0, a_s1, a_s2;
  unsigned a_mask = 0;
  f32x4 b_reg[PB];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  auto load_slab = [&](int kt) {
    {   // ---- A: rows gathered with this thread's tap offset, raw values only ----
      const int idx = kt * KS + a_sub;
      const bool sub_ok = idx < NS;
      const int tap = sub_ok ? idx / cchunks : 0;
      const int c = (idx - tap * cchunks) * 32 + a_col;
      int dy = 0, dx = 0;
      if (p.taps > 1) {
        dy = tap / 3 - 1;
        dx = tap - (tap / 3) * 3 - 1;
        if (p.flip) { dy = -dy; dx = -dx; }
      }
      const long off = (long)dy * p.W + dx;
      a_s0 = zero4; a_s1 = zero4; a_s2 = zero4;
      if (sub_ok && p.a.s0 != nullptr) {
        a_s0 = ld4(p.a.s0 + c);
        a_s2 = ld4(p.a.s2 + c);
        if (TWO) a_s1 = ld4(p.a.s1 + c);
      }
      a_mask = 0;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        bool ok = a_ok[i] && sub_ok;
        if (p.taps > 1) {
          const int yy = a_y[i] + dy, xx = a_x[i] + dx;
          ok = ok && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
        }
        a_v0[i] = zero4;
        if (TWO) a_v1[i] = zero4;
        if (ok) {
          a_mask |= 1u << i;
          a_v0[i] = ld4(p.a.X0 + (a_m[i] + off) * p.a.ld0 + c);
          if (TWO) a_v1[i] = ld4(p.a.X1 + (a_m[i] + off) * p.a.ld1 + c);
        }
      }
    }
    // ---- B ----
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int f = tid + i * NT;
      f32x4 v = zero4;
      if (!KMAJOR) {
        const int n = n0 + f / Q;
        const int bq = f % Q;
        const int idx = kt * KS + (bq >> 3);
        if (n < p.N && idx < NS) {
          const int tap = idx / cchunks;
          const int c = (idx - tap * cchunks) * 32 + 4 * (bq & 7);
          v = ld4(p.Wt + (long)n * p.ldb + (long)tap * p.C + c);
        }
      } else {
        const int kk = f / (BN / 4);
        const int n = n0 + 4 * (f - kk * (BN / 4));
        const int idx = kt * KS + (kk >> 5);
        if (n < p.N && idx < NS) {
          const int tap = idx / cchunks;
          const int c = (idx - tap * cchunks) * 32 + (kk & 31);
          v = ld4(p.Wt + (long)c * p.ldb + (long)tap * p.N + n);
        }
      }
      b_reg[i] = v;
    }
  };

  auto store_slab = [&](int buf) {
    float* Ab = As + buf * BM * LDA;
    float* Bb = Bs + buf * B_ELEMS;
    const bool affine = p.a.s0 != nullptr;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      f32x4 v = a_v0[i];
      if (affine) {          // deferred prologue: BN affine (+ second source) + ReLU, zero where invalid
        v = a_s0 * v + a_s2;
        if (TWO) v += a_s1 * a_v1[i];
        if (p.a.relu) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (!((a_mask >> i) & 1u)) v = zero4;
      }
      *reinterpret_cast<f32x4*>(Ab + (a_row0 + i * RSTEP) * LDA + 4 * aq) = v;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int f = tid + i * NT;
      if (!KMAJOR) {
        *reinterpret_cast<f32x4*>(Bb + (f / Q) * LDA + 4 * (f % Q)) = b_reg[i];
      } else {
        const int kk = f / (BN / 4);
        *reinterpret_cast<f32x4*>(Bb + kk * LDBK + 4 * (f - kk * (BN / 4))) = b_reg[i];
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

  // ---- main loop: register-staged double buffer ---------------------------------------------
  load_slab(0);
  store_slab(0);
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) load_slab(kt + 1);      // global loads fly under this slab's MFMAs
    compute(buf);
    if (kt + 1 < KT) store_slab(buf ^ 1);
    __syncthreads();
  }

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

  // ---- epilogue (wave group ks == 0 holds the result) -----------------------------------------
  float st1[TN], st2[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) { st1[j] = 0.f; st2[j] = 0.f; }
  if (ks == 0) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + li;
      const bool colok = col < p.N;
      float e0v = 1.f, e1v = 0.f;
      if (colok && p.e0 != nullptr) e0v = p.e0[col];
      if (colok && p.e1 != nullptr) e1v = p.e1[col];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        // batch every global read of the tile (mask operand, old dBuf value) ahead of the stores:
        // interleaved load/store pairs would serialise 16 dependent L2 round trips per tile
        float axv[16], oldv[16];
        const long rbase = m0 + (wm * TM + i) * 32 + 4 * lh;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long row = rbase + (r & 3) + 8 * (r >> 2);
          axv[r] = 0.f;
          oldv[r] = 0.f;
          if (colok && row < p.M) {
            if (p.ep_mask) axv[r] = p.aux[row * p.ldaux + col];
            if (p.ep_accum) oldv[r] = p.out[row * p.ldo + col];
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long row = rbase + (r & 3) + 8 * (r >> 2);
          if (colok && row < p.M) {
            float v = acc[i][j][r];
            const float ax = axv[r];
            if (p.ep_mask) v = (fmaf(e0v, ax, e1v) > 0.f) ? v : 0.f;
            if (p.stats == 1) { st1[j] += v; st2[j] += v * v; }
            else if (p.stats == 2) { st1[j] += v; st2[j] += v * ax; }
            p.out[row * p.ldo + col] = p.ep_accum ? fmaf(e0v, v, oldv[r]) : v;
          }
        }
      }
    }
  }
  if (p.stats) {
    float* red = smem;   // [WM][2][BN]; every wave is past the last barrier above
    if (ks == 0) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float a = st1[j] + __shfl_xor(st1[j], 32, 64);
        float b = st2[j] + __shfl_xor(st2[j], 32, 64);
        if (lh == 0) {
          const int c = (wn * TN + j) * 32 + li;
          red[(wm * 2 + 0) * BN + c] = a;
          red[(wm * 2 + 1) * BN + c] = b;
        }
      }
    }
    __syncthreads();
    for (int c = tid; c < 2 * BN; c += NT) {
      const int which = c / BN, cc = c - which * BN;
      if (n0 + cc < p.N) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) s += red[(w * 2 + which) * BN + cc];
        p.stats_out[((long)tile_x * 2 + which) * p.N + n0 + cc] = s;
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, int KS, bool TWO, bool KMAJOR>
void launch_inst(const ConvGemmParams& p, hipStream_t s) {
  constexpr int BKT = 32 * KS;
  constexpr size_t b_elems = KMAJOR ? (size_t)BKT * (BN + 4) : (size_t)BN * (BKT + 4);
  constexpr size_t lds = (2 * (size_t)BM * (BKT + 4) + 2 * b_elems) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<BM, BN, WM, WN, KS, TWO, KMAJOR>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  dim3 grid((unsigned)((p.M + BM - 1) / BM), (unsigned)((p.N + BN - 1) / BN));
  hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, KS, TWO, KMAJOR>), grid, dim3(64 * WM * WN * KS), lds, s, p);
}

template <int BM, int BN, int WM, int WN, int KS>
void launch_cfg(const ConvGemmParams& p, hipStream_t s) {
  const bool two = p.a.X1 != nullptr, km = p.b_kmajor != 0;
  if (two && km) launch_inst<BM, BN, WM, WN, KS, true, true>(p, s);
  else if (two) launch_inst<BM, BN, WM, WN, KS, true, false>(p, s);
  else if (km) launch_inst<BM, BN, WM, WN, KS, false, true>(p, s);
  else launch_inst<BM, BN, WM, WN, KS, false, false>(p, s);
}

// Row-tile height.  Measured on MI355X (bench/kernel_bench.py, B=30 shapes): for the 128-column
// configs 64-row tiles beat 128-row tiles up to ~1500 row tiles (finer granularity = smaller tail
// when tiles are dealt over 512 resident workgroup slots); the 32-column 3x3 forward prefers 128.
int pick_bm(int M, int N, int force_bm) {
  if (force_bm) return force_bm;
  if (N > 32) {
    const long nt = (N + 127) / 128;
    if ((long)((M + 63) / 64) * nt >= 8192) return 128;
    if ((long)((M + 63) / 64) * nt >= 256) return 64;
    return 32;
  }
  if ((M + 127) / 128 >= 384) return 128;
  if ((M + 63) / 64 >= 256) return 64;
  return 32;
}

}  // namespace

int conv_gemm_grid_x(int M, int N, int force_bm) {
  const int bm = pick_bm(M, N, force_bm);
  return (M + bm - 1) / bm;
}

void launch_conv_gemm(const ConvGemmParams& p, int force_bm, hipStream_t s) {
  const int bm = pick_bm(p.M, p.N, force_bm);
  if (p.N > 32) {
    if (bm == 128) launch_cfg<128, 128, 2, 2, 1>(p, s);
    else if (bm == 64) launch_cfg<64, 128, 2, 2, 1>(p, s);
    else launch_cfg<32, 128, 1, 4, 2>(p, s);
  } else {
    if (bm == 128) launch_cfg<128, 32, 4, 1, 1>(p, s);
    else if (bm == 64) launch_cfg<64, 32, 2, 1, 2>(p, s);
    else launch_cfg<32, 32, 1, 1, 4>(p, s);
  }
}

}  // namespace ddl
