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
// Tiling (wave64): workgroup = WM x WN waves, each wave owns TM x TN MFMA tiles of 32x32
// (16 accumulator VGPRs each).  K is walked in 32-wide slabs; A and B slabs are staged
// global -> registers -> LDS ([rows][32+4] fp32, K contiguous, +4 pad => conflict-free
// ds_read_b128 for the 16-lane groups) with the next slab's global loads in flight during the
// current slab's MFMAs (register-staged double buffer, one barrier per slab).  A ds_read_b128
// hands a lane 4 consecutive k; MFMA j of a group takes element j from both operands, so lane
// half h contributes k = 8s + 4h + j - the same permutation on A and B, hence a valid reduction.
// fp32 MFMA issues once per 64 cycles, so LDS and VALU (the fused prologue) are far from
// limiting; what matters is keeping every SIMD's MFMA queue full and not re-reading HBM.
#include "conv_params.h"
#include "ddl_kernels.h"

namespace ddl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 32;
constexpr int LDK = BK + 4;

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// Raw operand fetch and the deferred prologue.  The global loads of slab k+1 are issued before the
// MFMAs of slab k and only *consumed* (affine + ReLU, then the LDS write) after them, so their
// latency hides under the matrix work instead of stalling the wave at the load site.
template <bool TWO>
struct RawRow {
  f32x4 v0, v1, s0, s1, s2;
};

template <bool TWO>
__device__ __forceinline__ void row_src_fetch(RawRow<TWO>& r, const RowSrc& a, long row, int c, bool valid) {
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  r.v0 = z; r.v1 = z; r.s0 = z; r.s1 = z; r.s2 = z;
  if (valid) {
    r.v0 = ld4(a.X0 + row * a.ld0 + c);
    if (a.s0 != nullptr) {
      r.s0 = ld4(a.s0 + c);
      r.s2 = ld4(a.s2 + c);
      if (TWO) {
        r.v1 = ld4(a.X1 + row * a.ld1 + c);
        r.s1 = ld4(a.s1 + c);
      }
    }
  }
}

template <bool TWO>
__device__ __forceinline__ f32x4 row_src_apply(const RawRow<TWO>& r, const RowSrc& a) {
  f32x4 v = r.v0;
  if (a.s0 != nullptr) {        // invalid rows carry s0 = s2 = 0 -> 0
    v = r.s0 * v + r.s2;
    if (TWO) v += r.s1 * r.v1;
    if (a.relu) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
  }
  return v;
}

template <int BM, int BN, int WM, int WN, bool TWO>
__global__ void __launch_bounds__(64 * WM * WN) conv_gemm_kernel(const ConvGemmParams p) {
  constexpr int NT = 64 * WM * WN;
  constexpr int TM = BM / (32 * WM);
  constexpr int TN = BN / (32 * WN);
  constexpr int PA = (BM * 8) / NT;   // float4 loads per thread per A slab
  constexpr int PB = (BN * 8) / NT;
  static_assert((BM * 8) % NT == 0 && (BN * 8) % NT == 0, "slab must divide over the workgroup");
  static_assert(TM >= 1 && TN >= 1, "bad wave layout");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                    // [2][BM][LDK]
  float* Bs = smem + 2 * BM * LDK;     // [2][BN][LDK]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int li = lane & 31, lh = lane >> 5;
  const long m0 = (long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int cchunks = (p.C + BK - 1) / BK;
  const int KT = p.taps * cchunks;
  const int HW = p.H * p.W;

  // ---- per-thread row bookkeeping for the A slab ------------------------------------------
  long a_m[PA];
  int a_y[PA], a_x[PA];
  bool a_ok[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int f = tid + i * NT;
    const long m = m0 + (f >> 3);
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

  RawRow<TWO> a_raw[PA];
  f32x4 b_reg[PB];

  auto load_slab = [&](int kt) {
    const int tap = kt / cchunks;
    const int c0 = (kt - tap * cchunks) * BK;
    int dy = 0, dx = 0;
    if (p.taps > 1) {
      dy = tap / 3 - 1;
      dx = tap - (tap / 3) * 3 - 1;
      if (p.flip) { dy = -dy; dx = -dx; }
    }
    const long off = (long)dy * p.W + dx;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int f = tid + i * NT;
      const int c = c0 + 4 * (f & 7);
      bool ok = a_ok[i] && c < p.C;
      if (p.taps > 1) {
        const int yy = a_y[i] + dy, xx = a_x[i] + dx;
        ok = ok && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
      }
      row_src_fetch<TWO>(a_raw[i], p.a, a_m[i] + off, c, ok);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int f = tid + i * NT;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (!p.b_kmajor) {
        const int n = n0 + (f >> 3);
        const int c = c0 + 4 * (f & 7);
        if (n < p.N && c < p.C) v = ld4(p.Wt + (long)n * p.ldb + (long)tap * p.C + c);
      } else {
        const int kk = f / (BN / 4);
        const int n = n0 + 4 * (f - kk * (BN / 4));
        const int c = c0 + kk;
        if (n < p.N && c < p.C) v = ld4(p.Wt + (long)c * p.ldb + (long)tap * p.N + n);
      }
      b_reg[i] = v;
    }
  };

  auto store_slab = [&](int buf) {
    float* Ab = As + buf * BM * LDK;
    float* Bb = Bs + buf * BN * LDK;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int f = tid + i * NT;
      *reinterpret_cast<f32x4*>(Ab + (f >> 3) * LDK + 4 * (f & 7)) = row_src_apply<TWO>(a_raw[i], p.a);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int f = tid + i * NT;
      if (!p.b_kmajor) {
        *reinterpret_cast<f32x4*>(Bb + (f >> 3) * LDK + 4 * (f & 7)) = b_reg[i];
      } else {
        const int kk = f / (BN / 4);
        const int n = 4 * (f - kk * (BN / 4));
        Bb[(n + 0) * LDK + kk] = b_reg[i].x;
        Bb[(n + 1) * LDK + kk] = b_reg[i].y;
        Bb[(n + 2) * LDK + kk] = b_reg[i].z;
        Bb[(n + 3) * LDK + kk] = b_reg[i].w;
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
    const float* Ab = As + buf * BM * LDK + (wm * TM * 32 + li) * LDK + 4 * lh;
    const float* Bb = Bs + buf * BN * LDK + (wn * TN * 32 + li) * LDK + 4 * lh;
#pragma unroll
    for (int s = 0; s < BK / 8; ++s) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = ld4(Ab + i * 32 * LDK + 8 * s);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = ld4(Bb + j * 32 * LDK + 8 * s);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
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

  // ---- epilogue -----------------------------------------------------------------------------
  float st1[TN], st2[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) { st1[j] = 0.f; st2[j] = 0.f; }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + li;
    const bool colok = col < p.N;
    float e0v = 1.f, e1v = 0.f;
    if (colok && p.e0 != nullptr) e0v = p.e0[col];
    if (colok && p.e1 != nullptr) e1v = p.e1[col];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (colok && row < p.M) {
          float v = acc[i][j][r];
          float ax = 0.f;
          if (p.ep_mask) {
            ax = p.aux[row * p.ldaux + col];
            v = (fmaf(e0v, ax, e1v) > 0.f) ? v : 0.f;
          }
          if (p.stats == 1) { st1[j] += v; st2[j] += v * v; }
          else if (p.stats == 2) { st1[j] += v; st2[j] += v * ax; }
          float* o = p.out + row * p.ldo + col;
          if (p.ep_accum) *o += e0v * v; else *o = v;
        }
      }
    }
  }
  if (p.stats) {
    float* red = smem;   // [WM][2][BN]; all waves are past the loop's last barrier
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
    __syncthreads();
    for (int c = tid; c < 2 * BN; c += NT) {
      const int which = c / BN, cc = c - which * BN;
      if (n0 + cc < p.N) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) s += red[(w * 2 + which) * BN + cc];
        p.stats_out[((long)blockIdx.x * 2 + which) * p.N + n0 + cc] = s;
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, bool TWO>
void launch_cfg2(const ConvGemmParams& p, hipStream_t s) {
  constexpr size_t lds = (size_t)2 * (BM + BN) * LDK * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<BM, BN, WM, WN, TWO>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  dim3 grid((unsigned)((p.M + BM - 1) / BM), (unsigned)((p.N + BN - 1) / BN));
  hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, TWO>), grid, dim3(64 * WM * WN), lds, s, p);
}

template <int BM, int BN, int WM, int WN>
void launch_cfg(const ConvGemmParams& p, hipStream_t s) {
  if (p.a.X1 != nullptr) launch_cfg2<BM, BN, WM, WN, true>(p, s);
  else launch_cfg2<BM, BN, WM, WN, false>(p, s);
}

}  // namespace

int conv_gemm_grid_x(int M, int N, int force_bm) {
  int bm = force_bm;
  if (bm == 0) {
    if (N > 32) {
      const long nt = (N + 127) / 128;
      if ((long)((M + 127) / 128) * nt >= 384) bm = 128;
      else if ((long)((M + 63) / 64) * nt >= 256) bm = 64;
      else bm = 32;
    } else {
      if ((M + 127) / 128 >= 384) bm = 128;
      else if ((M + 63) / 64 >= 256) bm = 64;
      else bm = 32;
    }
  }
  return (M + bm - 1) / bm;
}

void launch_conv_gemm(const ConvGemmParams& p, int force_bm, hipStream_t s) {
  const int gx = conv_gemm_grid_x(p.M, p.N, force_bm);
  const int bm = (p.M + gx - 1) / gx <= 32 ? 32 : ((p.M + gx - 1) / gx <= 64 ? 64 : 128);
  // recompute bm exactly as conv_gemm_grid_x chose it
  int chosen = force_bm;
  if (chosen == 0) {
    for (int cand : {128, 64, 32})
      if ((p.M + cand - 1) / cand == gx) { chosen = cand; break; }
  }
  (void)bm;
  if (p.N > 32) {
    if (chosen == 128) launch_cfg<128, 128, 2, 2>(p, s);
    else if (chosen == 64) launch_cfg<64, 128, 2, 2>(p, s);
    else launch_cfg<32, 128, 1, 4>(p, s);
  } else {
    if (chosen == 128) launch_cfg<128, 32, 4, 1>(p, s);
    else if (chosen == 64) launch_cfg<64, 32, 2, 1>(p, s);
    else launch_cfg<32, 32, 1, 1>(p, s);
  }
}

}  // namespace ddl
