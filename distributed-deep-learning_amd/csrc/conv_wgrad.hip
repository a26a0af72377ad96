// Weight gradients for gfx950: dW[N x taps*C] = dO^T[N x M] * rows(A)[M x taps*C], fp32 MFMA.
//
// Reference ops: cuDNN conv backward-filter behind every Conv2d of DenseNet-121 (SURVEY 7.4
// hard part 3: "reduces over M = B*H*W (large) into a tiny output => split-K across CUs +
// deterministic reduction").  The reduction axis is the pixel axis, so the grid splits M into
// chunks; every workgroup keeps its whole output tile in MFMA accumulators while it walks its
// chunk 32 pixels at a time, then writes ONE partial slab; `slab_reduce` adds the slabs into
// the flat gradient arena in a fixed order (bit-reproducible, unlike float atomics, and cheaper:
// plain 16 B stores + one streamed read instead of 1.3 TB/s chip-wide atomic adds).
//
// Both operands are produced on the fly by the two-source affine prologue (conv_params.h):
// the activation is re-normalised (BN+ReLU recomputed from the saved pre-BN tensor) and the
// output gradient is corrected (BN backward / deferred dense-block terms) while being staged,
// so neither tensor is ever materialised in HBM.
//
// MFMA mapping (v_mfma_f32_32x32x2_f32): D[i = n][j = c] += sum_m dO[m][n] * A[m][c]; both LDS
// tiles keep the pixel axis as rows, so lane (i, h) reads row 2s+h, column i: 32 consecutive
// floats per half-wave - conflict-free ds_read_b32.
#include "conv_params.h"
#include "ddl_kernels.h"

namespace ddl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// Per-thread operand column state: a thread always stages the same channel quad, so the affine
// vectors are fetched once; pixel data is fetched raw one step ahead and transformed only when it
// is written to LDS (after the current step's MFMAs) - the loads' latency hides under the MFMAs.
struct ColAffine {
  f32x4 s0, s1, s2;
  bool affine, two, relu;
};

__device__ __forceinline__ ColAffine col_affine(const RowSrc& a, int c, bool col_ok) {
  ColAffine k;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  k.s0 = z; k.s1 = z; k.s2 = z;
  k.affine = a.s0 != nullptr;
  k.two = a.X1 != nullptr;
  k.relu = a.relu != 0;
  if (k.affine && col_ok) {
    k.s0 = ld4(a.s0 + c);
    k.s2 = ld4(a.s2 + c);
    if (k.two) k.s1 = ld4(a.s1 + c);
  }
  return k;
}

struct RawPix {
  f32x4 v0, v1;
  bool ok;
};

__device__ __forceinline__ RawPix pix_fetch(const RowSrc& a, long row, int c, bool valid) {
  RawPix r;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  r.v0 = z; r.v1 = z; r.ok = valid;
  if (valid) {
    r.v0 = ld4(a.X0 + row * a.ld0 + c);
    if (a.X1 != nullptr) r.v1 = ld4(a.X1 + row * a.ld1 + c);
  }
  return r;
}

__device__ __forceinline__ f32x4 pix_apply(const RawPix& r, const ColAffine& k) {
  f32x4 v = r.v0;
  if (k.affine) {
    v = k.s0 * v + k.s2;
    if (k.two) v += k.s1 * r.v1;
    if (k.relu) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    if (!r.ok) v = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  return v;
}

constexpr int RK = 32;   // pixels per slab step

// ---------------------------------------------------------------------------------------------
// 1x1: output tile 128 (n) x 128 (c) per workgroup, 2x2 waves of 64x64.
// grid = (c tiles, n tiles, S)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) wgrad1x1_kernel(const WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ds = smem;                     // [2][RK][128]
  float* As = smem + 2 * RK * 128;      // [2][RK][128]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int c0 = blockIdx.x * 128, n0 = blockIdx.y * 128;
  const long mb = (long)blockIdx.z * p.rows_per_wg;
  long me = mb + p.rows_per_wg;
  if (me > p.M) me = p.M;
  const int steps = (int)((me - mb + RK - 1) / RK);

  const int q = 4 * (tid & 31);          // this thread's channel quad in both tiles (f & 31 == tid & 31)
  const ColAffine kd = col_affine(p.d, n0 + q, n0 + q < p.N);
  const ColAffine ka = col_affine(p.a, c0 + q, c0 + q < p.C);
  RawPix d_raw[4], a_raw[4];
  auto load_step = [&](int st) {
    const long mrow = mb + (long)st * RK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + i * 256;
      const long m = mrow + (f >> 5);
      d_raw[i] = pix_fetch(p.d, m, n0 + q, m < me && n0 + q < p.N);
      a_raw[i] = pix_fetch(p.a, m, c0 + q, m < me && c0 + q < p.C);
    }
  };
  auto store_step = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + i * 256;
      *reinterpret_cast<f32x4*>(Ds + buf * RK * 128 + (f >> 5) * 128 + q) = pix_apply(d_raw[i], kd);
      *reinterpret_cast<f32x4*>(As + buf * RK * 128 + (f >> 5) * 128 + q) = pix_apply(a_raw[i], ka);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (steps > 0) {
    load_step(0);
    store_step(0);
  }
  __syncthreads();
  for (int st = 0; st < steps; ++st) {
    const int buf = st & 1;
    if (st + 1 < steps) load_step(st + 1);
    const float* Db = Ds + buf * RK * 128 + wm * 64 + li;
    const float* Ab = As + buf * RK * 128 + wn * 64 + li;
#pragma unroll
    for (int s = 0; s < RK / 2; ++s) {
      const int r = 2 * s + lh;
      const float d0 = Db[r * 128], d1 = Db[r * 128 + 32];
      const float a0 = Ab[r * 128], a1 = Ab[r * 128 + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(d0, a0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d0, a1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(d1, a0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d1, a1, acc[1][1], 0, 0, 0);
    }
    if (st + 1 < steps) store_step(buf ^ 1);
    __syncthreads();
  }
  // slab write: slabs[z][n][c]
  float* slab = p.slabs + (long)blockIdx.z * p.N * p.C;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = c0 + wn * 64 + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (n < p.N && c < p.C) slab[(long)n * p.C + c] = acc[i][j][r];
      }
    }
}

// ---------------------------------------------------------------------------------------------
// 3x3 (stride 1, pad 1), N = 32 output channels, C = 128 input channels per workgroup column
// tile.  grid = (S, 3 (dy), C/128).  Wave w owns channels [32w, 32w+32) for the three dx taps.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) wgrad3x3_kernel(const WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int AR = RK + 2;                       // activation rows staged per step (dx halo)
  float* Ds = smem;                                // [2][RK][32]
  float* As = smem + 2 * RK * 32;                  // [2][AR][128]
  int* Fl = reinterpret_cast<int*>(smem + 2 * RK * 32 + 2 * AR * 128);   // [2][RK]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int dy = (int)blockIdx.y - 1;
  const int cbase = blockIdx.z * 128;
  const int HW = p.H * p.W;
  const long mb = (long)blockIdx.x * p.rows_per_wg;
  long me = mb + p.rows_per_wg;
  if (me > p.M) me = p.M;
  const int steps = (int)((me - mb + RK - 1) / RK);

  const ColAffine kd = col_affine(p.d, 4 * (tid & 7), true);
  const ColAffine ka = col_affine(p.a, cbase + 4 * (tid & 31), cbase + 4 * (tid & 31) < p.C);
  RawPix d_raw, a_raw[5];
  int fl_reg = 0;
  auto load_step = [&](int st) {
    const long mrow = mb + (long)st * RK;
    {   // dO tile: 32 rows x 32 channels = 256 float4
      const long m = mrow + (tid >> 3);
      d_raw = pix_fetch(p.d, m, 4 * (tid & 7), m < me);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {   // activation rows mrow-1+dy*W ... (+AR), 128 channels
      const int f = tid + i * 256;
      const int r = f >> 5;
      const long ms = mrow - 1 + (long)dy * p.W + r;
      a_raw[i] = pix_fetch(p.a, ms, cbase + 4 * (f & 31),
                           r < AR && ms >= 0 && ms < p.M && cbase + 4 * (f & 31) < p.C);
    }
    if (tid < RK) {
      const long m = mrow + tid;
      int fl = 0;
      if (m < me) {
        const int rem = (int)(m % HW);
        const int y = rem / p.W, x = rem - y * p.W;
        if (y + dy >= 0 && y + dy < p.H) {
          if (x - 1 >= 0) fl |= 1;
          fl |= 2;
          if (x + 1 < p.W) fl |= 4;
        }
      }
      fl_reg = fl;
    }
  };
  auto store_step = [&](int buf) {
    *reinterpret_cast<f32x4*>(Ds + buf * RK * 32 + (tid >> 3) * 32 + 4 * (tid & 7)) = pix_apply(d_raw, kd);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int f = tid + i * 256;
      if ((f >> 5) < AR)
        *reinterpret_cast<f32x4*>(As + buf * AR * 128 + (f >> 5) * 128 + 4 * (f & 31)) = pix_apply(a_raw[i], ka);
    }
    if (tid < RK) Fl[buf * RK + tid] = fl_reg;
  };

  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  if (steps > 0) {
    load_step(0);
    store_step(0);
  }
  __syncthreads();
  for (int st = 0; st < steps; ++st) {
    const int buf = st & 1;
    if (st + 1 < steps) load_step(st + 1);
    const float* Db = Ds + buf * RK * 32 + li;
    const float* Ab = As + buf * AR * 128 + wid * 32 + li;
    const int* Fb = Fl + buf * RK;
#pragma unroll
    for (int s = 0; s < RK / 2; ++s) {
      const int r = 2 * s + lh;
      const float d = Db[r * 32];
      const int fl = Fb[r];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        float a = Ab[(r + t) * 128];
        a = ((fl >> t) & 1) ? a : 0.f;
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(d, a, acc[t], 0, 0, 0);
      }
    }
    if (st + 1 < steps) store_step(buf ^ 1);
    __syncthreads();
  }
  const int KC = 9 * p.C;
  float* slab = p.slabs + (long)blockIdx.x * p.N * KC;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int tap = (dy + 1) * 3 + t;
    const int c = cbase + wid * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (n < p.N && c < p.C) slab[(long)n * KC + (long)tap * p.C + c] = acc[t][r];
    }
  }
}

// dW[e] += sum_s slabs[s][e]: 16 float4 columns x 16 S-slices per workgroup, independent loads
// 4 deep, fixed combine order => deterministic.
__global__ void __launch_bounds__(256) slab_reduce_kernel(const float* __restrict__ slabs, int S,
                                                          long total4, float* __restrict__ dW) {
  __shared__ f32x4 red[16][16];
  const int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const long i = blockIdx.x * 16L + col;
  const f32x4* src = reinterpret_cast<const f32x4*>(slabs);
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  if (i < total4) {
    int k = sl;
    for (; k + 48 < S; k += 64) {
      a0 += src[(long)k * total4 + i];
      a1 += src[(long)(k + 16) * total4 + i];
      a2 += src[(long)(k + 32) * total4 + i];
      a3 += src[(long)(k + 48) * total4 + i];
    }
    for (; k < S; k += 16) a0 += src[(long)k * total4 + i];
  }
  red[sl][col] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (sl == 0 && i < total4) {
    f32x4 s = reinterpret_cast<f32x4*>(dW)[i];
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][col];
    reinterpret_cast<f32x4*>(dW)[i] = s;
  }
}

}  // namespace

// rows per workgroup (multiple of 32) for a weight-gradient launch: as many chunks as it takes
// to fill the chip, but never so many that the slab traffic outweighs the operand traffic.
int wgrad_rows_per_wg(int M, int other_grid, int min_rows) {
  long want_chunks = (768 + other_grid - 1) / other_grid;
  long rows = (M + want_chunks - 1) / want_chunks;
  if (rows < min_rows) rows = min_rows;
  rows = (rows + 31) / 32 * 32;
  return (int)rows;
}

void launch_wgrad(const WgradParams& p, float* dW, hipStream_t s) {
  static bool attr = false;
  const size_t lds1 = (size_t)4 * RK * 128 * sizeof(float);
  const size_t lds3 = (size_t)(2 * RK * 32 + 2 * (RK + 2) * 128 + 2 * RK) * sizeof(float);
  if (!attr) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad1x1_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
    attr = true;
  }
  if (p.taps == 1) {
    dim3 grid((p.C + 127) / 128, (p.N + 127) / 128, p.S);
    hipLaunchKernelGGL(wgrad1x1_kernel, grid, dim3(256), lds1, s, p);
  } else {
    dim3 grid(p.S, 3, (p.C + 127) / 128);
    hipLaunchKernelGGL(wgrad3x3_kernel, grid, dim3(256), lds3, s, p);
  }
  const long total4 = (long)p.N * p.taps * p.C / 4;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((total4 + 15) / 16)), dim3(256), 0, s,
                     p.slabs, p.S, total4, dW);
}

}  // namespace ddl
