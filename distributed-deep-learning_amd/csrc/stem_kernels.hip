// DenseNet stem and classifier on gfx950 (MI355X), fp32:
//   conv0 7x7 / stride 2 / pad 3 (3 -> 64) as an fp32-MFMA implicit GEMM, norm0 statistics in its
//   epilogue, BN + ReLU + 3x3/2 max-pool in one pass, the matching backward (pool/ReLU/BN masks,
//   conv0 weight gradient), and the 1024 -> num_classes classifier forward / backward.
// Reference ops: torchvision densenet121 `features.conv0/norm0/relu0/pool0` + `classifier`
// behind single.py:297-299 (cuDNN conv / BN / pooling, cuBLAS GEMV there).
//
// Layout trick for the 7x7 stride-2 conv: the image batch is repacked once to fp32 NHWC with the
// channel dimension padded to 4 ("NHWC4", the /255 normalisation of the reference's `Normalize`
// fused in).  For output pixel (oy, ox) and filter row ky the receptive field is then ONE
// contiguous run of 8 pixels x 4 floats = 32 floats starting at input pixel (2*oy+ky-3, 2*ox-3)
// (the 8th pixel and the 4th channel meet zero weights), i.e. exactly one 32-wide MFMA K slab per
// filter row: GEMM K = 7 x 32 = 224, every A-operand load is an aligned float4, and neighbouring
// output pixels' rows are the same memory shifted by 8 floats.  The weight gradient exploits that
// shift: it stages the 40-pixel input run of a 16-output-pixel segment once and reads the im2col
// matrix as an overlapping view (row stride 8 floats) straight out of LDS.
#include "ddl_kernels.h"
#include "device_utils.h"

namespace ddl {

namespace {

constexpr int KP = 224;          // padded GEMM K: 7 filter rows x (8 pixels x 4 channels)
constexpr int NC0 = 64;          // conv0 output channels

// ---------------------------------------------------------------------------------------------
// input repack: uint8 | float NCHW [B,3,H,W] -> fp32 NHWC4 [B,H,W,4]  (uint8: /255, like the
// reference's Normalize); one thread per pixel, plane reads coalesced, one 16 B store.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) stem_pack_kernel(const T* __restrict__ in, float* __restrict__ out,
                                                        int HW, long total_px, float div) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total_px; i += (long)gridDim.x * 256L) {
    const long b = i / HW;
    const long px = i - b * HW;
    const T* src = in + b * 3 * HW + px;
    f32x4 v;
    v.x = (float)src[0] / div;
    v.y = (float)src[HW] / div;
    v.z = (float)src[2L * HW] / div;
    v.w = 0.f;
    st4(out + i * 4, v);
  }
}

// conv0 weight [64][3][7][7] (torch layout, lives in the parameter arena) -> [64][7][8][4] padded
__global__ void __launch_bounds__(256) stem_pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= NC0 * KP) return;
  const int n = i / KP, k = i - n * KP;
  const int ky = k >> 5, kx = (k >> 2) & 7, c = k & 3;
  wp[i] = (kx < 7 && c < 3) ? w[((n * 3 + c) * 7 + ky) * 7 + kx] : 0.f;
}

// ---------------------------------------------------------------------------------------------
// conv0 forward: y0[M0 x 64] = patches[M0 x 224] * Wp^T, + per-channel (sum, sum of squares)
// partials of the output tile (norm0 batch statistics).  128 x 64 tile, 4 waves (2 x 2), each
// wave 2 MFMA tiles of 32x32; 7 K slabs (filter rows), register-staged LDS double buffer.
// ---------------------------------------------------------------------------------------------
constexpr int FBM = 128, FLD = 36;

__global__ void __launch_bounds__(256) stem_conv_fwd_kernel(const float* __restrict__ xp,
                                                            const float* __restrict__ wp,
                                                            float* __restrict__ y0, float* __restrict__ stats,
                                                            int M, int H, int W, int H0, int W0,
                                                            long xp_bytes) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                       // [2][FBM][FLD]
  float* Bs = smem + 2 * FBM * FLD;       // [2][NC0][FLD]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int li = lane & 31, lh = lane >> 5;
  int tile_x;
  {   // XCD-aware tile order (see conv_gemm.hip): neighbouring row tiles share input rows in one L2
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    tile_x = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const long m0 = (long)tile_x * FBM;
  const int aq = tid & 7, a_row0 = tid >> 3;
  // buffer addressing (device_utils.h): invalid rows / padding pixels select the OOB offset
  const int HW0 = H0 * W0;
  const bool rb = needs_rebase(xp_bytes);
  const int bfirst = rb ? (int)(m0 / HW0) : 0;     // descriptor rebased at the first image of this tile (make_rsrc_rows)
  const rsrc_t rX = make_rsrc_rows(xp, xp_bytes, (long)bfirst * H * W, 4, rb);
  unsigned a_base[4];
  int a_iy0[4];
  bool a_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = (int)m0 + a_row0 + 32 * i;
    const int b = m / HW0;
    const int rem = m - b * HW0;
    const int oy = rem / W0, ox = rem - oy * W0;
    const int ix = 2 * ox - 3 + aq;
    a_iy0[i] = 2 * oy - 3;
    a_base[i] = (unsigned)(((b - bfirst) * H + a_iy0[i]) * W + ix) * 16u;      // byte offset of filter row 0 (may wrap: unused then)
    a_ok[i] = m < M && ix >= 0 && ix < W;
  }
  const unsigned row_bytes = (unsigned)W * 16u;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 a_reg[4], b_reg[2];
  auto load_slab = [&](int ky) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int iy = a_iy0[i] + ky;
      a_reg[i] = bld4(rX, (a_ok[i] && iy >= 0 && iy < H) ? a_base[i] + (unsigned)ky * row_bytes : OOB);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = tid + i * 256;
      b_reg[i] = ld4(wp + (f >> 3) * KP + ky * 32 + 4 * (f & 7));
    }
  };
  auto store_slab = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) st4(As + buf * FBM * FLD + (a_row0 + 32 * i) * FLD + 4 * aq, a_reg[i]);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = tid + i * 256;
      st4(Bs + buf * NC0 * FLD + (f >> 3) * FLD + 4 * (f & 7), b_reg[i]);
    }
  };
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  load_slab(0);
  store_slab(0);
  __syncthreads();
  for (int ky = 0; ky < 7; ++ky) {
    const int buf = ky & 1;
    if (ky + 1 < 7) load_slab(ky + 1);
    const float* Ab = As + buf * FBM * FLD + (wm * 64 + li) * FLD + 4 * lh;
    const float* Bb = Bs + buf * NC0 * FLD + (wn * 32 + li) * FLD + 4 * lh;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const f32x4 a0 = ld4(Ab + 8 * s), a1 = ld4(Ab + 32 * FLD + 8 * s), bf = ld4(Bb + 8 * s);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], bf[e], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], bf[e], acc[1], 0, 0, 0);
      }
    }
    if (ky + 1 < 7) store_slab(buf ^ 1);
    __syncthreads();
  }
  // epilogue: store + column statistics
  const int col = wn * 32 + li;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const long rbase = m0 + (wm * 2 + i) * 32 + 4 * lh;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long row = rbase + (r & 3) + 8 * (r >> 2);
      if (row < M) {
        const float v = acc[i][r];
        y0[row * NC0 + col] = v;
        s1 += v;
        s2 += v * v;
      }
    }
  }
  if (stats != nullptr) {
    float* red = smem;   // [2 (wm)][2][64]
    const float a = s1 + __shfl_xor(s1, 32, 64);
    const float b = s2 + __shfl_xor(s2, 32, 64);
    if (lh == 0) {
      red[(wm * 2 + 0) * NC0 + col] = a;
      red[(wm * 2 + 1) * NC0 + col] = b;
    }
    __syncthreads();
    if (tid < 2 * NC0) {
      const int which = tid >> 6, cc = tid & 63;
      stats[((long)tile_x * 2 + which) * NC0 + cc] = red[which * NC0 + cc] + red[(2 + which) * NC0 + cc];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// norm0 + ReLU + max-pool 3x3 / 2 / pad 1.  thread = one pooled pixel x 4 channels.  The window
// position of the maximum (first maximum in row-major scan order, NaN wins - torch semantics)
// is kept as one byte per element for the backward pass.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) stem_pool_fwd_kernel(const float* __restrict__ y0,
                                                            const float* __restrict__ sc,
                                                            const float* __restrict__ sh, float* __restrict__ p,
                                                            uint8_t* __restrict__ idx, long total, int H0, int W0,
                                                            int H1, int W1) {
  const long t = blockIdx.x * 256L + threadIdx.x;
  if (t >= total) return;
  const int c = 4 * (int)(t & 15);
  const long pix = t >> 4;
  const long b = pix / (H1 * W1);
  const int rem = (int)(pix - b * (H1 * W1));
  const int ph = rem / W1, pw = rem - ph * W1;
  const f32x4 scv = ld4(sc + c), shv = ld4(sh + c);
  const float ninf = -__builtin_inff();
  f32x4 best = {ninf, ninf, ninf, ninf};
  unsigned bi0 = 0, bi1 = 0, bi2 = 0, bi3 = 0;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int ih = 2 * ph - 1 + dy;
    if (ih < 0 || ih >= H0) continue;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int iw = 2 * pw - 1 + dx;
      if (iw < 0 || iw >= W0) continue;
      const f32x4 x = ld4(y0 + ((b * H0 + ih) * W0 + iw) * NC0 + c);
      f32x4 v;
      v.x = fmaxf(fmaf(scv.x, x.x, shv.x), 0.f);
      v.y = fmaxf(fmaf(scv.y, x.y, shv.y), 0.f);
      v.z = fmaxf(fmaf(scv.z, x.z, shv.z), 0.f);
      v.w = fmaxf(fmaf(scv.w, x.w, shv.w), 0.f);
      const unsigned t9 = dy * 3 + dx;
      if (v.x > best.x || v.x != v.x) { best.x = v.x; bi0 = t9; }
      if (v.y > best.y || v.y != v.y) { best.y = v.y; bi1 = t9; }
      if (v.z > best.z || v.z != v.z) { best.z = v.z; bi2 = t9; }
      if (v.w > best.w || v.w != v.w) { best.w = v.w; bi3 = t9; }
    }
  }
  st4(p + pix * NC0 + c, best);
  if (idx != nullptr)
    *reinterpret_cast<unsigned*>(idx + pix * NC0 + c) = bi0 | (bi1 << 8) | (bi2 << 16) | (bi3 << 24);
}

// ---------------------------------------------------------------------------------------------
// backward of max-pool + ReLU (+ the per-channel sums BN backward needs):
//   g[m][c] = (sc*y0+sh > 0) ? sum over the (<= 4) pooling windows whose recorded maximum is this
//             pixel of dp[window][c] : 0 ;   stats: (sum g, sum g*y0) per channel per workgroup.
// Same skeleton as rows_op (bn_kernels.hip): thread = float4 channel column, 16 row lanes.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) stem_pool_bwd_kernel(const float* __restrict__ y0,
                                                            const float* __restrict__ sc,
                                                            const float* __restrict__ sh,
                                                            const float* __restrict__ dp,
                                                            const uint8_t* __restrict__ idx,
                                                            float* __restrict__ g0, float* __restrict__ stats,
                                                            long M, int rows_per_wg, int H0, int W0, int H1,
                                                            int W1) {
  __shared__ __attribute__((aligned(16))) float red[16 * 2 * NC0];
  const int col = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = 4 * col;
  const long r0 = (long)blockIdx.x * rows_per_wg;
  long r1 = r0 + rows_per_wg;
  if (r1 > M) r1 = M;
  const f32x4 scv = ld4(sc + c), shv = ld4(sh + c);
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  const int HW0 = H0 * W0;
  for (long m = r0 + rl; m < r1; m += 16) {
    const long b = m / HW0;
    const int rem = (int)(m - b * HW0);
    const int oy = rem / W0, ox = rem - oy * W0;
    const f32x4 xv = ld4(y0 + m * NC0 + c);
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    const int ph_lo = oy >> 1, ph_hi = (oy + 1) >> 1;
    const int pw_lo = ox >> 1, pw_hi = (ox + 1) >> 1;
    for (int ph = ph_lo; ph <= ph_hi; ++ph) {
      if (ph >= H1) break;
      for (int pw = pw_lo; pw <= pw_hi; ++pw) {
        if (pw >= W1) break;
        const unsigned t9 = (unsigned)((oy - (2 * ph - 1)) * 3 + (ox - (2 * pw - 1)));
        const long off = ((b * H1 + ph) * W1 + pw) * NC0 + c;
        const unsigned id4 = *reinterpret_cast<const unsigned*>(idx + off);
        const f32x4 d = ld4(dp + off);
        if ((id4 & 0xffu) == t9) g.x += d.x;
        if (((id4 >> 8) & 0xffu) == t9) g.y += d.y;
        if (((id4 >> 16) & 0xffu) == t9) g.z += d.z;
        if ((id4 >> 24) == t9) g.w += d.w;
      }
    }
    g.x = fmaf(scv.x, xv.x, shv.x) > 0.f ? g.x : 0.f;
    g.y = fmaf(scv.y, xv.y, shv.y) > 0.f ? g.y : 0.f;
    g.z = fmaf(scv.z, xv.z, shv.z) > 0.f ? g.z : 0.f;
    g.w = fmaf(scv.w, xv.w, shv.w) > 0.f ? g.w : 0.f;
    st4(g0 + m * NC0 + c, g);
    s1 += g;
    s2 += g * xv;
  }
  st4(red + (rl * 2 + 0) * NC0 + c, s1);
  st4(red + (rl * 2 + 1) * NC0 + c, s2);
  __syncthreads();
  if (threadIdx.x < 2 * NC0) {
    float s = 0.f;
#pragma unroll
    for (int l = 0; l < 16; ++l) s += red[l * 2 * NC0 + threadIdx.x];
    stats[(long)blockIdx.x * 2 * NC0 + threadIdx.x] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// conv0 weight gradient: dWp[64 x 224] = dY0^T[64 x M0] * patches[M0 x 224], with
// dY0 = k0*g + k1*y0 + k2 (BN backward) formed while staging.  The pixel axis is walked in
// segments of 16 consecutive output pixels of one output row; per segment and filter row the
// 40-pixel input run (160 floats) is staged once and patches[r][k] is read as run[8*r + k].
// A workgroup owns a chunk of segments and all 7 filter rows: wave (nt, kg) accumulates the
// output-channel tile nt for filter rows {0..3} (kg = 0) or {4..6} (kg = 1) - up to 4 MFMA tiles -
// and writes its part of the chunk's slab; a fixed-order reduce adds the slabs into the arena.
// ---------------------------------------------------------------------------------------------
constexpr int SEG = 16, RUNF = 160;

__global__ void __launch_bounds__(256) stem_wgrad_kernel(const float* __restrict__ g0, const float* __restrict__ y0,
                                                         const float* __restrict__ k0, const float* __restrict__ k1,
                                                         const float* __restrict__ k2, const float* __restrict__ xp,
                                                         float* __restrict__ slabs, int nseg, int segs_per_wg,
                                                         int H, int W, int H0, int W0, int XS) {
  __shared__ __attribute__((aligned(16))) float Ds[2][32 * NC0];
  __shared__ __attribute__((aligned(16))) float Rs[2][7 * 2 * RUNF];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int nt = wid & 1, kg = wid >> 1;
  const int seg_b = blockIdx.x * segs_per_wg;
  int seg_e = seg_b + segs_per_wg;
  if (seg_e > nseg) seg_e = nseg;
  const int steps = (seg_e - seg_b + 1) / 2;
  const int segs_per_img = H0 * XS;

  // descriptors rebased at the first image of this workgroup's chunk (conv0's output passes 2 GiB at ~650 images)
  const long nimg = nseg / segs_per_img;
  const long y_bytes = nimg * H0 * W0 * NC0 * 4L;                                  // B*H0*W0*64 floats
  const bool rb = needs_rebase(y_bytes);
  const int b0 = rb ? seg_b / segs_per_img : 0;
  const rsrc_t rG = make_rsrc_rows(g0, y_bytes, (long)b0 * H0 * W0, NC0, rb), rY = make_rsrc_rows(y0, y_bytes, (long)b0 * H0 * W0, NC0, rb);
  const rsrc_t rX = make_rsrc_rows(xp, nimg * H * W * 16L, (long)b0 * H * W, 4, rb);
  const int dq = 4 * (tid & 15);
  const f32x4 c0 = ld4(k0 + dq), c1 = ld4(k1 + dq), c2 = ld4(k2 + dq);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 d_g[2], d_y[2], r_reg[3];
  bool d_ok[2];
  auto load_step = [&](int st) {
    const int s0 = seg_b + 2 * st;
#pragma unroll
    for (int i = 0; i < 2; ++i) {     // dY0 tile: 32 rows x 16 float4
      const int r = (tid >> 4) + 16 * i;          // == segment i of the step, pixel (tid >> 4)
      const int seg = s0 + i;
      {
        const int b = seg / segs_per_img;
        const int rem = seg - b * segs_per_img;
        const int oy = rem / XS, xs = rem - oy * XS;
        const int ox = xs * SEG + (r & 15);
        d_ok[i] = seg < seg_e && ox < W0;
        const unsigned off = d_ok[i] ? (unsigned)((((b - b0) * H0 + oy) * W0 + ox) * NC0 + dq) * 4u : OOB;
        d_g[i] = bld4(rG, off);
        d_y[i] = bld4(rY, off);
      }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {     // input runs: 7 filter rows x 2 segments x 40 pixels
      const int f = tid + i * 256;
      {
        const int ky = f / 80, rem80 = f - ky * 80;
        const int sg = rem80 / 40, j = rem80 - sg * 40;
        const int seg = s0 + sg;
        const int b = seg / segs_per_img;
        const int rem = seg - b * segs_per_img;
        const int oy = rem / XS, xs = rem - oy * XS;
        const int iy = 2 * oy + ky - 3, ix = 2 * xs * SEG - 3 + j;
        const bool ok = f < 7 * 2 * 40 && seg < seg_e && iy >= 0 && iy < H && ix >= 0 && ix < W;
        r_reg[i] = bld4(rX, ok ? (unsigned)(((b - b0) * H + iy) * W + ix) * 16u : OOB);
      }
    }
  };
  auto store_step = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x4 v = zero4;
      if (d_ok[i]) v = c0 * d_g[i] + c1 * d_y[i] + c2;
      st4(&Ds[buf][((tid >> 4) + 16 * i) * NC0 + dq], v);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int f = tid + i * 256;
      if (f < 7 * 2 * 40) st4(&Rs[buf][4 * f], r_reg[i]);     // [ky][sg][40 px][4] == 4*f
    }
  };

  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
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
    const float* Db = &Ds[buf][nt * 32 + li];
    const float* Rb = &Rs[buf][kg * 4 * 2 * RUNF + li];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int r = 2 * s + lh;
      const float d = Db[r * NC0];
      const float* run = Rb + (s >> 3) * RUNF + 8 * (r & 15);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (kg * 4 + t < 7)      // wave-uniform: the second filter-row group has 3 rows
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(d, run[t * 2 * RUNF], acc[t], 0, 0, 0);
      }
    }
    if (st + 1 < steps) store_step(buf ^ 1);
    __syncthreads();
  }
  float* slab = slabs + (long)blockIdx.x * NC0 * KP;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int ky = kg * 4 + t;
    if (ky < 7) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        slab[n * KP + ky * 32 + li] = acc[t][r];
      }
    }
  }
}

// dW0[n][c][ky][kx] += sum_s slabs[s][n][ky][kx][c]; fixed combine order (deterministic)
__global__ void __launch_bounds__(256) stem_wgrad_reduce_kernel(const float* __restrict__ slabs, int S,
                                                                float* __restrict__ dW) {
  __shared__ f32x4 red[16][16];
  constexpr long total4 = NC0 * KP / 4;
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
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][col];
    const int n = (int)(i / 56), rem = (int)(i - n * 56);
    const int ky = rem >> 3, kx = rem & 7;
    if (kx < 7) {
      dW[((n * 3 + 0) * 7 + ky) * 7 + kx] += s.x;
      dW[((n * 3 + 1) * 7 + ky) * 7 + kx] += s.y;
      dW[((n * 3 + 2) * 7 + ky) * 7 + kx] += s.z;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// classifier: logits[b][n] = pooled[b][:] . W[n][:] + bias[n]   (K = 1024, N = num_classes, tiny)
// one workgroup per batch row, one wave per output (strided), wave-shuffle reduction.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ out,
                                                         int K, int N) {
  // eight waves, one output each per pass; every lane keeps four independent partial sums so that its loads are
  // in flight together (the old form: 16 dependent load+fma rounds per output, two outputs on wave 0 at N = 5)
  const int b = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const float* xr = x + (long)b * K;
  for (int n = wid; n < N; n += 8) {
    const float* wr = w + (long)n * K;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int k = lane;
    for (; k + 192 < K; k += 256) {
      a0 = fmaf(xr[k], wr[k], a0);
      a1 = fmaf(xr[k + 64], wr[k + 64], a1);
      a2 = fmaf(xr[k + 128], wr[k + 128], a2);
      a3 = fmaf(xr[k + 192], wr[k + 192], a3);
    }
    for (; k < K; k += 64) a0 = fmaf(xr[k], wr[k], a0);
    float acc = (a0 + a1) + (a2 + a3);
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) out[(long)b * N + n] = acc + (bias != nullptr ? bias[n] : 0.f);
  }
}

// backward: dx[b][k] = sum_n dy[b][n] W[n][k];  dW[n][k] += sum_b dy[b][n] x[b][k];  db[n] += sum_b dy[b][n]
__global__ void __launch_bounds__(256) linear_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                         const float* __restrict__ w, float* __restrict__ dx,
                                                         float* __restrict__ dW, float* __restrict__ db, int B,
                                                         int K, int N) {
  const long t = blockIdx.x * 256L + threadIdx.x;
  if (t < (long)B * K) {
    const int b = (int)(t / K), k = (int)(t - (long)b * K);
    float acc = 0.f;
    for (int n = 0; n < N; ++n) acc = fmaf(dy[b * N + n], w[(long)n * K + k], acc);
    dx[t] = acc;
  }
  if (t < (long)N * K) {
    const int n = (int)(t / K), k = (int)(t - (long)n * K);
    float acc = 0.f;
#pragma unroll 8
    for (int b = 0; b < B; ++b) acc = fmaf(dy[b * N + n], x[(long)b * K + k], acc);
    dW[t] += acc;
  }
  if (t < N && db != nullptr) {
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += dy[b * N + (int)t];
    db[t] += acc;
  }
}

}  // namespace

// ---- launchers -----------------------------------------------------------------------------
void launch_stem_pack(const void* in, bool is_u8, float* out, int B, int H, int W, hipStream_t s) {
  const long total = (long)B * H * W;
  long grid = (total + 255) / 256;
  if (grid > 256 * 16) grid = 256 * 16;
  if (is_u8)
    hipLaunchKernelGGL(stem_pack_kernel<uint8_t>, dim3((int)grid), dim3(256), 0, s,
                       reinterpret_cast<const uint8_t*>(in), out, H * W, total, 255.0f);
  else
    hipLaunchKernelGGL(stem_pack_kernel<float>, dim3((int)grid), dim3(256), 0, s,
                       reinterpret_cast<const float*>(in), out, H * W, total, 1.0f);
}

void launch_stem_pack_weight(const float* w, float* wp, hipStream_t s) {
  hipLaunchKernelGGL(stem_pack_weight_kernel, dim3((NC0 * KP + 255) / 256), dim3(256), 0, s, w, wp);
}

int stem_conv_grid(long M0) { return (int)((M0 + FBM - 1) / FBM); }

void launch_stem_conv_fwd(const float* xp, const float* wp, float* y0, float* stats, long M0, int H, int W,
                          int H0, int W0, hipStream_t s) {
  constexpr size_t lds = (size_t)(2 * FBM * FLD + 2 * NC0 * FLD) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    DDL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_conv_fwd_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = true;
  }
  const long xp_bytes = (M0 / ((long)H0 * W0)) * H * W * 16;
  hipLaunchKernelGGL(stem_conv_fwd_kernel, dim3(stem_conv_grid(M0)), dim3(256), lds, s, xp, wp, y0, stats,
                     (int)M0, H, W, H0, W0, xp_bytes);
}

void launch_stem_pool_fwd(const float* y0, const float* sc, const float* sh, float* p, uint8_t* idx, int B,
                          int H0, int W0, int H1, int W1, hipStream_t s) {
  const long total = (long)B * H1 * W1 * 16;
  hipLaunchKernelGGL(stem_pool_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, y0, sc, sh, p,
                     idx, total, H0, W0, H1, W1);
}

static int stem_pool_bwd_rows(long M0) {
  long rows = (M0 + 1023) / 1024;
  if (rows < 64) rows = 64;
  return (int)((rows + 15) / 16 * 16);
}

int stem_pool_bwd_grid(long M0) {
  const int rows = stem_pool_bwd_rows(M0);
  return (int)((M0 + rows - 1) / rows);
}

void launch_stem_pool_bwd(const float* y0, const float* sc, const float* sh, const float* dp, const uint8_t* idx,
                          float* g0, float* stats, int B, int H0, int W0, int H1, int W1, hipStream_t s) {
  const long M0 = (long)B * H0 * W0;
  const int rows = stem_pool_bwd_rows(M0);
  hipLaunchKernelGGL(stem_pool_bwd_kernel, dim3(stem_pool_bwd_grid(M0)), dim3(256), 0, s, y0, sc, sh, dp, idx, g0,
                     stats, M0, rows, H0, W0, H1, W1);
}

// chunking of the weight-gradient's segment axis: ~768 workgroups, an even number of segments each
static int stem_wgrad_segs_per_wg(int nseg) {
  int per = (nseg + 767) / 768;
  if (per < 2) per = 2;
  return (per + 1) / 2 * 2;
}

int stem_wgrad_slabs(int B, int H0, int W0) {
  const int nseg = B * H0 * ((W0 + SEG - 1) / SEG);
  const int per = stem_wgrad_segs_per_wg(nseg);
  return (nseg + per - 1) / per;
}

void launch_stem_wgrad(const float* g0, const float* y0, const float* k0, const float* k1, const float* k2,
                       const float* xp, float* slabs, float* dW, int B, int H, int W, int H0, int W0,
                       hipStream_t s) {
  const int XS = (W0 + SEG - 1) / SEG;
  const int nseg = B * H0 * XS;
  const int per = stem_wgrad_segs_per_wg(nseg);
  const int S = (nseg + per - 1) / per;
  hipLaunchKernelGGL(stem_wgrad_kernel, dim3(S), dim3(256), 0, s, g0, y0, k0, k1, k2, xp, slabs, nseg, per, H, W,
                     H0, W0, XS);
  hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3((NC0 * KP / 4 + 15) / 16), dim3(256), 0, s, slabs, S, dW);
}

void launch_linear_fwd(const float* x, const float* w, const float* bias, float* out, int B, int K, int N,
                       hipStream_t s) {
  hipLaunchKernelGGL(linear_fwd_kernel, dim3(B), dim3(512), 0, s, x, w, bias, out, K, N);
}

void launch_linear_bwd(const float* dy, const float* x, const float* w, float* dx, float* dW, float* db, int B,
                       int K, int N, hipStream_t s) {
  const long n = (long)(B > N ? B : N) * K;
  hipLaunchKernelGGL(linear_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dy, x, w, dx, dW, db,
                     B, K, N);
}

}  // namespace ddl
