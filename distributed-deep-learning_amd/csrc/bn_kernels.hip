// BatchNorm bookkeeping and the bandwidth-bound glue kernels of the native DenseNet engine
// (gfx950, wave64).  Reference ops: cuDNN BatchNorm fwd/bwd (121 layers), AvgPool2d,
// adaptive_avg_pool2d, torch.cat.
//
// Design notes
//  * DenseNet's norm1 layers all normalise *prefixes of the same concatenated buffer*, and the
//    batch mean/variance of a channel depend only on the data, not on which layer looks at it.
//    Channel statistics are therefore computed ONCE, when the channel is produced (conv epilogue
//    partial sums -> `bn_prep`), and every later layer only derives its own affine
//    (scale = gamma*rstd, shift = beta - mean*scale) from the shared table.  The 58 full-tensor
//    statistics passes of the reference disappear.
//  * `bn_prep` / `bn_bwd_prep` are one workgroup per 32 channels (channels are independent):
//    32 slices x 32 channels reduce the per-tile partials (fp32 lanes, double combine), then the per-channel
//    scalars are produced.  Nothing here touches pixel data.
//  * `rows_op` is the single skeleton for the pixel-touching glue ops: a workgroup owns a tile of
//    rows, each thread a fixed float4 channel column, so per-channel sums stay in registers and
//    one LDS pass combines the row lanes.
#include "conv_params.h"
#include "ddl_kernels.h"
#include "device_utils.h"

namespace ddl {

namespace {

// sum partials[g][which][c] over g for 32 channels with the 32 slices of a 1024-thread workgroup;
// loads are issued 4 deep (independent accumulators) so the table read is not one long dependent chain
constexpr int PREP_SLICES = 32;
__device__ __forceinline__ void reduce_partials32(const float* __restrict__ partials, int G, int ldp,
                                                  int c_local0, double* s_a, double* s_b,
                                                  double (*red)[2][32]) {
  const int ch = threadIdx.x & 31, slice = threadIdx.x >> 5;
  const float* base = partials + c_local0 + ch;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
  int g = slice;
  for (; g + 3 * PREP_SLICES < G; g += 4 * PREP_SLICES) {
    a0 += base[((long)g * 2 + 0) * ldp];
    b0 += base[((long)g * 2 + 1) * ldp];
    a1 += base[((long)(g + PREP_SLICES) * 2 + 0) * ldp];
    b1 += base[((long)(g + PREP_SLICES) * 2 + 1) * ldp];
    a2 += base[((long)(g + 2 * PREP_SLICES) * 2 + 0) * ldp];
    b2 += base[((long)(g + 2 * PREP_SLICES) * 2 + 1) * ldp];
    a3 += base[((long)(g + 3 * PREP_SLICES) * 2 + 0) * ldp];
    b3 += base[((long)(g + 3 * PREP_SLICES) * 2 + 1) * ldp];
  }
  for (; g < G; g += PREP_SLICES) {
    a0 += base[((long)g * 2 + 0) * ldp];
    b0 += base[((long)g * 2 + 1) * ldp];
  }
  red[slice][0][ch] = ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
  red[slice][1][ch] = ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
  __syncthreads();
  if (slice == 0) {
    double a = 0.0, b = 0.0;
    for (int s = 0; s < PREP_SLICES; ++s) { a += red[s][0][ch]; b += red[s][1][ch]; }
    s_a[ch] = a;
    s_b[ch] = b;
  }
  __syncthreads();
}
}  // namespace

// ---------------------------------------------------------------------------------------------
// forward prep.  grid = C/32 workgroups of 256 threads; workgroup j owns channels [32j, 32j+32).
//   train: channels inside [new0, new0+n_new) first get their batch statistics from the conv
//          epilogue partials (sum, sum of squares over `count` pixels) -> chan_mean / chan_var /
//          chan_rstd; then every channel: scale/shift from the shared table, running stats
//          momentum update (unbiased variance, like torch).
//   eval : scale/shift from the running statistics.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) bn_prep_kernel(
    const float* __restrict__ partials, int G, int ldp, int new0, int n_new, double count,
    float* __restrict__ chan_mean, float* __restrict__ chan_var, float* __restrict__ chan_rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ run_mean,
    float* __restrict__ run_var, float* __restrict__ scale, float* __restrict__ shift, int C,
    float momentum, float eps, int train, const long long* __restrict__ acc, int acc_ld, int acc_c0, int acc_n,
    double acc_inv_count) {
  __shared__ double red[PREP_SLICES][2][32];
  __shared__ double s_a[32], s_b[32];
  const int c0 = blockIdx.x * 32;
  const int ch = threadIdx.x & 31;
  const int c = c0 + ch;
  // channels whose sums still sit in the fixed-point accumulators of the producing dense block (its last slice):
  // finalised here instead of by a bn_acc_finalize launch in front of this kernel
  if (train && acc != nullptr && threadIdx.x < 32 && c >= acc_c0 && c < acc_c0 + acc_n && c < C) {
    long long a1 = 0, a2 = 0;
    for (int r = 0; r < kStatReplicas; ++r) {
      a1 += acc[(long)(2 * r) * acc_ld + c];
      a2 += acc[(long)(2 * r + 1) * acc_ld + c];
    }
    const double mean = (double)a1 * acc_inv_count;
    double var = (double)a2 * acc_inv_count - mean * mean;
    if (var < 0.0) var = 0.0;
    chan_mean[c] = (float)mean;
    chan_var[c] = (float)var;
    chan_rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  }
  const bool is_new = train && partials != nullptr && c0 >= new0 && c0 < new0 + n_new;   // uniform
  if (is_new) {
    reduce_partials32(partials, G, ldp, c0 - new0, s_a, s_b, red);
    if (threadIdx.x < 32 && c < C) {
      const double mean = s_a[ch] / count;
      double var = s_b[ch] / count - mean * mean;
      if (var < 0.0) var = 0.0;
      chan_mean[c] = (float)mean;
      chan_var[c] = (float)var;
      chan_rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
  }
  if (threadIdx.x < 32 && c < C && gamma != nullptr) {
    float sc, sh;
    if (train) {
      const float mean = chan_mean[c], var = chan_var[c], rstd = chan_rstd[c];
      sc = gamma[c] * rstd;
      sh = beta[c] - mean * sc;
      if (run_mean != nullptr) {
        const float unbias = count > 1.0 ? (float)(count / (count - 1.0)) : 1.f;
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * var * unbias;
      }
    } else {
      sc = gamma[c] / sqrtf(run_var[c] + eps);
      sh = beta[c] - run_mean[c] * sc;
    }
    scale[c] = sc;
    shift[c] = sh;
  }
}

void launch_bn_prep(const float* partials, int G, int ldp, int new0, int n_new, double count,
                    float* chan_mean, float* chan_var, float* chan_rstd, const float* gamma,
                    const float* beta, float* run_mean, float* run_var, float* scale, float* shift,
                    int C, float momentum, float eps, bool train, const long long* acc, int acc_ld, int acc_c0,
                    int acc_n, hipStream_t s) {
  hipLaunchKernelGGL(bn_prep_kernel, dim3((C + 31) / 32), dim3(1024), 0, s, partials, G, ldp, new0, n_new,
                     count, chan_mean, chan_var, chan_rstd, gamma, beta, run_mean, run_var, scale, shift,
                     C, momentum, eps, train ? 1 : 0, acc, acc_ld, acc_c0, acc_n,
                     count > 0 ? 1.0 / (count * kStatScale) : 0.0);
}

// ---------------------------------------------------------------------------------------------
// fused-statistics bridge (FusedBn, conv_params.h): accumulators -> chan_mean / chan_var / chan_rstd for
// the channels no fused consumer finalises (the last slice of a dense block, read by the next unit).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bn_acc_finalize_kernel(const long long* __restrict__ acc, int acc_ld, int c0,
                                                              int n, double inv_count, float eps,
                                                              float* __restrict__ chan_mean, float* __restrict__ chan_var,
                                                              float* __restrict__ chan_rstd) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = c0 + i;
  long long s1 = 0, s2 = 0;
  for (int r = 0; r < kStatReplicas; ++r) {
    s1 += acc[(long)(2 * r) * acc_ld + c];
    s2 += acc[(long)(2 * r + 1) * acc_ld + c];
  }
  const double mean = (double)s1 * inv_count;
  double var = (double)s2 * inv_count - mean * mean;
  if (var < 0.0) var = 0.0;
  chan_mean[c] = (float)mean;
  chan_var[c] = (float)var;
  chan_rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
}

void launch_bn_acc_finalize(const long long* acc, int acc_ld, int c0, int n, double count, float eps, float* chan_mean,
                            float* chan_var, float* chan_rstd, hipStream_t s) {
  hipLaunchKernelGGL(bn_acc_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, s, acc, acc_ld, c0, n,
                     1.0 / (count * kStatScale), eps, chan_mean, chan_var, chan_rstd);
}

// ---------------------------------------------------------------------------------------------
// backward prep.  partials = per-tile (sum g, sum g*x) of the masked upstream gradient g.
//   dbeta += sum g ;  dgamma += rstd * (sum g*x - mean * sum g)
//   a = sum g / M ; b = dgamma_batch / M ; BN backward is  dx = sc*g + k1*x + k2  with
//   k1 = -sc*b*rstd ,  k2 = -sc*a + sc*b*rstd*mean.
//   mode 0: write (k0 = sc, k1, k2) - consumed by the next kernel's two-source prologue
//   mode 1: accumulate the deferred dense-block terms  NQ += k1 , NP += k2  (dX = dBuf + NQ*x + NP)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) bn_bwd_prep_kernel(
    const float* __restrict__ partials, int G, int ldp, double count, const float* __restrict__ chan_mean,
    const float* __restrict__ chan_rstd, const float* __restrict__ scale, float* __restrict__ dgamma,
    float* __restrict__ dbeta, float* __restrict__ k0, float* __restrict__ k1, float* __restrict__ k2,
    int C, int mode) {
  __shared__ double red[PREP_SLICES][2][32];
  __shared__ double s_a[32], s_b[32];
  const int c0 = blockIdx.x * 32;
  const int ch = threadIdx.x & 31;
  const int c = c0 + ch;
  // per-channel inputs are fetched ahead of the table reduction (a second dependent round trip otherwise)
  const bool fin = threadIdx.x < 32 && c < C;
  double mean = 0.0, rstd = 0.0, sc = 0.0;
  float db_old = 0.f, dg_old = 0.f, k1_old = 0.f, k2_old = 0.f;
  if (fin) {
    mean = chan_mean[c]; rstd = chan_rstd[c]; sc = scale[c];
    db_old = dbeta[c]; dg_old = dgamma[c];
    if (mode != 0) { k1_old = k1[c]; k2_old = k2[c]; }
  }
  reduce_partials32(partials, G, ldp, c0, s_a, s_b, red);
  if (fin) {
    const double sg = s_a[ch], sgx = s_b[ch];
    const double dgam = rstd * (sgx - mean * sg);
    dbeta[c] = db_old + (float)sg;
    dgamma[c] = dg_old + (float)dgam;
    const double a = sg / count, b = dgam / count;
    const double kk1 = -sc * b * rstd;
    const double kk2 = -sc * a + sc * b * rstd * mean;
    if (mode == 0) {
      k0[c] = (float)sc;
      k1[c] = (float)kk1;
      k2[c] = (float)kk2;
    } else {
      k1[c] = k1_old + (float)kk1;   // NQ
      k2[c] = k2_old + (float)kk2;   // NP
    }
  }
}

void launch_bn_bwd_prep(const float* partials, int G, int ldp, double count, const float* chan_mean,
                        const float* chan_rstd, const float* scale, float* dgamma, float* dbeta, float* k0,
                        float* k1, float* k2, int C, int mode, hipStream_t s) {
  hipLaunchKernelGGL(bn_bwd_prep_kernel, dim3((C + 31) / 32), dim3(1024), 0, s, partials, G, ldp, count,
                     chan_mean, chan_rstd, scale, dgamma, dbeta, k0, k1, k2, C, mode);
}

// ---------------------------------------------------------------------------------------------
// rows_op: pixel-touching glue with optional per-channel partial sums.
//   workgroup = ROWS_PER_WG rows x all C channels; thread = one float4 channel column
//   (C/4 columns, floor(256/(C/4)) row lanes - the threads past lanes*C/4 idle; any C % 4 == 0 with C/4 <= 256:
//   a pipeline stage that starts inside a dense block copies in 96, 160, ... channels).
// modes:
//   0 COPY_STATS : out[m*ldo+c] = in[m*ldi+c];                      stats (v, v*v)
//   1 POOL_FWD   : out[mp][c] = 0.25 * sum_{2x2} relu(sc*x+sh)      (rows = pooled pixels)
//   2 POOL_BWD   : g = (sc*x+sh>0) ? 0.25*d[mp][c] : 0 ; out[m*ldo+c] = sc*g ; stats (g, g*x)
//   3 GAP_BWD    : g = (sc*x+sh>0) ? d[b][c]/HW : 0    ; out[m*ldo+c] = sc*g ; stats (g, g*x)
//   4 AFFINE_ADD : out[m*ldo+c] = in[m*ldi+c] + sc[c]*x[m*ldx+c] + sh[c]      (no stats)
// ---------------------------------------------------------------------------------------------
struct RowsOpParams {
  int mode;
  long M;            // rows this launch iterates over (pooled rows for POOL_FWD)
  int C;
  const float* in;   // COPY: source ; POOL_BWD / GAP_BWD: upstream gradient d ; DEFER: dBuf
  int ldi;
  const float* x;    // the pre-BN activation (block buffer)
  int ldx;
  const float* sc;   // BN scale / AFFINE_ADD: multiplier of x
  const float* sh;   // BN shift / AFFINE_ADD: constant
  float* out;
  int ldo;
  int H, W;          // spatial size of the un-pooled tensor (pool / gap modes)
  float* stats;      // [gridDim.x][2][C] or nullptr
  long long* acc;    // COPY_STATS alternative: fixed-point accumulators [R][2][acc_ld] (conv_params.h FusedBn) or nullptr
  int acc_ld;
  int rows_per_wg;
};

__global__ void __launch_bounds__(256) rows_op_kernel(const RowsOpParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [lanes][2][C] for the stats combine
  const int cols = p.C >> 2;                // float4 columns
  const int lanes = 256 / cols;             // row lanes
  const int col = threadIdx.x % cols, rl = threadIdx.x / cols;
  const int c = col * 4;
  const long r0 = (long)blockIdx.x * p.rows_per_wg;
  long r1 = r0 + p.rows_per_wg;
  if (r1 > p.M) r1 = p.M;
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (p.sc != nullptr) { sc = ld4(p.sc + c); sh = ld4(p.sh + c); }
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  const int HW = p.H * p.W, Wp = p.W >> 1, HWp = (p.H >> 1) * (p.W >> 1);
  const float inv_hw = HW > 0 ? 1.f / (float)HW : 0.f;
  if (rl < lanes) {
    for (long m = r0 + rl; m < r1; m += lanes) {
      if (p.mode == 0) {
        const f32x4 v = ld4(p.in + m * p.ldi + c);
        st4(p.out + m * p.ldo + c, v);
        s1 += v;
        s2 += v * v;
      } else if (p.mode == 1) {
        const long b = m / HWp;
        const int rem = (int)(m - b * HWp);
        const int yp = rem / Wp, xp = rem - yp * Wp;
        const long base = b * HW + (long)(2 * yp) * p.W + 2 * xp;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const long mm = base + (t >> 1) * p.W + (t & 1);
          f32x4 z = sc * ld4(p.x + mm * p.ldx + c) + sh;
          z.x = fmaxf(z.x, 0.f); z.y = fmaxf(z.y, 0.f); z.z = fmaxf(z.z, 0.f); z.w = fmaxf(z.w, 0.f);
          acc += z;
        }
        st4(p.out + m * p.ldo + c, acc * 0.25f);
      } else if (p.mode == 2 || p.mode == 3) {
        const f32x4 xv = ld4(p.x + m * p.ldx + c);
        const f32x4 z = sc * xv + sh;
        f32x4 d;
        const long b = m / HW;
        if (p.mode == 2) {
          const int rem = (int)(m - b * HW);
          const int y = rem / p.W, xx = rem - y * p.W;
          d = ld4(p.in + (b * HWp + (long)(y >> 1) * Wp + (xx >> 1)) * p.ldi + c) * 0.25f;
        } else {
          d = ld4(p.in + b * p.ldi + c) * inv_hw;
        }
        f32x4 g;
        g.x = z.x > 0.f ? d.x : 0.f; g.y = z.y > 0.f ? d.y : 0.f;
        g.z = z.z > 0.f ? d.z : 0.f; g.w = z.w > 0.f ? d.w : 0.f;
        st4(p.out + m * p.ldo + c, sc * g);
        s1 += g;
        s2 += g * xv;
      } else {
        const f32x4 v = ld4(p.in + m * p.ldi + c) + sc * ld4(p.x + m * p.ldx + c) + sh;
        st4(p.out + m * p.ldo + c, v);
      }
    }
  }
  if (p.stats != nullptr || p.acc != nullptr) {
    if (rl < lanes) {
      st4(smem + (rl * 2 + 0) * p.C + c, s1);
      st4(smem + (rl * 2 + 1) * p.C + c, s2);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * p.C; i += 256) {
      float s = 0.f;
      for (int l = 0; l < lanes; ++l) s += smem[l * 2 * p.C + i];
      if (p.acc != nullptr) {       // the consumer derives the BN statistics itself: no table, no bn_prep launch
        const int which = i >= p.C, cc = i - which * p.C;
        __hip_atomic_fetch_add(p.acc + (long)(2 * (blockIdx.x & (kStatReplicas - 1)) + which) * p.acc_ld + cc,
                               __double2ll_rn((double)s * kStatScale), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        p.stats[(long)blockIdx.x * 2 * p.C + i] = s;
      }
    }
  }
}

static int rows_op_launch(const RowsOpParams& p0, hipStream_t s) {
  RowsOpParams p = p0;
  const int cols = p.C / 4;
  const int lanes = 256 / cols;
  // enough workgroups to fill the chip, bounded partial-table size
  long rows = (p.M + 1023) / 1024;
  if (rows < lanes * 4) rows = lanes * 4;
  rows = (rows + lanes - 1) / lanes * lanes;
  p.rows_per_wg = (int)rows;
  const int grid = (int)((p.M + rows - 1) / rows);
  const size_t lds = (p.stats || p.acc) ? (size_t)lanes * 2 * p.C * sizeof(float) : 0;
  hipLaunchKernelGGL(rows_op_kernel, dim3(grid), dim3(256), lds, s, p);
  return grid;
}

int rows_op_grid(long M, int C) {
  const int lanes = 256 / (C / 4);
  long rows = (M + 1023) / 1024;
  if (rows < lanes * 4) rows = lanes * 4;
  rows = (rows + lanes - 1) / lanes * lanes;
  return (int)((M + rows - 1) / rows);
}

int launch_rows_op(int mode, long M, int C, const float* in, int ldi, const float* x, int ldx,
                   const float* sc, const float* sh, float* out, int ldo, int H, int W, float* stats,
                   long long* acc, int acc_ld, hipStream_t s) {
  RowsOpParams p;
  p.mode = mode; p.M = M; p.C = C; p.in = in; p.ldi = ldi; p.x = x; p.ldx = ldx; p.sc = sc; p.sh = sh;
  p.out = out; p.ldo = ldo; p.H = H; p.W = W; p.stats = stats; p.acc = acc; p.acc_ld = acc_ld; p.rows_per_wg = 0;
  return rows_op_launch(p, s);
}

// ---------------------------------------------------------------------------------------------
// head forward: pooled[b][c] = mean_{hw} relu(sc*x+sh).  grid = (B, C/256), thread = channel.
// ---------------------------------------------------------------------------------------------
// 64 channels x 4 pixel groups per workgroup: the 49 (or H*W) dependent loads of one channel are split over four
// threads and reduced through LDS in a fixed order (12.9 -> ~5 us on the 7x7 maps at batch 30: the old form ran
// 30 x 4 workgroups with one serial chain of H*W loads per thread).
__global__ void __launch_bounds__(256) bnrelu_gap_fwd_kernel(const float* __restrict__ x, int ldx, int HW,
                                                             const float* __restrict__ sc,
                                                             const float* __restrict__ sh,
                                                             float* __restrict__ pooled, int C) {
  __shared__ float part[4][64];
  const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cl;
  const long b = blockIdx.x;
  float acc = 0.f;
  if (c < C) {
    const float s = sc[c], t = sh[c];
    const float* xb = x + b * HW * (long)ldx + c;
#pragma unroll 4
    for (int i = g; i < HW; i += 4) acc += fmaxf(fmaf(s, xb[(long)i * ldx], t), 0.f);
  }
  part[g][cl] = acc;
  __syncthreads();
  if (g == 0 && c < C) pooled[b * C + c] = (((part[0][cl] + part[1][cl]) + part[2][cl]) + part[3][cl]) / (float)HW;
}

void launch_bnrelu_gap_fwd(const float* x, int ldx, int B, int HW, const float* sc, const float* sh,
                           float* pooled, int C, hipStream_t s) {
  hipLaunchKernelGGL(bnrelu_gap_fwd_kernel, dim3(B, (C + 63) / 64), dim3(256), 0, s, x, ldx, HW, sc, sh,
                     pooled, C);
}

}  // namespace ddl
