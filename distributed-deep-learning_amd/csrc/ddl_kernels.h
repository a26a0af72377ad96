// C launchers of the hand-written gfx950 kernels.  No torch headers here: every launcher takes
// raw device pointers and the hipStream_t to enqueue on, never allocates, never synchronises
// (so every launch sequence is hipGraph-capturable).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>

// Experiment switches of the kernels' host side (tile overrides, chunking targets, ...): the defaults are the measured
// optimum, so a variable is honoured only under DDL_EXPERIMENTAL=1 - a stray DDL_* variable in somebody's shell never
// changes the code path of a production run (bench.py lists every DDL_* variable it saw: "experimental_env").
inline const char* ddl_knob(const char* name) {
  static const bool on = [] { const char* e = std::getenv("DDL_EXPERIMENTAL"); return e && e[0] == '1'; }();
  return on ? std::getenv(name) : nullptr;
}

// host-side HIP calls of the launchers (attribute set-up): fail loudly, pybind turns it into a RuntimeError
#define DDL_HIP_CHECK(expr)                                                                              \
  do {                                                                                                   \
    const hipError_t ddl_err_ = (expr);                                                                  \
    if (ddl_err_ != hipSuccess) throw std::runtime_error(std::string(#expr ": ") + hipGetErrorString(ddl_err_)); \
  } while (0)

namespace ddl {

// ---- elementwise.hip -------------------------------------------------------------------
// uint8 NCHW [B,3,H,W] -> fp32 /255, NHWC ([B,H,W,3]) or NCHW.
void launch_normalize_u8(const uint8_t* in, float* out, int B, int C, int H, int W, bool nhwc,
                         hipStream_t s);
// fused softmax + cross-entropy (mean) + d(loss)/d(logits) + argmax; logits [B,C] row-major.
void launch_copy_pair(const void* a_src, void* a_dst, long a_bytes, const void* b_src, void* b_dst, long b_bytes,
                      hipStream_t s);
void launch_ce_fwd_bwd(const float* logits, const int64_t* targets, float* loss /*[1], zeroed here*/,
                       float* dlogits, int64_t* preds, int B, int C, hipStream_t s);
// flat Adam over [start,end) of four parallel arrays; step counter lives on the device.
void launch_adam_inc_step(float* step_t, float* aux /*[2]: 1-b1^t, 1/sqrt(1-b2^t)*/, double b1, double b2,
                          hipStream_t s);
void launch_adam_flat(float* p, float* g, float* m, float* v, const float* aux, int64_t start,
                      int64_t end, double lr, double b1, double b2, double eps, double wd, bool zero_grad,
                      hipStream_t s);
// fence-test helpers: busy-wait for `micros` on the stream (bounded), p[0:n] *= a
void launch_spin(double micros, hipStream_t s);
void launch_scale(float* p, long n, float a, hipStream_t s);

}  // namespace ddl

#include "conv_params.h"

namespace ddl {
// ---- conv_gemm.hip ---------------------------------------------------------------------
int conv_gemm_grid_x(const ConvGemmParams& p, int force_bm);   // number of row tiles (= rows of the stats table)
void launch_conv_gemm(const ConvGemmParams& p, int force_bm, hipStream_t s);
// ---- conv3x3_halo.hip ------------------------------------------------------------------
int conv3x3_halo_bm(const ConvGemmParams& p, int force);       // row-tile height if a halo kernel takes p, else 0
void launch_conv3x3_halo(const ConvGemmParams& p, int bm, hipStream_t s);
// ---- conv_wgrad.hip --------------------------------------------------------------------
int wgrad_rows_per_wg(int M, int other_grid, int min_rows, int target_wgs);
void wgrad1x1_tiling(int M, int C, int N, int force_sub, int& sub, int& ctiles);
void launch_wgrad(const WgradParams& p, hipStream_t s);                      // writes the slabs only
void launch_wgrad_group(WgradGroupParams& g, hipStream_t s);                // <= 24 layers in one launch, slabs only
void launch_slab_reduce_multi(SlabReduceParams& p, hipStream_t s);           // dW += sum of slabs, <= 64 entries
// ---- bn_kernels.hip --------------------------------------------------------------------
void launch_bn_prep(const float* partials, int G, int ldp, int new0, int n_new, double count,
                    float* chan_mean, float* chan_var, float* chan_rstd, const float* gamma,
                    const float* beta, float* run_mean, float* run_var, float* scale, float* shift,
                    int C, float momentum, float eps, bool train, const long long* acc, int acc_ld, int acc_c0,
                    int acc_n, hipStream_t s);
void launch_bn_bwd_prep(const float* partials, int G, int ldp, double count, const float* chan_mean,
                        const float* chan_rstd, const float* scale, float* dgamma, float* dbeta, float* k0,
                        float* k1, float* k2, int C, int mode, hipStream_t s);
void launch_bn_acc_finalize(const long long* acc, int acc_ld, int c0, int n, double count, float eps, float* chan_mean,
                            float* chan_var, float* chan_rstd, hipStream_t s);
int rows_op_grid(long M, int C);
int launch_rows_op(int mode, long M, int C, const float* in, int ldi, const float* x, int ldx,
                   const float* sc, const float* sh, float* out, int ldo, int H, int W, float* stats, long long* acc, int acc_ld,
                   hipStream_t s);
void launch_bnrelu_gap_fwd(const float* x, int ldx, int B, int HW, const float* sc, const float* sh,
                           float* pooled, int C, hipStream_t s);
// ---- stem_kernels.hip ------------------------------------------------------------------
void launch_stem_pack(const void* in, bool is_u8, float* out, int B, int H, int W, hipStream_t s);
void launch_stem_pack_weight(const float* w /*[64,3,7,7]*/, float* wp /*[64,224]*/, hipStream_t s);
int stem_conv_grid(long M0);                      // row tiles (= rows of the statistics table)
void launch_stem_conv_fwd(const float* xp, const float* wp, float* y0, float* stats, long M0, int H, int W,
                          int H0, int W0, hipStream_t s);
void launch_stem_pool_fwd(const float* y0, const float* sc, const float* sh, float* p, uint8_t* idx, int B,
                          int H0, int W0, int H1, int W1, hipStream_t s);
int stem_pool_bwd_grid(long M0);                  // rows of its statistics table
void launch_stem_pool_bwd(const float* y0, const float* sc, const float* sh, const float* dp, const uint8_t* idx,
                          float* g0, float* stats, int B, int H0, int W0, int H1, int W1, hipStream_t s);
int stem_wgrad_slabs(int B, int H0, int W0);      // slabs [S][64][224] the caller must provide
void launch_stem_wgrad(const float* g0, const float* y0, const float* k0, const float* k1, const float* k2,
                       const float* xp, float* slabs, float* dW /*[64,3,7,7], +=*/, int B, int H, int W, int H0,
                       int W0, hipStream_t s);
void launch_linear_fwd(const float* x, const float* w, const float* bias, float* out, int B, int K, int N,
                       hipStream_t s);
void launch_linear_bwd(const float* dy, const float* x, const float* w, float* dx, float* dW, float* db, int B,
                       int K, int N, hipStream_t s);
}  // namespace ddl
