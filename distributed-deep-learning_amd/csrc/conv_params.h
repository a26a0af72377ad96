// Parameter blocks of the implicit-GEMM convolution kernels (NHWC, fp32, gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace ddl {

// "Row source": the GEMM operand that walks over pixels (rows m = (b, y, x) of an NHWC tensor).
// Element (m, c) = s0[c] * X0[m*ld0 + c] + s1[c] * X1[m*ld1 + c] + s2[c], then optional ReLU.
//   s0 == nullptr -> plain X0 (identity);  X1 == nullptr -> single source.
// This one form covers: BN+ReLU of the forward pass (s0 = gamma*rstd, s2 = beta - mean*s0),
// the BN-backward correction dh = k0*g + k1*h + k2, and the deferred dense-block gradient
// dY = dBuf - Q*x - P  (see models/native_densenet.py).
struct RowSrc {
  const float* X0;
  const float* X1;
  const float* s0;
  const float* s1;
  const float* s2;
  int ld0;
  int ld1;
  int relu;
  long bytes0;        // extent of the X0 / X1 views in bytes (64-bit: kernels rebase their descriptors per workgroup,
  long bytes1;        //   device_utils.h make_rsrc_rows)
};

// Fused batch-norm statistics (forward, training).  The producing conv adds its per-tile column sums
// to 64-bit fixed-point accumulators (integer atomics are order-independent, so the result stays
// deterministic) and the consuming conv derives scale / shift from them in its own prologue: no
// bn_prep launch sits between the two on the critical path.  Workgroup (0,0) of the consumer also
// writes the tables the backward pass reads and updates the running statistics.
//   * kStatReplicas accumulator copies (row tile index mod 8) keep the atomics of a 700-tile grid from
//     serialising on one cache line; the consumer adds the copies (exact integer sums).
//   * channels below `new0` were finalised by an earlier consumer (dense block: every layer's 1x1 conv
//     sees the whole prefix but only the last 32 channels are new): they come from the float tables.
// Range / resolution of the accumulators: a 63-bit sum at scale 2^28 holds |sum| < 3.4e10 with 3.7e-9 absolute
// resolution per addend - a channel's sum of squares over 1.6M pixels (batch 512 on the 56x56 maps) stays in range
// up to an rms activation of ~146, far above anything a batch-normalised network produces; the per-tile partials
// that are added are fp32 sums of <= 128 rows, so the fixed-point rounding is below their own rounding.
constexpr double kStatScale = 268435456.0;      // 2^28: fixed-point scale of the FORWARD accumulators
// The backward (sum g, sum g*h) accumulators of FusedK hold gradient sums: per-tile partials are routinely below
// 1e-6 in the early blocks, where a 2^-28 quantum would round most of their bits away.  They get their own
// scale: 2^48 -> 3.6e-15 absolute resolution, |sum| < 32768 (a mean-reduced loss keeps every gradient sum
// orders of magnitude below that; a partial beyond the range saturates the conversion, it cannot wrap).
constexpr double kGradStatScale = 281474976710656.0;   // 2^48
constexpr int kStatReplicas = 8;
// epilogue transposition image: row stride in floats (rocprofv3: 5-10 % of the LDS-active cycles are bank
// conflicts at 36, 18 % at 40; kernel times are the same either way)
constexpr int kEpiLd = 36;
constexpr int kMaxAffineC = 1024;               // forward BN prologue: scale / shift of <= 1024 channels staged in LDS
struct FusedBn {
  const long long* acc;     // in: [R][2][acc_ld] sum / sum of squares of the row source's channels (nullptr: off)
  int acc_ld;
  int new0;                 // channels [new0, C) come from `acc`, channels [0, new0) from mean/var/rstd tables
  const float* gamma;
  const float* beta;
  float* run_mean;          // momentum update (may be nullptr)
  float* run_var;
  float* scale_out;         // gamma*rstd, beta - mean*scale: saved for backward
  float* shift_out;
  float* mean;              // channel statistics tables: read below new0, written from new0 on
  float* var;
  float* rstd;
  double inv_count;         // 1 / (pixels * kStatScale)
  float unbias, momentum, eps;
  long long* acc_out;       // out (stats == 3): accumulators of the produced channels, [R][2][acc_out_ld]
  int acc_out_ld;
};

// Fused BN-backward coefficients.  BN backward is dh = k0*g + k1*h + k2 with per-channel k's that depend
// on (sum g, sum g*h) over all pixels.  The data-gradient kernel that produces g adds those sums to
// accumulators (as above); its consumers - the 1x1 data gradient and the 1x1 weight gradient - derive the
// k's in their prologues (device_utils.h: fused_k), so no bn_bwd_prep launch separates them.
struct FusedK {
  const long long* acc;     // [R][2][acc_ld]: sum g, sum g*h   (nullptr: off)
  int acc_ld;
  const float* mean;        // forward statistics of this BN
  const float* rstd;
  const float* scale;       // gamma * rstd
  double inv_count;         // 1 / pixels
  float* dgamma;            // += by workgroup (0,0) of the data-gradient consumer (nullptr: not this kernel's job)
  float* dbeta;
};

// BN-backward bookkeeping in the TAIL of the data-gradient kernel that produced its statistics (kinds 2 / 3 with
// fixed-point accumulators): every workgroup takes a ticket per column tile after its statistics atomics; the
// last arriver of a column tile turns the accumulated (sum g, sum g*x) of its columns into dgamma / dbeta and the
// coefficients of dx = k0*g + k1*x + k2 - what a separate bn_bwd_prep launch did between two kernels of the
// backward chain (2 per dense layer on the critical path; knock-out: 0.64 ms of the 9.0 ms step at batch 30).
// Hand-off = agent-scope release before the ticket, agent-scope acquire in the last arriver; integer sums, so the
// result does not depend on the arrival order.
struct TailPrep {
  unsigned* counters;   // [column tiles] tickets; zero before the launch, reset by the last arriver (nullptr: off)
  int mode;             // 0: k0 / k1 / k2 written;  1: deferred dense-block terms  NQ (k1) += , NP (k2) +=
  int fence;            // 1: agent-scope release / acquire fences around the ticket (comparison only, see conv_common.h)
  double inv_count;     // 1 / pixels
  const float* mean;    // forward statistics of this BN
  const float* rstd;
  const float* scale;   // gamma * rstd
  float* dgamma;        // +=
  float* dbeta;
  float* k0;
  float* k1;
  float* k2;
};

// out[M x N] (op)= rows(A)[M x taps*C] * W
struct ConvGemmParams {
  RowSrc a;
  FusedBn bn;
  FusedK fk;
  TailPrep tail;
  int M, C, N;        // pixels, channels per tap of the row source (GEMM K = taps*C), output channels
  int H, W;           // spatial size (3x3 taps only)
  int taps;           // 1 or 9 (3x3, stride 1, pad 1)
  int flip;           // 1: tap offsets negated (data-gradient of a 3x3 conv)
  // weights: n-major  W[n*ldb + tap*C + c]      (forward: layout [N][taps][C])
  //          k-major  W[c*ldb + tap*N + n]      (dgrad: the same forward tensor read transposed)
  const float* Wt;
  int ldb;
  int b_kmajor;
  // epilogue
  float* out;
  int ldo;            // row stride of out (channel-slice writes into a block buffer: ldo > N)
  int ep_mask;        // 1: g = (e0[n]*aux[m][n] + e1[n] > 0) ? acc : 0   (ReLU backward through BN)
  int ep_accum;       // 1: out += e0[n] * g  (BN1 backward into the shared dBuf);  0: out = g
  const float* aux;
  int ldaux;
  const float* e0;
  const float* e1;
  int stats;          // 0 none; 1: per-column sum(v), sum(v*v); 2: sum(g), sum(g*aux); into stats_out, or bn.acc_out if set
  float* stats_out;   // [gridDim.x][2][N]
  // derived on the host (conv_bindings.cpp: finish_params)
  int kind;           // 0 FWD, 1 TDGRAD, 2 DGRAD3, 3 DGRAD1 (see conv_gemm.hip)
  int bf16;           // 1: bf16 MFMA inputs (rounded at the LDS write), fp32 accumulate
  int inv_cc;         // ceil(2^16 / (C / 32)): sub-slab index -> (tap, channel chunk) without a division
  unsigned wbytes;    // extent of the weight tensor in bytes
  long obytes, auxbytes;               // extents of out / aux in bytes (rebased per workgroup like the sources)
#ifdef DDL_PHASE_PROBE
  long long* dbg;     // [2][16] phase timestamps (100 MHz) of the first and the last workgroup (bench/phase_probe.py)
#endif
};

// dW[N x taps*C] += dO^T[N x M] * rows(A)[M x taps*C]   (weight gradient; split over M, slab output)
struct WgradParams {
  FusedK fk;          // 1x1: dO = k0*X0 + k1*X1 + k2 with the k's derived from accumulators
  RowSrc d;           // dO source, N channels
  RowSrc a;           // activation source, C channels
  int M, C, N;
  int H, W;
  int taps;           // 1 or 9
  int rows_per_wg;    // multiple of 32
  float* slabs;       // [S][N][taps*C]  partial sums, one slab per M-chunk
  int S;
  int sub;            // 1x1: 32-column sub-tiles per workgroup tile (1..4) and the number of channel tiles
  int ctiles;
  int bf16;           // 1: operands rounded to bf16 at the LDS write, v_mfma_f32_32x32x16_bf16 via transposing LDS reads
#ifdef DDL_PHASE_PROBE
  int ko;             // timing knock-outs of the pixel loop (2: no staging, 3: no MFMAs; wrong results) - probe builds only
  long long* dbg;     // [2][16]: per-phase time (100 MHz ticks) summed over the pixel steps of the first / a middle workgroup
#endif
};

// layer-grouped weight gradients (conv_wgrad.hip: wgrad_group_kernel): one entry per dense layer; M, N, the
// leading dimensions, the pixel chunking and the operand form (two dO sources, affine + ReLU activation) are shared
constexpr int kMaxWgradGroup = 24;
struct WgradGroupEntry {
  const float* d0;    // dO sources and their per-channel vectors (dO = ds0*d0 + ds1*d1 + ds2)
  const float* d1;
  const float* ds0;
  const float* ds1;
  const float* ds2;
  const float* a0;    // activation source, as0 / as2 its affine (nullptr: identity)
  const float* as0;
  const float* as2;
  float* slabs;       // [S][N][taps*C]
  int C;
  int ctiles;         // 1x1: ceil(C / 128)
  int block0;         // first workgroup of this entry (filled by the launcher)
  int pad_;
};
struct WgradGroupParams {
  WgradGroupEntry e[kMaxWgradGroup];
  int n;
  int M, N, H, W, taps;
  int S, rows_per_wg;
  int ld_d0, ld_d1, ld_a0;
  int a_relu;
  int bf16;           // as WgradParams::bf16
};

// deferred slab reductions: dW += sum over S slabs, many weight gradients in one launch (conv_wgrad.hip)
constexpr int kMaxSlabEntries = 64;
struct SlabEntry {
  const float* slabs;
  float* dW;
  long total4;        // float4 elements of dW
  int S;
  int block0;         // first workgroup of this entry (filled by the launcher)
};
struct SlabReduceParams {
  SlabEntry e[kMaxSlabEntries];
  int n;
};

}  // namespace ddl
