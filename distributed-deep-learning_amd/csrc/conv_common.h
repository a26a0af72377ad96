// Pieces shared by the implicit-GEMM convolution kernels (conv_gemm.hip, conv3x3_halo.hip): the
// per-channel prologue tables in LDS and the LDS-transposed epilogue with fused statistics.
#pragma once
#include "conv_params.h"
#include "device_utils.h"

namespace ddl {

// Forward kind (0): BN scale / shift of all C input channels -> LDS, copied from the tables or derived
// from the fused-statistics accumulators (FusedBn); workgroup (0,0) then also writes the backward tables
// and updates the running statistics.  1x1 data-gradient kind (3): k0 / k2 / k1 of the BN backward.
template <int NT, int KIND>
__device__ __forceinline__ void conv_fill_tables(const ConvGemmParams& p, bool affine, float* scs, float* shs,
                                                 float* k1s, int tid) {
    if (!(KIND == 0 || KIND == 3) || !affine) return;
    if (KIND == 3) {
      const bool writer = blockIdx.x == 0 && blockIdx.y == 0 && p.fk.dgamma != nullptr;
      for (int c = tid; c < p.C; c += NT) {
        float k0, k1, k2;
        if (p.fk.acc != nullptr) {
          float dg, db;
          fused_k(p.fk, c, kStatReplicas, 1.0 / kGradStatScale, k0, k1, k2, dg, db);
          if (writer) { p.fk.dgamma[c] += dg; p.fk.dbeta[c] += db; }
        } else {
          k0 = p.a.s0[c]; k1 = p.a.s1[c]; k2 = p.a.s2[c];
        }
        scs[c] = k0; shs[c] = k2; k1s[c] = k1;
      }
      return;
    }
    if (p.bn.acc == nullptr) {
      for (int c = tid; c < p.C; c += NT) { scs[c] = p.a.s0[c]; shs[c] = p.a.s2[c]; }
      return;
    }
    const bool writer = blockIdx.x == 0 && blockIdx.y == 0;
    for (int c = tid; c < p.C; c += NT) {
      float meanf, varf, rstd;
      if (c < p.bn.new0) {                 // finalised by an earlier consumer
        meanf = p.bn.mean[c]; varf = p.bn.var[c]; rstd = p.bn.rstd[c];
      } else {
        long long s1 = 0, s2 = 0;
#pragma unroll
        for (int r = 0; r < kStatReplicas; ++r) {
          s1 += p.bn.acc[(long)(2 * r) * p.bn.acc_ld + c];
          s2 += p.bn.acc[(long)(2 * r + 1) * p.bn.acc_ld + c];
        }
        const double mean = (double)s1 * p.bn.inv_count;
        double var = (double)s2 * p.bn.inv_count - mean * mean;
        if (var < 0.0) var = 0.0;
        meanf = (float)mean; varf = (float)var;
        rstd = __frsqrt_rn(varf + p.bn.eps);                     // + one Newton step: full fp32 accuracy
        rstd = rstd * (1.5f - 0.5f * (varf + p.bn.eps) * rstd * rstd);
        if (writer) { p.bn.mean[c] = meanf; p.bn.var[c] = varf; p.bn.rstd[c] = rstd; }
      }
      const float sc = p.bn.gamma[c] * rstd;
      const float sh = p.bn.beta[c] - meanf * sc;
      scs[c] = sc;
      shs[c] = sh;
      if (writer) {
        p.bn.scale_out[c] = sc;
        p.bn.shift_out[c] = sh;
        if (p.bn.run_mean != nullptr) {
          p.bn.run_mean[c] = (1.f - p.bn.momentum) * p.bn.run_mean[c] + p.bn.momentum * meanf;
          p.bn.run_var[c] = (1.f - p.bn.momentum) * p.bn.run_var[c] + p.bn.momentum * varf * p.bn.unbias;
        }
      }
    }
}

// Epilogue: out (op)= acc with optional ReLU-mask / accumulate, per-column statistics into the per-tile
// table or the fixed-point accumulators.  `ks` != 0 wave groups (K split) hold nothing and only take
// part in the barriers.  NT = threads of the workgroup.
template <int BM, int BN, int WM, int WN, int NT, int KIND, bool RB>
__device__ __forceinline__ void conv_epilogue(const ConvGemmParams& p, f32x16 (&acc)[BM / (32 * WM)][BN / (32 * WN)],
                                              float* smem, int m0, int n0, int tile_x, int tid, int ks, int wmn) {
  constexpr bool EPM = KIND >= 2, EPA = KIND == 3;
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
  const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wm = wmn / WN, wn = wmn - wm * WN;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // ---- epilogue (wave group ks == 0 holds the result) -----------------------------------------
  // The MFMA C/D layout gives a lane 16 values of ONE column (rows strided by the row pitch): stored
  // as is, a 32x32 tile costs 16 scalar stores (+ 32 scalar loads for the masked / accumulating
  // kinds).  Each wave instead transposes its tile through LDS (the staging area is free now) so that
  // a lane owns 4 consecutive columns of 4 rows: 4 dwordx4 loads / stores per tile and operand.
  // Rows >= M and columns >= N fall outside the descriptors (loads give 0, stores are dropped) and
  // their accumulators are exactly 0 (zeroed operand rows / weight columns): no per-element predicates.
  constexpr int TLD = kEpiLd;              // transposition image row stride (floats)
  f32x4 st1[TN], st2[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) { st1[j] = zero4; st2[j] = zero4; }
  const int erow = lane >> 3, ecq = lane & 7;      // epilogue role: row (+8p) and column quad of the tile
  if (ks == 0) {
    // RB (views beyond 2 GiB): descriptors rebased at this tile's first row, offsets relative to it
    constexpr bool rb = RB;
    const int mrel = rb ? 0 : m0;                                            // row offset that stays in the 32-bit offsets
    const rsrc_t rO = make_rsrc_rows(p.out, p.obytes, rb ? m0 : 0, p.ldo, rb);
    const rsrc_t rX = make_rsrc_rows(EPM ? p.aux : p.out, EPM ? p.auxbytes : p.obytes, rb ? m0 : 0, EPM ? p.ldaux : p.ldo, rb);
    float* T = smem + wmn * (32 * TLD);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + 4 * ecq;
      const bool colok = col < p.N;        // N % 4 == 0: a column quad is valid or invalid as a whole
      f32x4 e0v = {1.f, 1.f, 1.f, 1.f}, e1v = zero4;
      if (EPM || EPA) {
        e0v = ld4(p.e0 + (colok ? col : 0));
        if (EPM) e1v = ld4(p.e1 + (colok ? col : 0));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * lh) * TLD + li] = acc[i][j][r];
        const unsigned row0 = (unsigned)(mrel + (wm * TM + i) * 32 + erow);    // relative to the descriptors' first row
        const unsigned obase = colok ? (row0 * (unsigned)p.ldo + (unsigned)col) * 4u : OOB;
        const unsigned xbase = colok ? (row0 * (unsigned)p.ldaux + (unsigned)col) * 4u : OOB;
        f32x4 axv[4], oldv[4], v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {      // every read of the tile (mask operand, old dBuf) ahead of the stores
          axv[q] = EPM ? bld4(rX, xbase + (unsigned)(8 * q * p.ldaux * 4)) : zero4;
          oldv[q] = EPA ? bld4(rO, obase + (unsigned)(8 * q * p.ldo * 4)) : zero4;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = ld4(T + (erow + 8 * q) * TLD + 4 * ecq);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 g = v[q];
          if (EPM) {
            const f32x4 z = e0v * axv[q] + e1v;
            g.x = z.x > 0.f ? g.x : 0.f; g.y = z.y > 0.f ? g.y : 0.f;
            g.z = z.z > 0.f ? g.z : 0.f; g.w = z.w > 0.f ? g.w : 0.f;
            st1[j] += g;
            st2[j] += g * axv[q];
          } else {
            st1[j] += g;
            st2[j] += g * g;
          }
          bst4(rO, obase + (unsigned)(8 * q * p.ldo * 4), EPA ? e0v * g + oldv[q] : g);
        }
      }
    }
  }
  DDL_PROBE(p, 5);
  if (p.stats) {
    // per-lane column-quad sums -> red[which][row group (WM * 8)][BN] -> one sum per column
    constexpr int RG = WM * 8;
    float* red = smem;
    __syncthreads();                       // transposition images are dead
    if (ks == 0) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int c = (wn * TN + j) * 32 + 4 * ecq;
        *reinterpret_cast<f32x4*>(red + (0 * RG + wm * 8 + erow) * BN + c) = st1[j];
        *reinterpret_cast<f32x4*>(red + (1 * RG + wm * 8 + erow) * BN + c) = st2[j];
      }
    }
    __syncthreads();
    for (int c = tid; c < 2 * BN; c += NT) {
      const int which = c / BN, cc = c - which * BN;
      if (n0 + cc < p.N) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < RG; ++g) s += red[(which * RG + g) * BN + cc];
        if (p.bn.acc_out != nullptr) {   // fused statistics: fixed-point integer atomics
          __hip_atomic_fetch_add(p.bn.acc_out + (long)(2 * (tile_x & (kStatReplicas - 1)) + which) * p.bn.acc_out_ld + n0 + cc,
                                 __double2ll_rn((double)s * (KIND >= 2 ? kGradStatScale : kStatScale)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          p.stats_out[((long)tile_x * 2 + which) * p.N + n0 + cc] = s;
        }
      }
    }
  }
  DDL_PROBE(p, 6);
  // ---- BN-backward bookkeeping by the last-arriving workgroup of this column tile (TailPrep) -----------------
  if ((KIND == 2 || KIND == 3) && p.tail.counters != nullptr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's statistics atomics are out
    __syncthreads();
    DDL_PROBE(p, 7);
    int* flag = reinterpret_cast<int*>(smem);
    if (tid == 0) {
      // Everything the last arriver reads from its peers - the statistics accumulators - was written with
      // agent-scope ATOMICS and is read back with agent-scope atomic loads: both are performed at the device's
      // coherence point, never in a per-XCD L2, and every wave waited for the acknowledgement of its atomics
      // (vmcnt(0) above) before this ticket.  So no release / acquire fence is needed - and a release here would
      // write back the XCD's whole L2 (the megabytes of dBuf this kernel just produced) once per workgroup:
      // measured +0.6 ms per step (`fence` = 1 keeps that form for the comparison).
      if (p.tail.fence) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      const unsigned t = __hip_atomic_fetch_add(p.tail.counters + blockIdx.y, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = t == gridDim.x - 1;
      if (last) {
        __hip_atomic_store(p.tail.counters + blockIdx.y, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        if (p.tail.fence) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
      flag[0] = last;
    }
    __syncthreads();
    if (flag[0]) {
      for (int cc = tid; cc < BN; cc += NT) {
        const int c = n0 + cc;
        if (c >= p.N) continue;
        long long s1 = 0, s2 = 0;
#pragma unroll
        for (int r = 0; r < kStatReplicas; ++r) {
          s1 += __hip_atomic_load(p.bn.acc_out + (long)(2 * r) * p.bn.acc_out_ld + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s2 += __hip_atomic_load(p.bn.acc_out + (long)(2 * r + 1) * p.bn.acc_out_ld + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const double sg = (double)s1 * (1.0 / kGradStatScale), sgx = (double)s2 * (1.0 / kGradStatScale);
        const double mean = p.tail.mean[c], rstd = p.tail.rstd[c], sc = p.tail.scale[c];
        const double dgam = rstd * (sgx - mean * sg);
        const double a = sg * p.tail.inv_count, b = dgam * p.tail.inv_count;
        const double kk1 = -sc * b * rstd;
        const double kk2 = -sc * a + sc * b * rstd * mean;
        p.tail.dbeta[c] += (float)sg;
        p.tail.dgamma[c] += (float)dgam;
        if (p.tail.mode == 0) {
          p.tail.k0[c] = (float)sc;
          p.tail.k1[c] = (float)kk1;
          p.tail.k2[c] = (float)kk2;
        } else {
          p.tail.k1[c] += (float)kk1;   // NQ
          p.tail.k2[c] += (float)kk2;   // NP
        }
      }
    }
  }
}

}  // namespace ddl
