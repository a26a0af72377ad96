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
//
// BF = true (`--dtype bf16`): the same kernels on v_mfma_f32_32x32x16_bf16.  Both operands have the REDUCTION axis
// (pixels) as LDS rows, but the bf16 MFMA wants 8 consecutive k per lane - i.e. a column of the tile.  gfx950's
// `ds_read_b64_tr_b16` does exactly that transposition on the way out of LDS: per 16 lanes a 4-row x 16-column block,
// lane 4q+p supplies the address of row q / columns 4p..4p+3 and receives column (lane % 16) of the four rows.  Two
// such reads give a lane its 8 k-values; tiles are rounded to bf16 (RNE) when they are written, after the fused
// prologue, and stay pixel-major - no transposing store, no second layout.  Row strides are 64 B (mod 256 B) so the
// four rows of a block fall into disjoint bank groups (checked on the device: bench/r3/tr_read_check.hip).
#include <cstdlib>
#include <type_traits>

#include "conv_params.h"
#include "ddl_kernels.h"
#include "device_utils.h"

namespace ddl {

namespace {

// Per-thread operand column state: a thread always stages the same channel quad, so the affine
// vectors are fetched once; pixel data is fetched raw one step ahead and transformed only when it
// is written to LDS (after the current step's MFMAs) - the loads' latency hides under the MFMAs.
struct ColAffine {
  f32x4 s0, s1, s2;
  float lo;      // ReLU as max(v, lo): lo = 0 (ReLU) or -inf (none) - no branch in the staging path
};

__device__ __forceinline__ ColAffine col_affine(const RowSrc& a, int c, bool col_ok) {
  ColAffine k;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  k.s0 = (f32x4){1.f, 1.f, 1.f, 1.f};      // identity when the source has no affine
  k.s1 = z; k.s2 = z;
  k.lo = a.relu != 0 ? 0.f : -__builtin_inff();
  if (a.s0 != nullptr) {
    const int cc = col_ok ? c : 0;
    k.s0 = ld4(a.s0 + cc);
    k.s2 = ld4(a.s2 + cc);
    if (a.X1 != nullptr) k.s1 = ld4(a.s1 + cc);
  }
  return k;
}

// the two descriptors + row strides (bytes) of a row source
struct SrcDesc {
  rsrc_t r0, r1;
  unsigned ld0, ld1;
  int row0;           // the descriptors start at this row (the workgroup's pixel chunk, minus a halo): see make_rsrc_rows
  bool two;
};

__device__ __forceinline__ SrcDesc src_desc(const RowSrc& a, int row0) {
  SrcDesc d;
  d.two = a.X1 != nullptr;
  // always rebased (a few dozen launches per step: the scalar arithmetic does not show), unlike the conv kernels
  d.row0 = row0 > 0 ? row0 : 0;
  d.r0 = make_rsrc_rows(a.X0, a.bytes0, d.row0, a.ld0);
  d.r1 = make_rsrc_rows(d.two ? a.X1 : a.X0, d.two ? a.bytes1 : a.bytes0, d.row0, d.two ? a.ld1 : a.ld0);
  d.ld0 = (unsigned)(a.ld0 * 4);
  d.ld1 = (unsigned)(a.ld1 * 4);
  return d;
}

struct RawPix {
  f32x4 v0, v1;
  bool ok;
};

// TWO is a compile-time property of the call site (a run-time branch between the two loads would
// make hipcc wait for the first before issuing the second)
template <bool TWO>
__device__ __forceinline__ RawPix pix_fetch(const SrcDesc& d, int row, int c, bool valid) {
  RawPix r;
  r.ok = valid;
  r.v0 = bld4(d.r0, valid ? (unsigned)(row - d.row0) * d.ld0 + (unsigned)(c * 4) : OOB);
  r.v1 = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (TWO) r.v1 = bld4(d.r1, valid ? (unsigned)(row - d.row0) * d.ld1 + (unsigned)(c * 4) : OOB);
  return r;
}

template <bool TWO>
__device__ __forceinline__ f32x4 pix_apply(const RawPix& r, const ColAffine& k) {
  f32x4 v = k.s0 * r.v0 + k.s2;
  if (TWO) v += k.s1 * r.v1;
  v.x = fmaxf(v.x, k.lo); v.y = fmaxf(v.y, k.lo); v.z = fmaxf(v.z, k.lo); v.w = fmaxf(v.w, k.lo);
  return r.ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
}

constexpr int RK = 32;   // pixels per slab step

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
// 8 consecutive rows (k) of one column of a row-major bf16 LDS tile, `p` = this lane's block address (see above)
__device__ __forceinline__ bf16x8 lds_tr8(const __bf16* p, int ld) {
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 4 * ld));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
constexpr int bf_lda(int sub) { return 32 * sub + (sub == 4 ? 32 : 0); }   // bf16 row stride of the 1x1 activation tile
constexpr int kBfLdd = 160;                                                // ... of the 1x1 dO tile (128 + 32)

// XCD-aware work order.  Workgroups are dealt round-robin over the 8 XCDs (one L2 each); the T
// workgroups that share a pixel chunk (the channel tiles of the 1x1 kernel, the three filter rows of
// the 3x3 kernel) re-read the same dO / activation rows, so they are given linear ids that agree
// mod 8 and lie next to each other in time: id = lo + 8 * (t + T * hi), chunk = 8 * hi + lo.  The
// second and later readers then hit that XCD's L2 instead of going back to HBM.
__device__ __forceinline__ bool xcd_group(int id, int T, int S, int& t, int& chunk) {
  const int lo = id & 7, rest = id >> 3;
  const int hi = rest / T;
  // An XCD deals its workgroups round-robin over its 32 CUs: the k-th and the (k+32)-th of an XCD share a CU (probe:
  // 234 of 234 pairs, profiles/r3/wgrad_probe/README.md).  Tiles of different cost (a last channel tile with fewer live
  // sub-tiles) would pair up like with like - (96, 96)-column pairs ran 52 us, (64, 64) pairs 42 us - so every pass
  // over the CUs rotates the tile order by one: a CU then holds one of each.
  t = rest - hi * T;
  if (32 % T == 0) t = (t + ((hi * T) >> 5)) % T;       // otherwise k and k + 32 differ mod T already
  chunk = hi * 8 + lo;
  return chunk < S;
}

// ---------------------------------------------------------------------------------------------
// 1x1: output tile 128 (n) x 32*SUB (c) per workgroup.  The four waves split the n axis (32 rows each) and every
// wave walks all SUB 32-column sub-tiles of the tile: whatever C is, the four waves do the same work (the 2x2
// wave layout of round 1 left two waves idle whenever the last channel tile was at most half full - C grows by 32
// per dense layer, so in block 1 that was the common case).  The host picks SUB and the number of channel tiles
// so that the tiles are balanced (C = 160: 96 + 64, not 128 + 32) and, for the small feature maps, so that the
// channel axis - not the pixel axis - supplies the workgroups: the slab traffic is S x N x C x 4 bytes, and with
// 32-column tiles a 14x14 layer needs 4-5x fewer pixel chunks S for the same number of workgroups.
// grid = (c tiles x n tiles x S), XCD-grouped.
// ---------------------------------------------------------------------------------------------
template <int SUB, bool DTWO, int PF, bool BF>
__device__ __forceinline__ void wgrad1x1_body(const WgradParams& p, const int bid, float* smem) {
  constexpr int TC = 32 * SUB;
  constexpr int LDD = kBfLdd, LDA = bf_lda(SUB);
  float* Ds = smem;                     // [2][RK][128]
  float* As = smem + 2 * RK * 128;      // [2][RK][TC]
  __bf16* Dh = reinterpret_cast<__bf16*>(smem);    // BF: [2][RK][LDD]
  __bf16* Ah = Dh + 2 * RK * LDD;                  //     [2][RK][LDA]
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar wave index
  const int li = lane & 31, lh = lane >> 5;
  const int ctiles = p.ctiles, ntiles = (p.N + 127) / 128;
  int t_idx, chunk;
  if (!xcd_group(bid, ctiles * ntiles, p.S, t_idx, chunk)) return;     // grid padded to 8 chunks
  // the C / 32 sub-tiles are dealt evenly over the channel tiles (10 over 3 tiles: 3 + 3 + 4, not 4 + 4 + 2)
  const int nsub = (p.C + 31) / 32, ct = t_idx % ctiles;
  const int sub0 = ct * nsub / ctiles;
  const int nlive = (ct + 1) * nsub / ctiles - sub0;        // live sub-tiles of this tile: 1 ... SUB (host: ctiles * SUB >= nsub)
  const int c0 = sub0 * 32, n0 = (t_idx / ctiles) * 128;
  const int mb = chunk * p.rows_per_wg;
  int me = mb + p.rows_per_wg;
  if (me > p.M) me = p.M;
  const int steps = (me - mb + RK - 1) / RK;
  const SrcDesc dd = src_desc(p.d, mb), da = src_desc(p.a, mb);

  const int q = 4 * (tid & 31);          // dO tile: this thread's channel quad, rows (tid >> 5) + 8 i
  const int ar = tid >> 3;               // activation tile: this thread's row, channel quads 4 (tid & 7) + 32 j
  const int aq = 4 * (tid & 7);
  ColAffine kd = col_affine(p.d, n0 + q, n0 + q < p.N);
  ColAffine ka[SUB];
#pragma unroll
  for (int j = 0; j < SUB; ++j) ka[j] = col_affine(p.a, c0 + 32 * j + aq, c0 + 32 * j + aq < p.C);
  if (p.fk.acc != nullptr) {
    // fused BN backward: dO = k0*g + k1*h + k2 with the k's derived here from the (sum g, sum g*h)
    // accumulators the producing data-gradient kernel filled (no bn_bwd_prep launch in between)
    float* kt = smem;                    // [3][128], dead before the first store_step
    if (tid < 128) {
      float k0 = 0.f, k1 = 0.f, k2 = 0.f, dg, db;
      if (n0 + tid < p.N) fused_k(p.fk, n0 + tid, kStatReplicas, 1.0 / kGradStatScale, k0, k1, k2, dg, db);
      kt[tid] = k0; kt[128 + tid] = k1; kt[256 + tid] = k2;
    }
    __syncthreads();
    kd.s0 = ld4(kt + q); kd.s1 = ld4(kt + 128 + q); kd.s2 = ld4(kt + 256 + q);
    __syncthreads();
  }
  struct Raw { RawPix d[4], a[SUB]; };
  auto load_step = [&](int st, Raw& R) {
    const int mrow = mb + st * RK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mrow + (tid >> 5) + 8 * i;
      R.d[i] = pix_fetch<DTWO>(dd, m, n0 + q, m < me && n0 + q < p.N);
    }
    const int m = mrow + ar;
#pragma unroll
    for (int j = 0; j < SUB; ++j)
      R.a[j] = pix_fetch<false>(da, m, c0 + 32 * j + aq, m < me && j < nlive && c0 + 32 * j + aq < p.C);
  };
  auto store_step = [&](int buf, const Raw& R) {
    if constexpr (BF) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<bf16x4*>(Dh + buf * RK * LDD + ((tid >> 5) + 8 * i) * LDD + q) =
            __builtin_convertvector(pix_apply<DTWO>(R.d[i], kd), bf16x4);
#pragma unroll
      for (int j = 0; j < SUB; ++j)
        *reinterpret_cast<bf16x4*>(Ah + buf * RK * LDA + ar * LDA + 32 * j + aq) =
            __builtin_convertvector(pix_apply<false>(R.a[j], ka[j]), bf16x4);
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<f32x4*>(Ds + buf * RK * 128 + ((tid >> 5) + 8 * i) * 128 + q) = pix_apply<DTWO>(R.d[i], kd);
#pragma unroll
    for (int j = 0; j < SUB; ++j)
      *reinterpret_cast<f32x4*>(As + buf * RK * TC + ar * TC + 32 * j + aq) = pix_apply<false>(R.a[j], ka[j]);
  };

  f32x16 acc[SUB];
#pragma unroll
  for (int j = 0; j < SUB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  // Sub-tiles beyond C (the last channel tile of an uneven split) are skipped.  The number of live sub-tiles NL is a
  // property of the WORKGROUP, so the whole pixel loop is instantiated per NL (`run` below) instead of guarding every
  // MFMA: a wave-uniform `if (live[j])` compiles to a branch around each MFMA, every MFMA sits in its own basic block,
  // and hipcc can neither hoist the operands' LDS reads over the preceding MFMAs nor batch them.  With one or two
  // waves per SIMD nobody else covers that latency: the MFMA + LDS-read part of a step ran at 65 % (one resident
  // workgroup) / 79 % (two) of the matrix pipe's pace with the staging knocked out (profiles/r3/wgrad_probe/README.md).

  // transposing-read block address of this lane: row 8*lh + q, columns 16*(half of li) + 4p
  const int trow = 8 * lh + ((lane & 15) >> 2), tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  auto compute = [&](int buf, auto nl_tag) {
    constexpr int NL = decltype(nl_tag)::value;
    if constexpr (BF) {
#pragma unroll
      for (int ks = 0; ks < RK / 16; ++ks) {
        const bf16x8 d = lds_tr8(Dh + buf * RK * LDD + (16 * ks + trow) * LDD + wid * 32 + tcol, LDD);
        bf16x8 av[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) av[j] = lds_tr8(Ah + buf * RK * LDA + (16 * ks + trow) * LDA + 32 * j + tcol, LDA);
#pragma unroll
        for (int j = 0; j < NL; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d, av[j], acc[j], 0, 0, 0);
      }
      return;
    }
    // one basic block of 16 k-pairs, software-pipelined: the operands of pair s + PD are read from LDS in front of
    // the MFMAs of pair s (PD pairs = ~128-256 cycles of matrix-pipe time ahead; hipcc left to itself sinks the reads
    // back behind the MFMAs, hence the scheduling barriers)
    constexpr int PD = NL >= 3 ? 1 : (NL == 2 ? 2 : 3);
    const float* Db = Ds + buf * RK * 128 + wid * 32 + li + lh * 128;
    const float* Ab = As + buf * RK * TC + li + lh * TC;
    float dq[PD + 1], aq[PD + 1][NL];
#pragma unroll
    for (int g = 0; g < PD; ++g) {
      dq[g] = Db[2 * g * 128];
#pragma unroll
      for (int j = 0; j < NL; ++j) aq[g][j] = Ab[2 * g * TC + 32 * j];
    }
#pragma unroll
    for (int s = 0; s < RK / 2; ++s) {
      if (s + PD < RK / 2) {
        dq[(s + PD) % (PD + 1)] = Db[2 * (s + PD) * 128];
#pragma unroll
        for (int j = 0; j < NL; ++j) aq[(s + PD) % (PD + 1)][j] = Ab[2 * (s + PD) * TC + 32 * j];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NL; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(dq[s % (PD + 1)], aq[s % (PD + 1)][j], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // probe builds (bench/kernel_bench.py --probe): where a workgroup's time goes, summed over its pixel steps
#ifdef DDL_PHASE_PROBE
  long long tp[6] = {0, 0, 0, 0, 0, 0};
  long long tl = (long long)wall_clock64();
  const long long tstart = tl, cstart = (long long)clock64();      // shader-clock cycles beside the 100 MHz wall clock
#define DDL_WPROBE(i) do { const long long n_ = (long long)wall_clock64(); tp[i] += n_ - tl; tl = n_; } while (0)
#define DDL_WPROBE_VMWAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define DDL_WPROBE(i) do {} while (0)
#define DDL_WPROBE_VMWAIT() do {} while (0)
#endif
  auto run = [&](auto nl) {
    if (PF == 1) {
      Raw R;
      if (steps > 0) {
        load_step(0, R);
        store_step(0, R);
      }
      __syncthreads();
      DDL_WPROBE(0);                                    // prologue: tables + first step staged
#pragma unroll 1
      for (int st = 0; st < steps; ++st) {
        const int buf = st & 1;
#ifdef DDL_PHASE_PROBE      // knock-outs (DDL_WGRAD_KO=2: no staging in the loop, 3: no MFMAs) - timing experiments, wrong results
        const bool ko_stage = p.ko == 2, ko_mfma = p.ko == 3;
#else
        constexpr bool ko_stage = false, ko_mfma = false;
#endif
        if (st + 1 < steps && !ko_stage) load_step(st + 1, R);
        DDL_WPROBE(1);                                  // loads issued
        if (!ko_mfma) compute(buf, nl);
        DDL_WPROBE(2);                                  // LDS reads + MFMAs issued
        DDL_WPROBE_VMWAIT();
        DDL_WPROBE(3);                                  // exposed wait for the next step's loads
        if (st + 1 < steps && !ko_stage) store_step(buf ^ 1, R);
        DDL_WPROBE(4);                                  // prologue arithmetic + LDS writes
        __syncthreads();
        DDL_WPROBE(5);                                  // barrier
      }
    } else {
      // prefetch distance 2: step st in LDS, st+1 in registers, st+2 being loaded.  A 32-column tile has only
      // 16 MFMAs (1024 cycles) per wave and step - far less than one trip to HBM / L2 under load, so with a
      // single step in flight the loop runs at memory latency; two steps give every load two compute phases.
      Raw R0, R1;
      if (steps > 0) load_step(0, R0);
      if (steps > 1) load_step(1, R1);
      if (steps > 0) store_step(0, R0);
      __syncthreads();
#pragma unroll 1
      for (int st = 0; st < steps; st += 2) {
        if (st + 2 < steps) load_step(st + 2, R0);
        compute(0, nl);
        if (st + 1 < steps) store_step(1, R1);
        __syncthreads();
        if (st + 1 >= steps) break;
        if (st + 3 < steps) load_step(st + 3, R1);
        compute(1, nl);
        if (st + 2 < steps) store_step(0, R0);
        __syncthreads();
      }
    }
  };
  if (nlive >= SUB) run(std::integral_constant<int, SUB>{});
  else if (SUB > 2 && nlive == SUB - 1) run(std::integral_constant<int, (SUB > 2 ? SUB - 1 : 1)>{});
  else if (SUB > 3 && nlive == SUB - 2) run(std::integral_constant<int, (SUB > 3 ? SUB - 2 : 1)>{});
  else run(std::integral_constant<int, 1>{});
#ifdef DDL_PHASE_PROBE
  if (p.dbg != nullptr && tid == 0 && (bid == 0 || bid == 264)) {
    long long* o = p.dbg + (bid == 0 ? 0 : 16);
    for (int i = 0; i < 6; ++i) o[i] = tp[i];
    o[6] = steps; o[7] = tstart; o[8] = tl; o[9] = (long long)clock64() - cstart;
  }
  if (p.dbg != nullptr && tid == 0 && bid < 2048) {      // (start, end, XCC / CU id) of every workgroup
    p.dbg[32 + 3 * bid] = tstart; p.dbg[33 + 3 * bid] = (long long)wall_clock64();
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    p.dbg[34 + 3 * bid] = ((long long)xcc << 32) | hw;
  }
#endif
  // slab write: slabs[chunk][n][c]
  float* slab = p.slabs + (long)chunk * p.N * p.C;
#pragma unroll
  for (int j = 0; j < SUB; ++j) {
    const int c = c0 + 32 * j + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + wid * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (j < nlive && n < p.N && c < p.C) slab[(long)n * p.C + c] = acc[j][r];
    }
  }
}

template <int SUB, bool DTWO, int PF, bool BF>
__global__ void __launch_bounds__(256, 2) wgrad1x1_kernel(const WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  wgrad1x1_body<SUB, DTWO, PF, BF>(p, (int)blockIdx.x, smem);
}

// ---------------------------------------------------------------------------------------------
// 3x3 (stride 1, pad 1), N = 32 output channels, C = 128 input channels per workgroup column
// tile.  grid = (S, 3 (dy), C/128).  Wave w owns channels [32w, 32w+32) for the three dx taps.
// ---------------------------------------------------------------------------------------------
// BF: the border mask of a tap depends on the PIXEL, which is the k axis a lane holds 8 of - so instead of masking
// operands in the loop the dO tile is staged three times, once per dx tap with that tap's mask applied (3 x 2 KB).
template <bool DTWO, bool BF>
__device__ __forceinline__ void wgrad3x3_body(const WgradParams& p, const int bid, float* smem) {
  constexpr int AR = RK + 2;                       // activation rows staged per step (dx halo)
  constexpr int LDD = 32, LDA = 160;               // BF row strides (elements): 64 B and 320 B
  __bf16* Dh = reinterpret_cast<__bf16*>(smem);    // BF: [2][3][RK][LDD]
  __bf16* Ah = Dh + 2 * 3 * RK * LDD;              //     [2][AR][LDA]
  float* Ds = smem;                                // [2][RK][32]
  float* As = smem + 2 * RK * 32;                  // [2][AR][128]
  int* Fl = reinterpret_cast<int*>(smem + 2 * RK * 32 + 2 * AR * 128);   // [2][RK]
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar wave index
  const int li = lane & 31, lh = lane >> 5;
  const int ctiles = (p.C + 127) / 128;
  int t_idx, chunk;
  if (!xcd_group(bid, 3 * ctiles, p.S, t_idx, chunk)) return;          // grid padded to 8 chunks
#ifdef DDL_PHASE_PROBE
  const long long tstart3 = (long long)wall_clock64(), cstart3 = (long long)clock64();
#endif
  const int dy = t_idx % 3 - 1;
  const int cbase = (t_idx / 3) * 128;
  const int HW = p.H * p.W;
  const int mb = chunk * p.rows_per_wg;
  int me = mb + p.rows_per_wg;
  if (me > p.M) me = p.M;
  const int steps = (me - mb + RK - 1) / RK;
  const SrcDesc dd = src_desc(p.d, mb), da = src_desc(p.a, mb - p.W - 1);      // activation rows reach one image row back

  const ColAffine kd = col_affine(p.d, 4 * (tid & 7), true);
  const ColAffine ka = col_affine(p.a, cbase + 4 * (tid & 31), cbase + 4 * (tid & 31) < p.C);
  RawPix d_raw, a_raw[5];
  int fl_reg = 0;
  auto load_step = [&](int st) {
    const int mrow = mb + st * RK;
    {   // dO tile: 32 rows x 32 channels = 256 float4
      const int m = mrow + (tid >> 3);
      d_raw = pix_fetch<DTWO>(dd, m, 4 * (tid & 7), m < me);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {   // activation rows mrow-1+dy*W ... (+AR), 128 channels
      const int f = tid + i * 256;
      const int r = f >> 5;
      const int ms = mrow - 1 + dy * p.W + r;
      a_raw[i] = pix_fetch<false>(da, ms, cbase + 4 * (f & 31),
                           r < AR && ms >= 0 && ms < p.M && cbase + 4 * (f & 31) < p.C);
    }
    if (BF || tid < RK) {                         // BF: every thread, for the dO row it stages
      const int m = mrow + (BF ? (tid >> 3) : tid);
      int fl = 0;
      if (m < me) {
        const int rem = m % HW;
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
    if constexpr (BF) {
      const f32x4 v = pix_apply<DTWO>(d_raw, kd), z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 3; ++t)
        *reinterpret_cast<bf16x4*>(Dh + ((buf * 3 + t) * RK + (tid >> 3)) * LDD + 4 * (tid & 7)) =
            __builtin_convertvector(((fl_reg >> t) & 1) ? v : z, bf16x4);
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int f = tid + i * 256;
        if ((f >> 5) < AR)
          *reinterpret_cast<bf16x4*>(Ah + (buf * AR + (f >> 5)) * LDA + 4 * (f & 31)) =
              __builtin_convertvector(pix_apply<false>(a_raw[i], ka), bf16x4);
      }
      return;
    }
    *reinterpret_cast<f32x4*>(Ds + buf * RK * 32 + (tid >> 3) * 32 + 4 * (tid & 7)) = pix_apply<DTWO>(d_raw, kd);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int f = tid + i * 256;
      if ((f >> 5) < AR)
        *reinterpret_cast<f32x4*>(As + buf * AR * 128 + (f >> 5) * 128 + 4 * (f & 31)) = pix_apply<false>(a_raw[i], ka);
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
#ifdef DDL_PHASE_PROBE      // knock-outs (DDL_WGRAD_KO=2: no staging in the loop, 3: no MFMAs) - timing experiments, wrong results
    const bool ko_stage = p.ko == 2, ko_mfma = p.ko == 3;
#else
    constexpr bool ko_stage = false, ko_mfma = false;
#endif
    if (st + 1 < steps && !ko_stage) load_step(st + 1);
    if (ko_mfma) {
    } else if constexpr (BF) {
      const int trow = 8 * lh + ((lane & 15) >> 2), tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
#pragma unroll
      for (int ks = 0; ks < RK / 16; ++ks) {
        bf16x8 d[3], a[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          d[t] = lds_tr8(Dh + ((buf * 3 + t) * RK + 16 * ks + trow) * LDD + tcol, LDD);
          a[t] = lds_tr8(Ah + (buf * AR + 16 * ks + trow + t) * LDA + wid * 32 + tcol, LDA);
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d[t], a[t], acc[t], 0, 0, 0);
      }
    } else {
      // software-pipelined like the 1x1 kernel: the operands of k-pair s+1 are read in front of the MFMAs of pair s
      // (left alone hipcc reads each operand right in front of its MFMA behind s_waitcnt lgkmcnt(0)); activation row
      // r + 2 of pair s is row r of pair s + 1, so a pair costs four LDS reads: dO, mask, two activation rows
      const float* Db = Ds + buf * RK * 32 + li + lh * 32;
      const float* Ab = As + buf * AR * 128 + wid * 32 + li + lh * 128;
      const int* Fb = Fl + buf * RK + lh;
      float dq[2], aq[2][3];
      int fq[2];
      dq[0] = Db[0]; fq[0] = Fb[0];
#pragma unroll
      for (int t = 0; t < 3; ++t) aq[0][t] = Ab[t * 128];
#pragma unroll
      for (int s = 0; s < RK / 2; ++s) {
        const int c = s & 1, n = c ^ 1;
        if (s + 1 < RK / 2) {
          dq[n] = Db[2 * (s + 1) * 32];
          fq[n] = Fb[2 * (s + 1)];
          aq[n][0] = aq[c][2];
          aq[n][1] = Ab[(2 * (s + 1) + 1) * 128];
          aq[n][2] = Ab[(2 * (s + 1) + 2) * 128];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const float a = ((fq[c] >> t) & 1) ? aq[c][t] : 0.f;
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(dq[c], a, acc[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (st + 1 < steps && !ko_stage) store_step(buf ^ 1);
    __syncthreads();
  }
#ifdef DDL_PHASE_PROBE
  if (p.dbg != nullptr && tid == 0 && bid < 2048) {      // (start, end, XCC / CU id) of every workgroup
    p.dbg[32 + 3 * bid] = tstart3; p.dbg[33 + 3 * bid] = (long long)wall_clock64();
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    p.dbg[34 + 3 * bid] = ((long long)xcc << 32) | hw;
    if (bid == 0) { p.dbg[6] = steps; p.dbg[7] = tstart3; p.dbg[8] = p.dbg[33]; p.dbg[9] = (long long)clock64() - cstart3; }
  }
#endif
  const int KC = 9 * p.C;
  float* slab = p.slabs + (long)chunk * p.N * KC;
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

template <bool DTWO, bool BF>
__global__ void __launch_bounds__(256) wgrad3x3_kernel(const WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  wgrad3x3_body<DTWO, BF>(p, (int)blockIdx.x, smem);
}

// ---------------------------------------------------------------------------------------------
// Layer-grouped weight gradients.  The weight gradients of a dense layer feed nothing downstream, and everything
// they read (the layer's g2 / h / BN coefficients, its slice of the block's gradient buffer, the block buffer)
// stays valid until the end of the block's backward.  For the small feature maps (14x14, 7x7) one layer's
// weight gradient is a 10-25 us kernel that is all ramp-up, latency and slab traffic (PMC: 16-30 % MFMA
// utilisation, 40-50 % of the wave time waiting); G layers in ONE launch give every workgroup a long pixel loop
// (no 100-way split of 5880 pixels), a few hundred workgroups of real work and G times fewer launches.
// Workgroup -> (layer entry, tile, pixel chunk); entries share M, N, the leading dimensions and the chunking.
// ---------------------------------------------------------------------------------------------
template <int TAPS, bool BF>
__global__ void __launch_bounds__(256, 2) wgrad_group_kernel(const WgradGroupParams g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int e = 0;
  while (e + 1 < g.n && (int)blockIdx.x >= g.e[e + 1].block0) ++e;             // scalar walk over <= 24 entries
  const WgradGroupEntry en = g.e[e];
  const int dch = TAPS == 1 ? g.N : g.N;           // dO channels
  const int ach = en.C;                            // activation channels
  WgradParams p;
  p.fk.acc = nullptr; p.fk.acc_ld = 0; p.fk.mean = nullptr; p.fk.rstd = nullptr; p.fk.scale = nullptr;
  p.fk.inv_count = 0.0; p.fk.dgamma = nullptr; p.fk.dbeta = nullptr;
  p.d.X0 = en.d0; p.d.X1 = en.d1; p.d.s0 = en.ds0; p.d.s1 = en.ds1; p.d.s2 = en.ds2;
  p.d.ld0 = g.ld_d0; p.d.ld1 = g.ld_d1; p.d.relu = 0;
  p.d.bytes0 = (((long)g.M - 1) * g.ld_d0 + dch) * 4;
  p.d.bytes1 = (((long)g.M - 1) * g.ld_d1 + dch) * 4;
  p.a.X0 = en.a0; p.a.X1 = nullptr; p.a.s0 = en.as0; p.a.s1 = nullptr; p.a.s2 = en.as2;
  p.a.ld0 = g.ld_a0; p.a.ld1 = 0; p.a.relu = g.a_relu;
  p.a.bytes0 = (((long)g.M - 1) * g.ld_a0 + ach) * 4;
  p.a.bytes1 = 0;
  p.M = g.M; p.C = en.C; p.N = g.N; p.H = g.H; p.W = g.W; p.taps = TAPS;
  p.rows_per_wg = g.rows_per_wg; p.slabs = en.slabs; p.S = g.S; p.sub = 4; p.ctiles = en.ctiles;
  p.bf16 = g.bf16;
#ifdef DDL_PHASE_PROBE
  p.dbg = nullptr; p.ko = 0;
#endif
  const int bid = (int)blockIdx.x - en.block0;
  if (TAPS == 1) wgrad1x1_body<4, true, 1, BF>(p, bid, smem);
  else wgrad3x3_body<true, BF>(p, bid, smem);
}

// dW[e] += sum_s slabs[s][e] for a whole LIST of weight gradients in one launch: the dense-block backward queues
// (slabs, S, dW) of every layer and reduces them once, at the end of the block (119 slab_reduce launches per step
// in round 1 -> 4 + the transitions).  Per workgroup: 16 float4 columns x 16 S-slices of one entry, independent
// loads 4 deep, fixed combine order => deterministic.
__global__ void __launch_bounds__(256) slab_reduce_multi_kernel(const SlabReduceParams p) {
  __shared__ f32x4 red[16][16];
  int e = 0;
  while (e + 1 < p.n && (int)blockIdx.x >= p.e[e + 1].block0) ++e;           // scalar walk over <= 64 entries
  const SlabEntry en = p.e[e];
  const int S = en.S;
  const long total4 = en.total4;
  const int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const long i = ((long)blockIdx.x - en.block0) * 16L + col;
  const f32x4* src = reinterpret_cast<const f32x4*>(en.slabs);
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
    f32x4 s = reinterpret_cast<f32x4*>(en.dW)[i];
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][col];
    reinterpret_cast<f32x4*>(en.dW)[i] = s;
  }
}

}  // namespace

// rows per workgroup (multiple of 32) for a weight-gradient launch: as many chunks as it takes
// to fill the chip, but never so many that the slab traffic outweighs the operand traffic.
// `target_wgs` = resident workgroup slots of the kernel (1x1: 64 KB LDS -> 2 per CU = 512; 3x3: 3 per CU =
// 768), so the grid is one full round rather than one and a half (1x1 at 768: 79 -> 76 us, 44 -> 39 us).
int wgrad_rows_per_wg(int M, int other_grid, int min_rows, int target_wgs) {
  long want_chunks = (target_wgs + other_grid - 1) / other_grid;
  long rows = (M + want_chunks - 1) / want_chunks;
  if (rows < min_rows) rows = min_rows;
  rows = (rows + 31) / 32 * 32;
  return (int)rows;
}

// 1x1 tiling: SUB 32-column sub-tiles per workgroup tile and the number of channel tiles.
//  * the widest tile (fewest re-reads of dO, least staging per MFMA) whose pixel chunks stay >= `min_steps`
//    steps of 32 rows with one round of resident workgroups; the small feature maps fall through to SUB = 1,
//    where the channel axis supplies the workgroups and the slab volume S x N x C x 4 B shrinks with S;
//  * tiles are balanced: ctiles = ceil(nsub / SUB), then SUB = ceil(nsub / ctiles).
void wgrad1x1_tiling(int M, int C, int N, int force_sub, int& sub, int& ctiles) {
  static const int min_steps = [] { const char* e = ddl_knob("DDL_WGRAD_MIN_STEPS"); return e ? atoi(e) : 2; }();
  const int nsub = (C + 31) / 32, ntiles = (N + 127) / 128;
  auto balanced = [&](int want, int& sb, int& ct) {
    ct = (nsub + want - 1) / want;
    sb = (nsub + ct - 1) / ct;
    if (sb == 2) { sb = 1; ct = nsub; }     // no SUB = 2 instantiation: hipcc's register allocation of that one form
  };                                         // needs > 256 VGPRs (spills); 32-column tiles take its place
  if (force_sub >= 1 && force_sub <= 4) {
    balanced(force_sub, sub, ctiles);
    return;
  }
  for (int want = 4; want >= 1; --want) {
    balanced(want, sub, ctiles);
    const long chunks = (512 + (long)ctiles * ntiles - 1) / ((long)ctiles * ntiles);
    if (want == 1 || (long)M >= chunks * 32L * min_steps) return;
  }
}

void launch_wgrad(const WgradParams& p, hipStream_t s) {
  static bool attr = false;
  const size_t lds3 = (size_t)(2 * RK * 32 + 2 * (RK + 2) * 128 + 2 * RK) * sizeof(float);
  const size_t lds3h = (size_t)(2 * 3 * RK * 32 + 2 * (RK + 2) * 160) * 2;
  auto lds1 = [](int sub) { return (size_t)2 * RK * (128 + 32 * sub) * sizeof(float); };
  auto lds1h = [](int sub) { return (size_t)2 * RK * (kBfLdd + bf_lda(sub)) * 2; };     // >= the 1.5 KB fused-k table
  if (!attr) {
#define DDL_SET_LDS(K, BYTES) DDL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&K), \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES)))
#define DDL_SET_LDS1(SUBV, PFV)                                                                                       \
    DDL_SET_LDS((wgrad1x1_kernel<SUBV, false, PFV, false>), lds1(SUBV)); DDL_SET_LDS((wgrad1x1_kernel<SUBV, true, PFV, false>), lds1(SUBV)); \
    DDL_SET_LDS((wgrad1x1_kernel<SUBV, false, PFV, true>), lds1h(SUBV)); DDL_SET_LDS((wgrad1x1_kernel<SUBV, true, PFV, true>), lds1h(SUBV))
    DDL_SET_LDS1(1, 2); DDL_SET_LDS1(1, 1); DDL_SET_LDS1(3, 1); DDL_SET_LDS1(4, 1);
    DDL_SET_LDS((wgrad3x3_kernel<false, false>), lds3); DDL_SET_LDS((wgrad3x3_kernel<true, false>), lds3);
    DDL_SET_LDS((wgrad3x3_kernel<false, true>), lds3h); DDL_SET_LDS((wgrad3x3_kernel<true, true>), lds3h);
#undef DDL_SET_LDS1
#undef DDL_SET_LDS
    attr = true;
  }
  const bool two = p.d.X1 != nullptr, bf = p.bf16 != 0;
  if (p.taps == 1) {
    dim3 grid((unsigned)(p.ctiles * ((p.N + 127) / 128) * 8 * ((p.S + 7) / 8)));
#define DDL_W1B(SUBV, PFV, BFV, BYTES)                                                                         \
    if (two) hipLaunchKernelGGL((wgrad1x1_kernel<SUBV, true, PFV, BFV>), grid, dim3(256), BYTES, s, p);          \
    else hipLaunchKernelGGL((wgrad1x1_kernel<SUBV, false, PFV, BFV>), grid, dim3(256), BYTES, s, p)
#define DDL_W1(SUBV, PFV)                                                  \
    if (bf) { DDL_W1B(SUBV, PFV, true, lds1h(SUBV)); } else { DDL_W1B(SUBV, PFV, false, lds1(SUBV)); }
    static const bool pf2 = [] { const char* e = ddl_knob("DDL_WGRAD_PF2"); return !(e && e[0] == '0'); }();
    switch (p.sub) {
      case 1: if (pf2) { DDL_W1(1, 2); } else { DDL_W1(1, 1); } break;
      case 3: DDL_W1(3, 1); break;
      default: DDL_W1(4, 1); break;
    }
#undef DDL_W1
#undef DDL_W1B
  } else {
    dim3 grid((unsigned)(3 * ((p.C + 127) / 128) * 8 * ((p.S + 7) / 8)));
    if (bf) {
      if (two) hipLaunchKernelGGL((wgrad3x3_kernel<true, true>), grid, dim3(256), lds3h, s, p);
      else hipLaunchKernelGGL((wgrad3x3_kernel<false, true>), grid, dim3(256), lds3h, s, p);
    } else {
      if (two) hipLaunchKernelGGL((wgrad3x3_kernel<true, false>), grid, dim3(256), lds3, s, p);
      else hipLaunchKernelGGL((wgrad3x3_kernel<false, false>), grid, dim3(256), lds3, s, p);
    }
  }
}

// grouped launch: fills block0 of every entry (workgroups per entry = tiles x 8 x ceil(S / 8), XCD-grouped)
void launch_wgrad_group(WgradGroupParams& g, hipStream_t s) {
  static bool attr = false;
  // DDL_WGRAD_GROUP_LDS1 / _LDS9: LDS bytes REQUESTED per workgroup (>= what the kernel uses).  Asking for more
  // lowers the number of co-resident weight-gradient workgroups per CU, which leaves LDS / registers for the
  // (latency-critical) data-gradient chain that runs beside them.
  static const size_t want1 = [] { const char* e = ddl_knob("DDL_WGRAD_GROUP_LDS1"); return e ? (size_t)atol(e) : 0; }();
  static const size_t want9 = [] { const char* e = ddl_knob("DDL_WGRAD_GROUP_LDS9"); return e ? (size_t)atol(e) : 0; }();
  size_t lds1 = (size_t)2 * RK * (128 + 128) * sizeof(float);
  size_t lds3 = (size_t)(2 * RK * 32 + 2 * (RK + 2) * 128 + 2 * RK) * sizeof(float);
  size_t lds1h = (size_t)2 * RK * (kBfLdd + bf_lda(4)) * 2;
  size_t lds3h = (size_t)(2 * 3 * RK * 32 + 2 * (RK + 2) * 160) * 2;
  if (want1 > lds1) lds1 = want1;
  if (want9 > lds3) lds3 = want9;
  if (want1 > lds1h) lds1h = want1;
  if (want9 > lds3h) lds3h = want9;
  if (!attr) {
#define DDL_SET_LDS(K, BYTES) DDL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&K), \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES)))
    DDL_SET_LDS((wgrad_group_kernel<1, false>), lds1); DDL_SET_LDS((wgrad_group_kernel<9, false>), lds3);
    DDL_SET_LDS((wgrad_group_kernel<1, true>), lds1h); DDL_SET_LDS((wgrad_group_kernel<9, true>), lds3h);
#undef DDL_SET_LDS
    attr = true;
  }
  if (g.n <= 0) return;
  long blocks = 0;
  const int s8 = 8 * ((g.S + 7) / 8);
  for (int i = 0; i < g.n; ++i) {
    g.e[i].block0 = (int)blocks;
    const int tiles = g.taps == 1 ? g.e[i].ctiles * ((g.N + 127) / 128) : 3 * ((g.e[i].C + 127) / 128);
    blocks += (long)tiles * s8;
  }
  const dim3 grid((unsigned)blocks);
  if (g.bf16) {
    if (g.taps == 1) hipLaunchKernelGGL((wgrad_group_kernel<1, true>), grid, dim3(256), lds1h, s, g);
    else hipLaunchKernelGGL((wgrad_group_kernel<9, true>), grid, dim3(256), lds3h, s, g);
  } else {
    if (g.taps == 1) hipLaunchKernelGGL((wgrad_group_kernel<1, false>), grid, dim3(256), lds1, s, g);
    else hipLaunchKernelGGL((wgrad_group_kernel<9, false>), grid, dim3(256), lds3, s, g);
  }
}

// one launch for up to kMaxSlabEntries queued (slabs, S, total4, dW) reductions; block0 is filled here
void launch_slab_reduce_multi(SlabReduceParams& p, hipStream_t s) {
  if (p.n <= 0) return;
  long blocks = 0;
  for (int i = 0; i < p.n; ++i) {
    p.e[i].block0 = (int)blocks;
    blocks += (p.e[i].total4 + 15) / 16;
  }
  hipLaunchKernelGGL(slab_reduce_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
}

}  // namespace ddl
