// Forward 3x3 convolution (stride 1, pad 1) as an implicit GEMM whose input is staged ONCE per workgroup.
//
// conv_gemm.hip treats the nine taps as nine K slabs and re-stages the (shifted) input rows for each
// of them: every input element is loaded from L2, pushed through the BN+ReLU prologue and written to
// LDS nine times.  With N = 32 output channels that staging work, not the MFMA loop, bounds the kernel
// (rocprofv3: 10 VALU instructions per MFMA, 44 % MFMA utilisation on the 56x56 maps).
//
// A row tile is BM consecutive pixels of the flattened NHWC tensor, so the rows any tap can touch are
// the contiguous range [m0 - W - 1, m0 + BM + W + 1).  This kernel stages that range once per
// 32-channel chunk (prologue applied once per element), and every tap reads its operand rows from
// LDS at a constant row offset (dy*W + dx).  A (pixel, tap) pair that leaves the image is zeroed when
// the MFMA operand is read (one v_cndmask per operand register, from the lane's 9-bit tap mask).
// Weights are staged per tap through a double buffer (a whole filter row per barrier was tried: the
// extra LDS costs a workgroup per CU and more than the saved barriers).  Output + statistics: the shared
// epilogue (conv_common.h), so the result layout is identical to conv_gemm's forward kind.
//
// Measured (MI355X, fp32): 56x56x128 -> 32 at batch 30: 104 -> 78 us (89 TFLOP/s); batch 128: 406 -> 334 us,
// 28x28: 108 -> 93 us.  The 3x3 data gradient (N = 128, K = 288) was tried on the same scheme and is not
// faster than its conv_gemm tiles - its staging was never the bottleneck - so it stays there.
#include "conv_common.h"
#include "conv_params.h"
#include "ddl_kernels.h"
#include "device_utils.h"

namespace ddl {

namespace {

constexpr int HALO_MAXW = 56;            // widest feature map of DenseNet-121 after the stem
constexpr int kMaxSmallC = 128;          // small-M form: channels of the staged range (the dense layers' bottleneck width)

template <int BM, int BN, int WM, int WN, bool RB>
__global__ void __launch_bounds__(64 * WM * WN) conv3x3_halo_kernel(const ConvGemmParams p) {
  constexpr int NT = 64 * WM * WN;
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
  constexpr int LDA = 36;                // row stride of the K-contiguous LDS images: 32 channels + 4 pad
  constexpr int RMAX = BM + 2 * HALO_MAXW + 2;
  constexpr int PA = (RMAX * 8 + NT - 1) / NT;        // float4 per thread for the widest range
  constexpr int PB = (BN * 8) / NT;                   // float4 per thread for one tap of weights
  constexpr int B_ELEMS = BN * LDA;
  static_assert((BN * 8) % NT == 0 && NT % 8 == 0, "a weight tap must divide over the workgroup");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const bool affine = p.a.s0 != nullptr || p.bn.acc != nullptr;
  const int soff = affine ? 2 * p.C : 0;
  float* scs = smem;                     // BN scale / shift of the C input channels
  float* shs = smem + p.C;
  const int halo = p.W + 1;
  const int R = BM + 2 * halo;           // staged pixel rows
  float* As = smem + soff;               // [R][LDA]
  float* Bs = As + R * LDA;              // [2][B_ELEMS]

  const int tid = threadIdx.x, lane = tid & 63, wmn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wmn / WN, wn = wmn - wm * WN;
  const int li = lane & 31, lh = lane >> 5;
  int tile_x;
  {   // XCD-aware tile order (see conv_gemm.hip)
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    tile_x = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int m0 = tile_x * BM;
  const int n0 = blockIdx.y * BN;
  const int chunks = p.C >> 5;

  constexpr bool rb = RB;                                // views beyond 2 GiB (host decision, device_utils.h)
  const int mbase = (rb && m0 > halo) ? m0 - halo : 0;   // first row the staged range can touch
  const rsrc_t rA = make_rsrc_rows(p.a.X0, p.a.bytes0, mbase, p.a.ld0, rb);
  const rsrc_t rW = make_rsrc(p.Wt, p.wbytes);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // ---- staging roles ---------------------------------------------------------------------------
  const int cq = tid & 7;                // channel quad of the chunk this thread stages (NT % 8 == 0)
  unsigned a_off[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int r = (tid + i * NT) >> 3;
    const int m = m0 - halo + r;
    a_off[i] = (r < R && m >= 0 && m < p.M) ? (unsigned)(m - mbase) * (unsigned)(p.a.ld0 * 4) + (unsigned)(16 * cq) : OOB;
  }
  // ---- compute roles: 9-bit tap mask of this lane's pixel in each of its row blocks ------------------
  unsigned bits[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + (wm * TM + i) * 32 + li;
    unsigned b = 0;
    if (m < p.M) {
      const int rem = m % (p.H * p.W);
      const int y = rem / p.W, x = rem - y * p.W;
      const unsigned yb = (y > 0) | 2u | ((unsigned)(y < p.H - 1) << 2);
      const unsigned xb = (x > 0) | 2u | ((unsigned)(x < p.W - 1) << 2);
      b = ((yb & 1u) ? xb : 0u) | ((yb & 2u) ? (xb << 3) : 0u) | ((yb & 4u) ? (xb << 6) : 0u);
    }
    bits[i] = b;
  }

  // weights travel three taps ahead of their use (one tap of MFMAs is shorter than an L2 round trip):
  // b_q[t % 3] holds tap t from its load until it is written to the LDS buffer one tap before its use
  f32x4 a_v[PA], b_q[3][PB];
  auto load_a = [&](int chunk) {
    const unsigned dc = (unsigned)(chunk * 128);        // 32 channels * 4 bytes
#pragma unroll
    for (int i = 0; i < PA; ++i) a_v[i] = bld4(rA, a_off[i] == OOB ? OOB : a_off[i] + dc);
  };
  auto store_a = [&](int chunk) {
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = zero4;
    if (affine) {
      sc = ld4(scs + chunk * 32 + 4 * cq);
      sh = ld4(shs + chunk * 32 + 4 * cq);
    }
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int r = (tid + i * NT) >> 3;
      f32x4 v = a_v[i];
      if (affine) {
        v = sc * v + sh;
        if (p.a.relu) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
      }
      // rows outside the tensor hold finite garbage (the prologue of zeros); every (pixel, tap) pair that
      // could read them is masked by `bits`
      if (r < R) *reinterpret_cast<f32x4*>(As + r * LDA + 4 * cq) = v;
    }
  };
  auto load_b = [&](int chunk, int tap, f32x4 (&dst)[PB]) {     // weights [N][9][C], one tap, 32 channels
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int f = tid + i * NT;
      const int n = n0 + (f >> 3);
      const unsigned off = (unsigned)((n * p.ldb + tap * p.C + chunk * 32 + 4 * (f & 7)) * 4);
      dst[i] = bld4(rW, n < p.N ? off : OOB);
    }
  };
  auto store_b = [&](int buf, const f32x4 (&src)[PB]) {
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int f = tid + i * NT;
      *reinterpret_cast<f32x4*>(Bs + buf * B_ELEMS + (f >> 3) * LDA + 4 * (f & 7)) = src[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int tap, int buf) {
    const int ty = (tap * 11) >> 5, tx = tap - 3 * ty;
    const int toff = (ty - 1) * p.W + (tx - 1);        // row offset of this tap in the flattened tensor
    const float* Ab = As + (halo + toff + wm * TM * 32 + li) * LDA + 4 * lh;
    const float* Bb = Bs + buf * B_ELEMS + (wn * TN * 32 + li) * LDA + 4 * lh;
    bool on[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) on[i] = (bits[i] >> tap) & 1u;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        af[i] = ld4(Ab + i * 32 * LDA + 8 * s);
        af[i] = on[i] ? af[i] : zero4;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = ld4(Bb + j * 32 * LDA + 8 * s);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < TM; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
      }
    }
  };

  // ---- main loop: chunk = 32 input channels; per chunk the range is staged once, then nine taps.
  // Step s = chunk * 9 + tap reads weight buffer s & 1 = (chunk + tap) & 1.
  load_a(0);
  load_b(0, 0, b_q[0]);
  load_b(0, 1, b_q[1]);
  load_b(0, 2, b_q[2]);
  conv_fill_tables<NT, 0>(p, affine, scs, shs, scs, tid);        // scale / shift tables (fused statistics)
  __syncthreads();
  store_a(0);
  store_b(0, b_q[0]);
  __syncthreads();
  for (int chunk = 0; chunk < chunks; ++chunk) {
    const bool more = chunk + 1 < chunks;
    if (more) load_a(chunk + 1);                                  // in flight under the nine taps
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      compute(tap, (chunk + tap) & 1);
      if (tap < 8 || more) store_b((chunk + tap + 1) & 1, b_q[(tap + 1) % 3]);
      if (tap + 3 < 9) load_b(chunk, tap + 3, b_q[tap % 3]);
      else if (more) load_b(chunk + 1, tap + 3 - 9, b_q[tap % 3]);
      __syncthreads();
    }
    if (more) {
      store_a(chunk + 1);
      __syncthreads();
    }
  }
  conv_epilogue<BM, BN, WM, WN, NT, 0, RB>(p, acc, smem, m0, n0, tile_x, tid, 0, wmn);
}

// ---------------------------------------------------------------------------------------------
// Small-M form (14x14 / 7x7 maps, and 28x28 at small batch): fewer row tiles than CUs, so the kernel is
// bound by its chain of dependent phases, not by throughput.  conv_gemm's K-split tiles run
// ceil(36 / KS) slab iterations of load -> LDS -> barrier -> MFMA.  Here the whole input range (all C
// channels, halo rows included) is staged with ONE round of loads and one barrier; after that every
// wave works alone: wave (row block, k group) takes every KS-th (chunk, tap) pair, pulls the 32x32 weight
// block of the pair straight from L2 into registers (it is an MFMA operand as loaded: lane (n, h) holds
// channels 16h..16h+15 of output channel n) and reads its input rows from LDS at the tap's row offset.
// No barrier until the K-split combine.  The k-steps of a pair are ordered c = 16h + s for both operands.
// ---------------------------------------------------------------------------------------------
template <int BM, int KS, bool RB>
__global__ void __launch_bounds__(64 * (BM / 32) * KS) conv3x3_small_kernel(const ConvGemmParams p) {
  constexpr int WM = BM / 32, NT = 64 * WM * KS;
  constexpr int MAXW = 28;
  constexpr int RMAX = BM + 2 * MAXW + 2;
  constexpr int MAXQ = kMaxSmallC / 4;                 // float4 per staged row
  constexpr int PA = (RMAX * MAXQ + NT - 1) / NT;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const bool affine = p.a.s0 != nullptr || p.bn.acc != nullptr;
  const int lda = p.C + 4;                             // staged row stride (floats)
  const int halo = p.W + 1;
  const int R = BM + 2 * halo;
  float* scs = smem;
  float* shs = smem + p.C;
  float* As = smem + (affine ? 2 * p.C : 0);           // [R][lda]

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: wave-uniform guards branch on SCC
  const int ks = wid / WM, wm = wid - ks * WM;
  const int li = lane & 31, lh = lane >> 5;
  int tile_x;
  {   // XCD-aware tile order (see conv_gemm.hip)
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    tile_x = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int m0 = tile_x * BM;
  const int n0 = blockIdx.y * 32;
  const int chunks = p.C >> 5;
  const int npairs = chunks * 9;
  const int qrow = p.C >> 2;                           // float4 per row

  constexpr bool rb = RB;
  const int mbase = (rb && m0 > halo) ? m0 - halo : 0;   // first row the staged range can touch
  const rsrc_t rA = make_rsrc_rows(p.a.X0, p.a.bytes0, mbase, p.a.ld0, rb);
  const rsrc_t rW = make_rsrc(p.Wt, p.wbytes);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // ---- one round of loads for the whole range ---------------------------------------------------
  f32x4 a_v[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int f = tid + i * NT;
    const int r = f / qrow, cq = f - r * qrow;
    const int m = m0 - halo + r;
    const bool ok = r < R && m >= 0 && m < p.M;
    a_v[i] = bld4(rA, ok ? (unsigned)(m - mbase) * (unsigned)(p.a.ld0 * 4) + (unsigned)(16 * cq) : OOB);
  }
  // this lane's 9-bit tap mask (its pixel is row li of row block wm)
  unsigned bits = 0;
  {
    const int m = m0 + wm * 32 + li;
    if (m < p.M) {
      const int rem = m % (p.H * p.W);
      const int y = rem / p.W, x = rem - y * p.W;
      const unsigned yb = (y > 0) | 2u | ((unsigned)(y < p.H - 1) << 2);
      const unsigned xb = (x > 0) | 2u | ((unsigned)(x < p.W - 1) << 2);
      bits = ((yb & 1u) ? xb : 0u) | ((yb & 2u) ? (xb << 3) : 0u) | ((yb & 4u) ? (xb << 6) : 0u);
    }
  }
  // weight block of a pair: lane (n = li, h = lh) loads channels [16h, 16h + 16) of output channel n0 + li
  const unsigned w_row = (unsigned)((n0 + li) * p.ldb * 4 + 64 * lh);
  const bool w_ok = n0 + li < p.N;
  auto load_w = [&](int pair, f32x4 (&w)[4]) {
    const int chunk = pair / 9, tap = pair - 9 * chunk;
    const unsigned off = w_row + (unsigned)((tap * p.C + chunk * 32) * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = bld4(rW, (w_ok && pair < npairs) ? off + 16u * q : OOB);
  };
  f32x4 w0[4], w1[4];
  load_w(ks, w0);
  load_w(ks + KS, w1);

  conv_fill_tables<NT, 0>(p, affine, scs, shs, scs, tid);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int f = tid + i * NT;
    const int r = f / qrow, cq = f - r * qrow;
    f32x4 v = a_v[i];
    if (affine) {
      v = ld4(scs + 4 * cq) * v + ld4(shs + 4 * cq);
      if (p.a.relu) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
    }
    if (r < R) *reinterpret_cast<f32x4*>(As + r * lda + 4 * cq) = v;      // out-of-tensor rows: masked by `bits`
  }
  __syncthreads();

  // ---- every wave on its own: pairs ks, ks + KS, ... ---------------------------------------------
  f32x16 acc[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
  auto pair_mfma = [&](int pair, const f32x4 (&w)[4]) {
    const int chunk = pair / 9, tap = pair - 9 * chunk;
    const int ty = (tap * 11) >> 5, tx = tap - 3 * ty;
    const float* Ab = As + (halo + (ty - 1) * p.W + (tx - 1) + wm * 32 + li) * lda + chunk * 32 + 16 * lh;
    const bool on = (bits >> tap) & 1u;
    f32x4 a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a[q] = ld4(Ab + 4 * q);
      a[q] = on ? a[q] : zero4;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][e], w[q][e], acc[0][0], 0, 0, 0);
  };
  for (int pair = ks; pair < npairs; pair += 2 * KS) {       // weights two pairs ahead of their use
    pair_mfma(pair, w0);
    load_w(pair + 2 * KS, w0);
    if (pair + KS < npairs) pair_mfma(pair + KS, w1);
    load_w(pair + 3 * KS, w1);
  }
  __syncthreads();                                            // the staged range is dead: combine over ks

  if (KS > 1) {
    float* red = smem;                                        // [KS-1][WM][16][64]
    if (ks > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(((ks - 1) * WM + wm) * 16 + r) * 64 + lane] = acc[0][0][r];
    }
    __syncthreads();
    if (ks == 0) {
#pragma unroll
      for (int k2 = 1; k2 < KS; ++k2)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] += red[(((k2 - 1) * WM + wm) * 16 + r) * 64 + lane];
    }
    __syncthreads();
  }
  conv_epilogue<BM, 32, WM, 1, NT, 0, RB>(p, acc, smem, m0, n0, tile_x, tid, ks, wm);
}

template <int BM, int KS, bool RB>
void launch_small(const ConvGemmParams& p, hipStream_t s) {
  constexpr int WM = BM / 32, NT = 64 * WM * KS;
  constexpr size_t red = (size_t)(KS - 1) * WM * 16 * 64 * sizeof(float);
  constexpr size_t epi = (size_t)WM * 32 * kEpiLd * sizeof(float);
  constexpr size_t sta = (size_t)2 * WM * 8 * 32 * sizeof(float);
  constexpr size_t lds_max = ((size_t)(BM + 2 * 28 + 2) * (kMaxSmallC + 4) + 2 * kMaxSmallC) * sizeof(float);
  const bool tables = p.a.s0 != nullptr || p.bn.acc != nullptr;
  size_t lds = ((tables ? (size_t)2 * p.C : 0) + (size_t)(BM + 2 * p.W + 2) * (p.C + 4)) * sizeof(float);
  if (red > lds) lds = red;
  if (epi > lds) lds = epi;
  if (sta > lds) lds = sta;
  static bool attr_set = false;
  if (!attr_set) {
    DDL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_small_kernel<BM, KS, RB>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
    attr_set = true;
  }
  dim3 grid((unsigned)((p.M + BM - 1) / BM), (unsigned)((p.N + 31) / 32));
  hipLaunchKernelGGL((conv3x3_small_kernel<BM, KS, RB>), grid, dim3(NT), lds, s, p);
}

template <int BM, int BN, int WM, int WN, bool RB>
void launch_halo(const ConvGemmParams& p, hipStream_t s) {
  constexpr int NT = 64 * WM * WN;
  constexpr size_t b_elems = (size_t)BN * 36;
  constexpr size_t epi = (size_t)WM * WN * 32 * kEpiLd * sizeof(float);
  constexpr size_t sta = (size_t)2 * WM * 8 * BN * sizeof(float);
  constexpr size_t lds_max = ((size_t)(BM + 2 * HALO_MAXW + 2) * 36 + 2 * b_elems + 2 * kMaxAffineC) * sizeof(float);
  const bool tables = p.a.s0 != nullptr || p.bn.acc != nullptr;
  const size_t stage = ((tables ? (size_t)2 * p.C : 0) + (size_t)(BM + 2 * p.W + 2) * 36 + 2 * b_elems) * sizeof(float);
  size_t lds = stage > epi ? stage : epi;
  if (sta > lds) lds = sta;
  static bool attr_set = false;
  if (!attr_set) {
    DDL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<BM, BN, WM, WN, RB>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
    attr_set = true;
  }
  dim3 grid((unsigned)((p.M + BM - 1) / BM), (unsigned)((p.N + BN - 1) / BN));
  hipLaunchKernelGGL((conv3x3_halo_kernel<BM, BN, WM, WN, RB>), grid, dim3(NT), lds, s, p);
}

}  // namespace

// Which problems take these kernels: fp32 forward 3x3 with N <= 32.  Returns the row-tile height, 0 = not
// eligible (conv_gemm's tiles), and selects the form through `form`:
//   big   (128 rows, maps up to 56 wide): at least 384 row tiles - or `any_size` (tests, force 2000)
//   small (32 rows, maps up to 28 wide, C <= 128): fewer row tiles than that - or force 2001
int conv3x3_halo_bm(const ConvGemmParams& p, int force) {
  if (p.taps != 9 || p.kind != 0 || p.bf16 || p.W < 1 || (p.C & 31) || p.N > 32) return 0;
  const bool big_ok = p.W <= HALO_MAXW && p.C <= kMaxAffineC;
  const bool small_ok = p.W <= 28 && p.C <= kMaxSmallC;
  if (force == 2000) return big_ok ? 128 : 0;
  if (force == 2001) return small_ok ? 32 : 0;
  if (force != 0) return 0;
  if (big_ok && (p.M + 127) / 128 >= 384) return 128;
  return small_ok ? 32 : 0;
}

void launch_conv3x3_halo(const ConvGemmParams& p, int bm, hipStream_t s) {
  const bool rb = needs_rebase(p.a.bytes0, p.a.bytes1, p.obytes, p.auxbytes);   // views beyond 2 GiB: rebasing instantiation
  if (bm == 128) { if (rb) launch_halo<128, 32, 4, 1, true>(p, s); else launch_halo<128, 32, 4, 1, false>(p, s); }
  else { if (rb) launch_small<32, 8, true>(p, s); else launch_small<32, 8, false>(p, s); }
}

}  // namespace ddl
