// Device-side helpers shared by the gfx950 kernels: vector types, 16-byte loads / stores, and
// buffer-descriptor (SRD) addressing.
//
// Buffer addressing: 32-bit byte offsets against a wave-uniform descriptor with hardware bounds
// checking - an out-of-range offset loads 0 and drops stores.  Every "is this row / tap / column
// valid" decision becomes a select between the real offset and OOB, so operand loads and epilogues
// are straight-line code (the branchy flat-address form spent ~10x more instructions on addressing
// than on MFMA issue and serialised its loads behind vmcnt(0) waits).  Descriptors must be built
// from kernel arguments / blockIdx only, so that hipcc keeps them in SGPRs.
#pragma once
#include <hip/hip_runtime.h>

namespace ddl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
using rsrc_t = __amdgpu_buffer_rsrc_t;

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

constexpr unsigned OOB = 0x80000000u;     // beyond any descriptor: extents are clamped to 2^31 - 1 (make_rsrc_rows)

__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
// Descriptor over the rows >= row0 of a row-major fp32 view of `bytes` bytes.  A 32-bit offset reaches 2 GiB, so
// tensors beyond that (batches of more than ~650 images at 56x56x256) are addressed by REBASING the descriptor per
// workgroup: row0 comes from blockIdx (scalar arithmetic, the descriptor stays in SGPRs), offsets are relative to
// it, and the window one workgroup touches (its row tile plus a 3x3 halo) is a few megabytes.  The extent is
// clamped below the OOB sentinel, so "offset = OOB" still selects nothing.
// The 238 forward / data-gradient launches of a step sit on one dependent chain, and ~60 extra scalar instructions
// at the top of each of them cost 0.06 ms per step (same-box A/B) - a run-time flag did not remove that - so the
// conv kernels are instantiated twice (template bool RB) and the host picks: `needs_rebase` on the views' extents.
// The weight-gradient and stem kernels (a few dozen launches beside the chain) always rebase.
__host__ __device__ __forceinline__ bool needs_rebase(long a, long b = 0, long c = 0, long d = 0) {
  return (a | b | c | d) > 0x7fffffffL;          // extents are non-negative
}
__device__ __forceinline__ rsrc_t make_rsrc_rows(const float* p, long bytes, long row0, int ld, bool rebase = true) {
  if (!rebase) return make_rsrc(p, (unsigned)bytes);      // uniform branch; row0 is 0 by the caller's contract
  const long off = row0 * (long)ld * 4;
  long rem = bytes - off;
  rem = rem < 0 ? 0 : (rem > 0x7fffffffL ? 0x7fffffffL : rem);
  return make_rsrc(reinterpret_cast<const char*>(p) + off, (unsigned)rem);
}
__device__ __forceinline__ f32x4 bld4(rsrc_t r, unsigned off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}
__device__ __forceinline__ void bst4(rsrc_t r, unsigned off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)off, 0, 0);
}

// BN-backward coefficients of channel c from the fused (sum g, sum g*h) accumulators (FusedK, conv_params.h):
//   dgamma = rstd * (sum g*h - mean * sum g), dbeta = sum g,
//   dh = k0*g + k1*h + k2:  k0 = sc, k1 = -sc*rstd*dgamma/M, k2 = -sc*dbeta/M + sc*rstd*mean*dgamma/M
template <typename FK>
__device__ __forceinline__ void fused_k(const FK& f, int c, int replicas, double inv_scale, float& k0, float& k1,
                                        float& k2, float& dgamma, float& dbeta) {
  long long s1 = 0, s2 = 0;
  for (int r = 0; r < replicas; ++r) {
    s1 += f.acc[(long)(2 * r) * f.acc_ld + c];
    s2 += f.acc[(long)(2 * r + 1) * f.acc_ld + c];
  }
  const double sg = (double)s1 * inv_scale, sgx = (double)s2 * inv_scale;
  const double mean = f.mean[c], rstd = f.rstd[c], sc = f.scale[c];
  const double dgam = rstd * (sgx - mean * sg);
  const double a = sg * f.inv_count, b = dgam * f.inv_count;
  k0 = (float)sc;
  k1 = (float)(-sc * b * rstd);
  k2 = (float)(-sc * a + sc * b * rstd * mean);
  dgamma = (float)dgam;
  dbeta = (float)sg;
}

}  // namespace ddl

// Phase probes (-DDDL_PHASE_PROBE builds only, bench/phase_probe.py): thread 0 of the first and the last workgroup of
// column tile 0 stamps the constant 100 MHz clock at phase boundaries of a kernel.
#ifdef DDL_PHASE_PROBE
#define DDL_PROBE(p, i)                                                                                        \
  do {                                                                                                         \
    if ((p).dbg != nullptr && threadIdx.x == 0 && blockIdx.y == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) \
      (p).dbg[(blockIdx.x == 0 ? 0 : 16) + (i)] = (long long)wall_clock64();                                    \
  } while (0)
#else
#define DDL_PROBE(p, i) do {} while (0)
#endif
