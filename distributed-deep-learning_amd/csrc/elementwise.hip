// Bandwidth-bound helper kernels for gfx950 (wave64): input normalisation (K14), fused
// softmax+cross-entropy(+argmax) forward/backward (K10/K11), flat fused Adam (+zero_grad) (K12/K13).
// Reference ops: Normalize (single.py:38-42), F.cross_entropy + torch.argmax (single.py:139-141),
// optim.Adam defaults (single.py:108,305).
#include "ddl_kernels.h"

namespace ddl {

// ---------------------------------------------------------------------------------------
// K14: uint8 NCHW -> fp32/255 NHWC|NCHW.  One thread per pixel: per-plane byte reads are
// coalesced across the wave, the NHWC write is 12 contiguous bytes per lane.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) normalize_u8_kernel(const uint8_t* __restrict__ in,
                                                           float* __restrict__ out, int C, int HW,
                                                           long total_px, bool nhwc) {
  const float k = 1.0f / 255.0f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total_px; i += (long)gridDim.x * 256L) {
    long b = i / HW;
    long px = i - b * HW;
    const uint8_t* src = in + b * C * HW + px;
    if (nhwc) {
      float* dst = out + i * C;
      for (int c = 0; c < C; ++c) dst[c] = (float)src[(long)c * HW] * k;
    } else {
      float* dst = out + b * C * HW + px;
      for (int c = 0; c < C; ++c) dst[(long)c * HW] = (float)src[(long)c * HW] * k;
    }
  }
}

void launch_normalize_u8(const uint8_t* in, float* out, int B, int C, int H, int W, bool nhwc,
                         hipStream_t s) {
  long total = (long)B * H * W;
  int grid = (int)((total + 255) / 256);
  if (grid > 256 * 8) grid = 256 * 8;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(normalize_u8_kernel, dim3(grid), dim3(256), 0, s, in, out, C, H * W, total, nhwc);
}

// ---------------------------------------------------------------------------------------
// K10/K11: one thread per row (C is tiny: 5 classes).  loss = mean_i (lse_i - z_i[t_i]);
// dlogits = (softmax - onehot) / B; preds = argmax (first max, like torch.argmax).
// ONE workgroup of 1024 threads walks the rows (a per-GPU batch is at most a few thousand rows of 5
// floats - the kernel is pure launch latency) and the loss is combined in a FIXED order: per-thread
// running sum in row order, wave shuffle tree, the 16 wave sums added left to right.  No float atomics,
// no memset node: the result is bit-identical from run to run at any batch size.
// A target outside [0, C) (torch raises a device assert there) poisons the loss with NaN and gets a
// zero gradient row instead of reading out of bounds.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) ce_fwd_bwd_kernel(const float* __restrict__ logits,
                                                          const int64_t* __restrict__ targets,
                                                          float* __restrict__ loss,
                                                          float* __restrict__ dlogits,
                                                          int64_t* __restrict__ preds, int B, int C) {
  __shared__ float wave_sums[16];
  float local = 0.f;
  const float invB = 1.0f / (float)B;
  for (int r = threadIdx.x; r < B; r += 1024) {
    const float* z = logits + (long)r * C;
    float mx = z[0];
    int arg = 0;
    for (int c = 1; c < C; ++c) {
      float v = z[c];
      if (v > mx) { mx = v; arg = c; }
    }
    float sum = 0.f;
    for (int c = 0; c < C; ++c) sum += expf(z[c] - mx);      // full-precision libm forms: this value is logged as `loss`
    float lse = mx + logf(sum);
    const int64_t t64 = targets[r];
    const bool ok = t64 >= 0 && t64 < C;
    const int t = ok ? (int)t64 : 0;
    local += ok ? lse - z[t] : __builtin_nanf("");
    float inv = 1.0f / sum;
    float* dz = dlogits + (long)r * C;
    for (int c = 0; c < C; ++c) {
      float p = expf(z[c] - mx) * inv;
      dz[c] = ok ? (p - (c == t ? 1.f : 0.f)) * invB : 0.f;
    }
    preds[r] = arg;
  }
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) wave_sums[wid] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int w = 0; w < 16; ++w) tot += wave_sums[w];
    loss[0] = tot * invB;
  }
}

// ---------------------------------------------------------------------------------------------
// copy_pair: the step's two input copies (image batch + labels into the captured graph's static buffers) as ONE
// launch - every eager operation between two graph launches costs a queue round trip at the step boundary.
// a: n16 16-byte chunks (grid-stride), b: nb bytes (first workgroup).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) copy_pair_kernel(const uint4* __restrict__ a_src, uint4* __restrict__ a_dst,
                                                        long n16, const unsigned char* __restrict__ b_src,
                                                        unsigned char* __restrict__ b_dst, long nb) {
  const long stride = (long)gridDim.x * 256;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n16; i += stride) a_dst[i] = a_src[i];
  if (blockIdx.x == 0)
    for (long i = threadIdx.x; i < nb; i += 256) b_dst[i] = b_src[i];
}

void launch_copy_pair(const void* a_src, void* a_dst, long a_bytes, const void* b_src, void* b_dst, long b_bytes,
                      hipStream_t s) {
  const long n16 = a_bytes / 16;
  long grid = (n16 + 255) / 256;
  if (grid > 2048) grid = 2048;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(copy_pair_kernel, dim3((unsigned)grid), dim3(256), 0, s, reinterpret_cast<const uint4*>(a_src),
                     reinterpret_cast<uint4*>(a_dst), n16, reinterpret_cast<const unsigned char*>(b_src),
                     reinterpret_cast<unsigned char*>(b_dst), b_bytes);
}

void launch_ce_fwd_bwd(const float* logits, const int64_t* targets, float* loss, float* dlogits,
                       int64_t* preds, int B, int C, hipStream_t s) {
  hipLaunchKernelGGL(ce_fwd_bwd_kernel, dim3(1), dim3(1024), 0, s, logits, targets, loss, dlogits,
                     preds, B, C);
}

// ---------------------------------------------------------------------------------------
// K12/K13: Adam over a flat arena.  16 B per lane per array (float4), 4 reads + 3 writes per
// element (+1 write when zeroing grads) - pure HBM streaming.  The step counter is a device
// float so that the launch sequence is identical every step (hipGraph replay).
// ---------------------------------------------------------------------------------------
// Single-thread prologue: step += 1 and the bias corrections in double precision
// (aux[0] = 1 - b1^t, aux[1] = 1/sqrt(1 - b2^t)), so the streaming kernel does no pow().
__global__ void adam_inc_step_kernel(float* step_t, float* aux, double b1, double b2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float t = step_t[0] + 1.0f;
    step_t[0] = t;
    aux[0] = (float)(1.0 - pow(b1, (double)t));
    aux[1] = (float)(1.0 / sqrt(1.0 - pow(b2, (double)t)));
  }
}

void launch_adam_inc_step(float* step_t, float* aux, double b1, double b2, hipStream_t s) {
  hipLaunchKernelGGL(adam_inc_step_kernel, dim3(1), dim3(64), 0, s, step_t, aux, b1, b2);
}

// omb1/omb2 = (1 - beta) rounded from double on the host (1.f - 0.999f is off by 5e-5 relative)
__device__ __forceinline__ void adam_one(float& p, float& g, float& m, float& v, float lr_c, float b1,
                                         float b2, float omb1, float omb2, float eps, float wd,
                                         float inv_sqrt_bc2) {
  float gg = g;
  if (wd != 0.f) gg += wd * p;
  m = b1 * m + omb1 * gg;
  v = b2 * v + omb2 * gg * gg;
  float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
  p -= lr_c * (m / denom);
}

__global__ void __launch_bounds__(256) adam_flat_kernel(float* __restrict__ p, float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        const float* __restrict__ aux, long n4,
                                                        float lr, float b1, float b2, float omb1,
                                                        float omb2, float eps, float wd,
                                                        bool zero_grad) {
  const float lr_c = lr / aux[0];
  const float inv_sqrt_bc2 = aux[1];
  float4* p4 = reinterpret_cast<float4*>(p);
  float4* g4 = reinterpret_cast<float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256L) {
    float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
    adam_one(pp.x, gg.x, mm.x, vv.x, lr_c, b1, b2, omb1, omb2, eps, wd, inv_sqrt_bc2);
    adam_one(pp.y, gg.y, mm.y, vv.y, lr_c, b1, b2, omb1, omb2, eps, wd, inv_sqrt_bc2);
    adam_one(pp.z, gg.z, mm.z, vv.z, lr_c, b1, b2, omb1, omb2, eps, wd, inv_sqrt_bc2);
    adam_one(pp.w, gg.w, mm.w, vv.w, lr_c, b1, b2, omb1, omb2, eps, wd, inv_sqrt_bc2);
    p4[i] = pp;
    m4[i] = mm;
    v4[i] = vv;
    if (zero_grad) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

void launch_adam_flat(float* p, float* g, float* m, float* v, const float* aux, int64_t start,
                      int64_t end, double lr, double b1, double b2, double eps, double wd, bool zero_grad,
                      hipStream_t s) {
  // arena ranges are multiples of 64 elements (FlatArena alignment) -> float4-safe
  long n4 = (long)(end - start) / 4;
  if (n4 <= 0) return;
  long grid = (n4 + 255) / 256;
  if (grid > 256 * 8) grid = 256 * 8;
  hipLaunchKernelGGL(adam_flat_kernel, dim3((int)grid), dim3(256), 0, s, p + start, g + start, m + start,
                     v + start, aux, n4, (float)lr, (float)b1, (float)b2, (float)(1.0 - b1), (float)(1.0 - b2),
                     (float)eps, (float)wd, zero_grad);
}

// ---------------------------------------------------------------------------------------------
// Fence-test helpers (tests/test_gpu_fences.py; reducer.cpp / p2p.cpp / parallel/dp.py `test_hook`).  On a one-GPU
// box every RCCL collective of a world of one is an identity, so a missing stream fence cannot change a bit.  With
// the hook enabled the engines enqueue `spin` + `scale by 0.5` on their communication stream between their two
// fences: the scale makes the communication stream's work visible in the data, the spin makes a missing fence lose
// the race deterministically.  The spin is bounded by a wall-clock deadline AND an iteration cap.
// ---------------------------------------------------------------------------------------------
__global__ void spin_kernel(long long ticks, int* sink) {
  const long long t0 = wall_clock64();
  long long it = 0;
  while (wall_clock64() - t0 < ticks && it < (1LL << 33)) ++it;
  if (sink != nullptr && it == -1) sink[0] = 1;      // never true: keeps the loop alive without a store
}

void launch_spin(double micros, hipStream_t s) {
  // wall_clock64 ticks at 100 MHz on gfx9 (constant, independent of the shader clock)
  const long long ticks = (long long)(micros * 100.0);
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, s, ticks, (int*)nullptr);
}

__global__ void __launch_bounds__(256) scale_kernel(float* __restrict__ p, long n, float a) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) p[i] *= a;
}

void launch_scale(float* p, long n, float a, hipStream_t s) {
  if (n <= 0) return;
  long grid = (n + 255) / 256;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(scale_kernel, dim3((unsigned)grid), dim3(256), 0, s, p, n, a);
}

}  // namespace ddl
