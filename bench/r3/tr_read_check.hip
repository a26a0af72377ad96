// Semantics check of ds_read_b64_tr_b16 as the weight-gradient kernels use it (conv_wgrad.hip, bf16 path):
// both MFMA operands of D[i][j] += sum_k X[k][i] * Y[k][j] are read from PIXEL-major (k-major) LDS tiles.
//   hipcc --offload-arch=gfx950 -O3 tr_read_check.hip -o tr_read_check && ./tr_read_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
constexpr int LD = 160;   // elements: 320 B rows
__global__ void k(const float* x, const float* y, float* out) {
  __shared__ __attribute__((aligned(16))) __bf16 sx[16 * LD], sy[16 * LD];
  for (int i = threadIdx.x; i < 16 * 32; i += 64) {
    sx[(i >> 5) * LD + (i & 31)] = (__bf16)x[i];
    sy[(i >> 5) * LD + (i & 31)] = (__bf16)y[i];
  }
  __syncthreads();
  const int l = threadIdx.x;
  const int q = (l & 15) >> 2, pp = l & 3, c0 = 16 * ((l >> 4) & 1), r0 = 8 * (l >> 5);
  const int off = (r0 + q) * LD + c0 + 4 * pp;
  bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sx + off));
  bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sx + off + 4 * LD));
  bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sy + off));
  bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sy + off + 4 * LD));
  bf16x8 a = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
  bf16x8 b = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  const int li = l & 31, lh = l >> 5;
  for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = acc[r];   // out[i][j = li]
}
int main() {
  std::vector<float> x(512), y(512), o(1024);
  srand(1);
  for (auto& v : x) v = (float)(rand() % 17 - 8);
  for (auto& v : y) v = (float)(rand() % 13 - 6);
  float *dx, *dy, *dout;
  hipMalloc(&dx, 2048); hipMalloc(&dy, 2048); hipMalloc(&dout, 4096);
  hipMemcpy(dx, x.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dy, y.data(), 2048, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dy, dout);
  hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    float w = 0; for (int kk = 0; kk < 16; ++kk) w += x[kk * 32 + i] * y[kk * 32 + j];
    if (w != o[i * 32 + j]) { if (bad < 5) printf("mismatch D[%d][%d] got %g want %g\n", i, j, o[i * 32 + j], w); ++bad; }
  }
  printf("tr_read_check: %d mismatches of 1024 -> %s\n", bad, bad ? "FAIL" : "OK");
  return bad != 0;
}
