// How many workgroups of a given dynamic-LDS size does one MI355X CU hold at once?
// (the 3x3 weight-gradient kernel asked for 43 KB and got 2 per CU, not 3: profiles/r3/wgrad_probe/README.md)
//   hipcc --offload-arch=gfx950 -O3 lds_occupancy.hip -o lds_occupancy && ./lds_occupancy
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>
__global__ void __launch_bounds__(256) k(long long* out, int spin_us) {
  extern __shared__ float smem[];
  const long long t0 = wall_clock64();
  if (threadIdx.x == 0) smem[0] = 1.f;
  while (wall_clock64() - t0 < 100LL * spin_us) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    out[3 * blockIdx.x] = t0; out[3 * blockIdx.x + 1] = wall_clock64();
    out[3 * blockIdx.x + 2] = ((long long)(xcc & 0xf) << 32) | ((hw >> 8) & 0xff);
  }
}
int main() {
  const int grid = 2048;
  long long* d; hipMalloc(&d, grid * 24);
  std::vector<long long> h(grid * 3);
  for (int kb : {8, 16, 20, 26, 27, 32, 39, 40, 41, 43, 48, 53, 54, 64, 79, 80, 81, 96, 128, 160}) {
    const int bytes = kb * 1024;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) { printf("%3d KB: attribute refused\n", kb); continue; }
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), bytes, 0, d, 20);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%3d KB: launch failed\n", kb); (void)hipGetLastError(); continue; }
    hipMemcpy(h.data(), d, grid * 24, hipMemcpyDeviceToHost);
    std::map<long long, std::vector<std::pair<long long, int>>> ev;
    for (int i = 0; i < grid; ++i) { ev[h[3 * i + 2]].push_back({h[3 * i], 1}); ev[h[3 * i + 2]].push_back({h[3 * i + 1], -1}); }
    int best = 0;
    for (auto& kv : ev) { std::sort(kv.second.begin(), kv.second.end()); int c = 0; for (auto& e : kv.second) { c += e.second; best = std::max(best, c); } }
    printf("%3d KB dynamic LDS: at most %d workgroups of 256 threads at once on a CU (%zu CUs seen)\n", kb, best, ev.size());
  }
  return 0;
}
