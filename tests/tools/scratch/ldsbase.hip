#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
__global__ __launch_bounds__(256, 2) void k(unsigned* out) {
  extern __shared__ char smem[];
  smem[threadIdx.x] = 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned lds = __builtin_amdgcn_s_getreg((31 << 11) | 6);
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    out[3 * blockIdx.x] = lds; out[3 * blockIdx.x + 1] = hw; out[3 * blockIdx.x + 2] = xcc;
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(100);
  }
  __syncthreads();
}
int main() {
  unsigned* d; hipMalloc(&d, 512 * 12);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 56320);
  hipLaunchKernelGGL(k, dim3(512), dim3(256), 56320, 0, d);
  unsigned h[1536]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  std::map<unsigned, int> bases; std::map<unsigned, int> cus;
  for (int i = 0; i < 512; ++i) { bases[h[3 * i]]++; cus[((h[3 * i + 2] & 15) << 8) | ((h[3 * i + 1] >> 8) & 0xFF)]++; }
  for (auto& kv : bases) printf("lds_alloc %08x x%d\n", kv.first, kv.second);
  printf("distinct cu keys %zu\n", cus.size());
  int c2 = 0; for (auto& kv : cus) if (kv.second == 2) ++c2;
  printf("cus with exactly 2 wgs: %d\n", c2);
  for (int i = 0; i < 20; ++i) printf("blk %d lds %08x hw %08x xcc %x\n", i, h[3 * i], h[3 * i + 1], h[3 * i + 2]);
  return 0;
}
