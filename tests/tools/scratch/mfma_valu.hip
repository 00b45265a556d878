// Does a wave running MFMAs slow a VALU-only wave on the same SIMD?  (dependent chain vs independent accumulators)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, int OP, int PRIO>
__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* t, int iters) {
  const int wave = threadIdx.x >> 6;   // 8 waves: two per SIMD. waves 0-3 = MFMA, 4-7 = VALU
  if (wave < 4) {
    f32x16 acc[NACC > 0 ? NACC : 1];
    for (int a = 0; a < (NACC > 0 ? NACC : 1); ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float av = threadIdx.x * 0.001f, bv = 1.0f;
    if (NACC > 0) {
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[u % NACC], 0, 0, 0);
      }
    } else {
      for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(16);
    }
    float s = 0.f; for (int a = 0; a < (NACC > 0 ? NACC : 1); ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    unsigned long long t0 = wall_clock64();
    float x = threadIdx.x * 0.5f, y = 1.0001f;
    unsigned u = threadIdx.x;
    for (int i = 0; i < 2000; ++i) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        if (OP == 0) x = x * y + 0.5f;                       // dependent fma chain
        else { u = u * 1664525u + 1013904223u; }             // integer mul/add (quarter rate mul)
      }
    }
    unsigned long long t1 = wall_clock64();
    out[blockIdx.x * 512 + threadIdx.x] = x + u;
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * 4 + wave - 4] = t1 - t0;
  }
}
template <int NACC, int OP, int PRIO = 0> void run(const char* name, int iters) {
  float* out; unsigned long long* t; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&t, 256 * 4 * 8);
  hipLaunchKernelGGL((k<NACC, OP, PRIO>), dim3(256), dim3(512), 0, 0, out, t, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL((k<NACC, OP, PRIO>), dim3(256), dim3(512), 0, 0, out, t, iters); hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[1024]; hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 1024; ++i) s += h[i];
  printf("%-34s kernel %.1f us; VALU wave: 32000 ops in %.2f us (%.1f ns/op)\n", name, ms * 1e3, s / 1024 * 0.01, s / 1024 * 10 / 32000);
  hipFree(out); hipFree(t);
}
int main() {
  const int iters = 600;   // 9600 MFMAs ~ 300 us
  run<0, 0>("partner sleeping, fma chain", iters);
  run<1, 0>("partner 1 acc (dependent), fma", iters);
  run<2, 0>("partner 2 acc, fma", iters);
  run<4, 0>("partner 4 acc, fma", iters);
  run<1, 0, 1>("partner 1 acc, fma, VALU wave s_setprio 3", iters);
  run<4, 0, 1>("partner 4 acc, fma, VALU wave s_setprio 3", iters);
  run<0, 1>("partner sleeping, int mul", iters);
  run<1, 1>("partner 1 acc (dependent), imul", iters);
  run<2, 1>("partner 2 acc, imul", iters);
  run<4, 1>("partner 4 acc, imul", iters);
  return 0;
}
