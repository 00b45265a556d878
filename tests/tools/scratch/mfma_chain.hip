// How fast does v_mfma_f32_32x32x2_f32 issue when every instruction accumulates into the SAME tile (one dependent chain),
// against 2 / 3 / 9 interleaved chains?  One wave per SIMD (256 threads per workgroup), one workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void chain_kernel(float* out, long long* cycles, int iters) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 36; ++k) acc[k % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k % CHAINS], 0, 0, 0);
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * THREADS + threadIdx.x] = s;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int CHAINS, int THREADS> void run(float* out, long long* cyc, int iters) {
  hipLaunchKernelGGL((chain_kernel<CHAINS, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, cyc, iters);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((chain_kernel<CHAINS, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, cyc, iters);
  (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long h[256]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double mean = 0; for (int i = 0; i < 256; ++i) mean += h[i]; mean /= 256;
  const double n = 36.0 * iters;
  printf("chains %d, %d waves per SIMD: %.1f cycles per MFMA and wave (s_memtime), %.2f TFLOP/s by events\n", CHAINS, THREADS / 256, mean / n,
         256.0 * (THREADS / 64) * n * 4096 / (ms * 1e-3) / 1e12);
}

int main() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8);
  run<9, 256>(out, cyc, 2000); run<1, 256>(out, cyc, 2000); run<2, 256>(out, cyc, 2000); run<9, 256>(out, cyc, 2000);
  run<1, 512>(out, cyc, 2000); run<2, 512>(out, cyc, 2000); run<9, 512>(out, cyc, 2000);
  return 0;
}
