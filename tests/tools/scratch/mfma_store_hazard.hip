#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float a, float b) {
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  out[threadIdx.x] = acc[0];
  out[threadIdx.x + 64] = acc[15];
}
