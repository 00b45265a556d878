// Which lane does a wave-wide DPP shift read from on gfx950?  (photometric_rows.hip relies on it: lane i of wave_shr:1 reads
// lane i-1, lane i of wave_shl:1 reads lane i+1, across the 16-lane row boundaries; bound_ctrl gives 0 at the wave's ends.)
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/dpp_probe tests/tools/scratch/dpp_probe.hip && /tmp/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  const int v = 100 + threadIdx.x;
  out[threadIdx.x] = __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true);        // wave_shr:1
  out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true);   // wave_shl:1
}
int main() {
  int* d; int h[128];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) {
    const int want_r = i == 0 ? 0 : 100 + i - 1, want_l = i == 63 ? 0 : 100 + i + 1;
    if (h[i] != want_r || h[64 + i] != want_l) { ++bad; printf("lane %d: wave_shr:1 -> %d (want %d), wave_shl:1 -> %d (want %d)\n", i, h[i], want_r, h[64 + i], want_l); }
  }
  printf("dpp_probe: %s\n", bad ? "MISMATCH" : "wave_shr:1 reads lane-1, wave_shl:1 reads lane+1, zeros at the ends: OK");
  return bad != 0;
}
