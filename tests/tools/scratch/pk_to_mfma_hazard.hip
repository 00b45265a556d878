// How many wait states must lie between an inline-asm v_pk_add_f32 and a v_mfma_f32_32x32x2_f32 that reads its result?
// (hipcc pads its own instructions for this hazard, not inline asm.)  For K = 0..10 wait states the same sequence is run
// with a result register that holds the PREVIOUS iteration's value until the packed add has written it; the output is
// compared with K = 15.  build + run (GPU box):  hipcc --offload-arch=gfx950 -O2 pk_to_mfma_hazard.hip -o /tmp/pkh && /tmp/pkh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SEQ(NOPS)                                                                                         \
  asm volatile("v_mov_b32 v100, %[a0]\n\tv_mov_b32 v101, %[a1]\n\tv_mov_b32 v102, %[b0]\n\tv_mov_b32 v103, %[b1]\n\t" \
               "s_nop 7\n\t"                                                                              \
               "v_pk_add_f32 v[104:105], v[100:101], v[102:103] neg_lo:[0,1] neg_hi:[0,1]\n\t"            \
               NOPS                                                                                       \
               "v_mfma_f32_32x32x2_f32 %[acc], v104, %[one], %[acc]\n\t"                                  \
               "s_nop 15\n\ts_nop 7\n\t"                                                                  \
               : [acc] "+v"(acc) : [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1), [one] "v"(one)  \
               : "v100", "v101", "v102", "v103", "v104", "v105")

template <int K> __global__ void kern(const float* in, float* out, int iters) {
  const int lane = threadIdx.x;
  float a0 = in[lane], a1 = in[64 + lane], b0 = in[128 + lane], b1 = in[192 + lane], one = 1.0f;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int it = 0; it < iters; ++it) {
    a0 = a0 * 1.0009765625f + 0.25f;                      // a new value every iteration: a stale v104 is a different number
    if (K == 0) SEQ("");
    else if (K == 1) SEQ("s_nop 0\n\t");
    else if (K == 2) SEQ("s_nop 1\n\t");
    else if (K == 3) SEQ("s_nop 2\n\t");
    else if (K == 4) SEQ("s_nop 3\n\t");
    else if (K == 5) SEQ("s_nop 4\n\t");
    else if (K == 6) SEQ("s_nop 5\n\t");
    else if (K == 7) SEQ("s_nop 6\n\t");
    else if (K == 8) SEQ("s_nop 7\n\t");
    else if (K == 10) SEQ("s_nop 9\n\t");
    else SEQ("s_nop 14\n\t");
  }
  for (int r = 0; r < 16; ++r) out[r * 64 + lane] = acc[r];
}

template <int K> static void run(const float* din, float* dout, float* h, int iters) {
  hipLaunchKernelGGL(kern<K>, dim3(1), dim3(64), 0, 0, din, dout, iters);
  hipMemcpy(h, dout, 1024 * 4, hipMemcpyDeviceToHost);
}

int main() {
  float hin[256], ref[1024], got[1024];
  for (int i = 0; i < 256; ++i) hin[i] = 0.37f * (float)((i * 2654435761u) % 1000) / 1000.f + 0.1f;
  float *din, *dout;
  hipMalloc(&din, sizeof(hin)); hipMalloc(&dout, 1024 * 4);
  hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
  const int iters = 200;
  run<15>(din, dout, ref, iters);
#define TRY(K) { int bad = 0; for (int rep = 0; rep < 20; ++rep) { run<K>(din, dout, got, iters); bad += memcmp(ref, got, sizeof(ref)) != 0; } \
                 printf("%2d wait states between v_pk_add_f32 (asm) and the MFMA that reads it: %s (%d of 20 runs differ)\n", K, bad ? "WRONG" : "ok", bad); }
  TRY(0) TRY(1) TRY(2) TRY(3) TRY(4) TRY(5) TRY(6) TRY(7) TRY(8) TRY(10)
  return 0;
}
