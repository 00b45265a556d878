// What does v_mfma_f32_32x32x2_f32 SUSTAIN on gfx950 once a loop looks like a real kernel's?  The table DESIGN.md quotes.
// One workgroup per CU (256 workgroups), one or two waves per SIMD (256 / 512 threads), 64 MFMAs per "tile" into four
// accumulators, on random operands (the clock the chip holds depends on the data).  Cases:
//   regs      operands in registers, nothing else in the loop
//   lds       the A operands of the next 16 MFMAs arrive by 4 x ds_read_b128 one step ahead (conflict-free, lane-linear)
//   lds+V6    ... plus 6 packed vector instructions per 4 MFMAs   (the weight gradient's transforms)
//   lds+V12   ... plus 12 per 4 MFMAs                              (twice that; the data gradient's main loop has 4-8)
//   +vm       ... plus one global load per tile, awaited with s_waitcnt vmcnt(0) at the end of the tile
//   split     two waves per SIMD with ROLES: waves 0-3 issue only MFMAs, waves 4-7 only vector instructions (v_pk_add_f32,
//             independent) — do the vector instructions of one wave hide under the other wave's fp32 MFMAs, or do their times add?
// Reported: cycles per MFMA and wave (s_memtime), the clock the chip held (s_memtime / s_memrealtime), TFLOP/s by HIP events.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_sustained tests/tools/scratch/mfma_sustained.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int LDSR, int NV, int VM, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void k_loop(const float* src, float* out, long long* cyc, long long* wall, int tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* lds = reinterpret_cast<f32x4*>(smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4* mine = lds + wave * 64 * 8 + lane;                 // 8 lane-linear f32x4 rows per wave
  for (int i = 0; i < 8; ++i) mine[i * 64] = *reinterpret_cast<const f32x4*>(src + ((threadIdx.x * 8 + i) * 4 & 65535));
  __syncthreads();
  f32x16 acc[4];
  for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float b[16];
  for (int i = 0; i < 16; ++i) b[i] = src[(threadIdx.x * 16 + i) & 65535];
  f32x4 xa[4];
  for (int m = 0; m < 4; ++m) xa[m] = mine[m * 64];
  f32x2 dd[6];
  for (int i = 0; i < 6; ++i) dd[i] = (f32x2){b[2 * i], b[2 * i + 1]};
  float gl = 0.f;
  const float* gsrc = src + (blockIdx.x * THREADS + threadIdx.x);
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int t = 0; t < tiles; ++t) {
    if (VM) asm volatile("global_load_dword %0, %1, off" : "=v"(gl) : "v"(gsrc + (long)(t & 63) * 131072) : "memory");
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 xn[4];
      if (LDSR) {
#pragma unroll
        for (int m = 0; m < 4; ++m) xn[m] = mine[(((q + 1) & 1) * 4 + m) * 64];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(dd[v % 6]) : "v"(dd[(v + 3) % 6]));      // (independent three apart)
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[c][e], b[4 * q + e], acc[c], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (LDSR) {
#pragma unroll
        for (int m = 0; m < 4; ++m) xa[m] = xn[m];
      }
    }
    if (VM) asm volatile("s_waitcnt vmcnt(0)" : "+v"(gl) :: "memory");
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  float s = gl;
  for (int i = 0; i < 6; ++i) s += dd[i].x + dd[i].y;
  for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * THREADS + threadIdx.x] = s;
  if (lane == 0) { cyc[blockIdx.x * 8 + wave] = t1 - t0; wall[blockIdx.x * 8 + wave] = w1 - w0; }
}

// waves 0-3: MFMAs only (64 per tile); waves 4-7: NVW packed vector instructions per tile, nothing else
template <int NVW>
__global__ __launch_bounds__(512, 1) void k_split(const float* src, float* out, long long* cyc, long long* wall, int tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float b[16];
  for (int i = 0; i < 16; ++i) b[i] = src[(threadIdx.x * 16 + i) & 65535];
  float s = 0.f;
  long long t0, t1, w0, w1;
  if (wave < 4) {
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    t0 = clock64(); w0 = wall_clock64();
    for (int t = 0; t < tiles; ++t) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[u], b[15 - u], acc[c], 0, 0, 0);
    }
    t1 = clock64(); w1 = wall_clock64();
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
  } else {
    f32x2 d[8];
    for (int i = 0; i < 8; ++i) d[i] = (f32x2){b[2 * i], b[2 * i + 1]};
    t0 = clock64(); w0 = wall_clock64();
    for (int t = 0; t < tiles; ++t) {
#pragma unroll
      for (int v = 0; v < NVW; ++v) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[v & 7]) : "v"(d[(v + 3) & 7]));
    }
    t1 = clock64(); w1 = wall_clock64();
    for (int i = 0; i < 8; ++i) s += d[i].x + d[i].y;
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (lane == 0) { cyc[blockIdx.x * 8 + wave] = t1 - t0; wall[blockIdx.x * 8 + wave] = w1 - w0; }
}

static float* g_src; static float* g_out; static long long *g_cyc, *g_wall;

static void report(const char* name, int threads, int tiles, float ms, int first_wave, int nwaves, double mfma_per_tile, const char* extra) {
  static long long hc[2048], hw[2048];
  (void)hipMemcpy(hc, g_cyc, sizeof(hc), hipMemcpyDeviceToHost); (void)hipMemcpy(hw, g_wall, sizeof(hw), hipMemcpyDeviceToHost);
  double c = 0, w = 0; int n = 0;
  for (int b = 0; b < 256; ++b) for (int i = first_wave; i < first_wave + nwaves; ++i) { c += hc[b * 8 + i]; w += hw[b * 8 + i]; ++n; }
  c /= n; w /= n;
  const double mf = mfma_per_tile * tiles;
  printf("%-34s %d wave(s)/SIMD  %6.1f cycles per MFMA and wave  clock %.2f GHz  %6.1f TFLOP/s (%.3f of 157.3)%s\n", name, threads / 256,
         mf > 0 ? c / mf : 0.0, c / w * 0.1, 256.0 * nwaves * mf * 4096 / (ms * 1e-3) / 1e12, 256.0 * nwaves * mf * 4096 / (ms * 1e-3) / 1e12 / 157.3, extra);
}

template <int LDSR, int NV, int VM, int THREADS> void run(const char* name, int tiles) {
  const int lds = THREADS * 8 * 16;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_loop<LDSR, NV, VM, THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {                     // (the first launch warms up; the best of the next three is reported)
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_loop<LDSR, NV, VM, THREADS>), dim3(256), dim3(THREADS), lds, 0, g_src, g_out, g_cyc, g_wall, tiles);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  report(name, THREADS, tiles, best, 0, THREADS / 64, 64.0, "");
}

template <int NVW> void run_split(const char* name, int tiles) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_split<NVW>), dim3(256), dim3(512), 0, 0, g_src, g_out, g_cyc, g_wall, tiles);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  static long long hc[2048];
  (void)hipMemcpy(hc, g_cyc, sizeof(hc), hipMemcpyDeviceToHost);
  double cv = 0; for (int b = 0; b < 256; ++b) for (int i = 4; i < 8; ++i) cv += hc[b * 8 + i]; cv /= 1024;
  char extra[160];
  snprintf(extra, sizeof(extra), "  | vector waves: %d v_pk_add_f32 per tile, %.1f cycles each", NVW, NVW ? cv / ((double)NVW * tiles) : 0.0);
  report(name, 512, tiles, best, 0, 4, 64.0, extra);
}

int main() {
  (void)hipMalloc(&g_src, 64 * 131072 * 4 + 65536 * 4); (void)hipMalloc(&g_out, 256 * 512 * 4);
  (void)hipMalloc(&g_cyc, 2048 * 8); (void)hipMalloc(&g_wall, 2048 * 8);
  {
    const size_t n = 64 * 131072 + 65536;
    float* h = (float*)malloc(n * 4);
    srand(1);
    for (size_t i = 0; i < n; ++i) h[i] = (float)(rand() & 0xffff) / 65536.f - 0.5f;
    (void)hipMemcpy(g_src, h, n * 4, hipMemcpyHostToDevice); free(h);
  }
  const int T = 3000;                                     // 192,000 MFMAs per wave: 5-10 ms per launch
  run<0, 0, 0, 256>("regs", T);            run<0, 0, 0, 512>("regs", T);
  run<1, 0, 0, 256>("lds", T);             run<1, 0, 0, 512>("lds", T);
  run<1, 6, 0, 256>("lds + 6 VALU / 4 MFMA", T);   run<1, 6, 0, 512>("lds + 6 VALU / 4 MFMA", T);
  run<1, 12, 0, 256>("lds + 12 VALU / 4 MFMA", T); run<1, 12, 0, 512>("lds + 12 VALU / 4 MFMA", T);
  run<1, 0, 1, 256>("lds + vmcnt(0) per tile", T); run<1, 0, 1, 512>("lds + vmcnt(0) per tile", T);
  run<1, 6, 1, 256>("lds + 6 VALU + vmcnt(0)", T); run<1, 6, 1, 512>("lds + 6 VALU + vmcnt(0)", T);
  run<1, 12, 1, 256>("lds + 12 VALU + vmcnt(0)", T); run<1, 12, 1, 512>("lds + 12 VALU + vmcnt(0)", T);
  run_split<0>("split: MFMA waves | idle waves", T);
  run_split<96>("split: MFMA | 96 VALU per tile", T);
  run_split<192>("split: MFMA | 192 VALU per tile", T);
  run_split<384>("split: MFMA | 384 VALU per tile", T);
  run_split<768>("split: MFMA | 768 VALU per tile", T);
  return 0;
}
