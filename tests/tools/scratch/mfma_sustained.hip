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

// waves 0-3: MFMAs only (64 per tile); waves 4-7: NVW vector instructions of kind OP per tile, nothing else
// OP: 0 v_pk_add_f32, 1 v_add_f32, 2 v_fma_f32, 3 v_cndmask_b32, 4 v_mul_f32, 5 v_pk_fma_f32, 6 v_pk_mul_f32, 7 v_mov_b32,
//     8 v_cmp_gt_f32, 9 v_add_u32, 10 ds_read_b32 (an LDS instruction's issue), 11 v_max_f32, 12 s_nop 0
template <int NVW, int OP = 0>
__global__ __launch_bounds__(512, 1) void k_split(const float* src, float* out, long long* cyc, long long* wall, int tiles) {
  __shared__ float lds_s[512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  lds_s[threadIdx.x] = src[threadIdx.x];
  __syncthreads();
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
    float e[8];
    for (int i = 0; i < 8; ++i) e[i] = b[i];
    unsigned laddr = (unsigned)(threadIdx.x * 4);
    unsigned laddr4 = (unsigned)((threadIdx.x & 63) * 16);
    unsigned sc[4] = {1u, 2u, 3u, 4u};
    unsigned long long msk = 0x00ff00ff00ff00ffull;
    f32x4 q4[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    t0 = clock64(); w0 = wall_clock64();
    for (int t = 0; t < tiles; ++t) {
#pragma unroll
      for (int v = 0; v < NVW; ++v) {
        if (OP == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[v & 7]) : "v"(d[(v + 3) & 7]));
        else if (OP == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(e[v & 7]) : "v"(e[(v + 3) & 7]));
        else if (OP == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(e[v & 7]) : "v"(e[(v + 3) & 7]));
        else if (OP == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(e[v & 7]) : "v"(e[(v + 3) & 7]));
        else if (OP == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(e[v & 7]) : "v"(e[(v + 3) & 7]));
        else if (OP == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[v & 7]) : "v"(d[(v + 3) & 7]));
        else if (OP == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[v & 7]) : "v"(d[(v + 3) & 7]));
        else if (OP == 7) asm volatile("v_mov_b32 %0, %1" : "+v"(e[v & 7]) : "v"(e[(v + 3) & 7]));
        else if (OP == 8) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(e[v & 7]), "v"(e[(v + 3) & 7]) : "vcc");
        else if (OP == 9) asm volatile("v_add_u32 %0, %0, %1" : "+v"(e[v & 7]) : "v"(e[(v + 3) & 7]));
        else if (OP == 10) { asm volatile("ds_read_b32 %0, %1" : "=v"(e[v & 7]) : "v"(laddr)); if ((v & 7) == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        else if (OP == 11) asm volatile("v_max_f32 %0, %0, %1" : "+v"(e[v & 7]) : "v"(e[(v + 3) & 7]));
        else if (OP == 12) asm volatile("s_nop 0");
        else if (OP == 13) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc[v & 3]));
        else if (OP == 14) asm volatile("s_mul_i32 %0, %0, 3" : "+s"(sc[v & 3]));
        else if (OP == 15) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(e[v & 7]) : "v"(e[(v + 3) & 7]), "s"(msk));
        else if (OP == 16) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(e[v & 7]) : "v"(e[(v + 3) & 7]));
        else if (OP == 17) asm volatile("v_cvt_f32_i32 %0, %1" : "+v"(e[v & 7]) : "v"(e[(v + 3) & 7]));
        else if (OP == 18) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sc[v & 3]) : "v"(e[v & 7]));
        else if (OP == 19) asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(e[v & 7]) : "s"(sc[v & 3]));
        else if (OP == 20) asm volatile("s_waitcnt lgkmcnt(0)");
        else if (OP == 21) asm volatile("v_pk_mov_b32 %0, %1, %1" : "+v"(d[v & 7]) : "v"(d[(v + 3) & 7]));
        else if (OP == 22) asm volatile("v_mul_f32_e64 %0, %0, %1 clamp" : "+v"(e[v & 7]) : "v"(e[(v + 3) & 7]));
        else if (OP == 23) asm volatile("ds_write_b32 %1, %0" :: "v"(e[v & 7]), "v"(laddr) : "memory");
        else if (OP == 24) { asm volatile("ds_read_b128 %0, %1" : "=v"(q4[v & 1]) : "v"(laddr4)); if ((v & 7) == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        else if (OP == 25) { asm volatile("global_load_dword %0, %1, off" : "=v"(e[v & 7]) : "v"(src + threadIdx.x + (v & 7) * 512) : "memory"); if ((v & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        else if (OP == 26) asm volatile("v_med3_i32 %0, %0, %1, %1" : "+v"(e[v & 7]) : "v"(e[(v + 3) & 7]));
        else if (OP == 27) asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(msk) : "v"(e[v & 7]), "v"(e[(v + 3) & 7]));
        else if (OP == 28) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(e[v & 7]) : "v"(e[(v + 3) & 7]));
        else asm volatile("s_nop 0");
      }
    }
    t1 = clock64(); w1 = wall_clock64();
    for (int i = 0; i < 8; ++i) s += d[i].x + d[i].y + e[i];
    s += (float)(sc[0] + sc[1] + sc[2] + sc[3]) + (float)(msk & 7) + q4[0].x + q4[1].y;
  }
  out[blockIdx.x * 512 + threadIdx.x] = s + lds_s[(threadIdx.x + 1) & 511];
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

template <int NVW, int OP = 0> void run_split(const char* name, int tiles) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_split<NVW, OP>), dim3(256), dim3(512), 0, 0, g_src, g_out, g_cyc, g_wall, tiles);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  static long long hc[2048];
  (void)hipMemcpy(hc, g_cyc, sizeof(hc), hipMemcpyDeviceToHost);
  double cv = 0; for (int b = 0; b < 256; ++b) for (int i = 4; i < 8; ++i) cv += hc[b * 8 + i]; cv /= 1024;
  char extra[160];
  // what one instruction of the partner costs the SIMD: (kernel time - the MFMA waves' time) / instructions, in cycles at the clock held
  static long long hw[2048];
  (void)hipMemcpy(hw, g_wall, sizeof(hw), hipMemcpyDeviceToHost);
  double cm = 0, wm = 0; for (int b = 0; b < 256; ++b) for (int i = 0; i < 4; ++i) { cm += hc[b * 8 + i]; wm += hw[b * 8 + i]; }
  const double ghz = cm / wm * 0.1;
  const double simd_cycles_per_tile = best * 1e-3 * ghz * 1e9 / tiles;
  snprintf(extra, sizeof(extra), "  | partner: %d per tile; SIMD time per tile %.0f cycles = 4096 + %.2f per partner instruction", NVW,
           simd_cycles_per_tile, NVW ? (simd_cycles_per_tile - 4096.0) / NVW : 0.0);
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
  run_split<384, 0>("split 384: v_pk_add_f32", T);   run_split<384, 1>("split 384: v_add_f32", T);
  run_split<384, 2>("split 384: v_fma_f32", T);      run_split<384, 3>("split 384: v_cndmask_b32", T);
  run_split<384, 4>("split 384: v_mul_f32", T);      run_split<384, 5>("split 384: v_pk_fma_f32", T);
  run_split<384, 6>("split 384: v_pk_mul_f32", T);   run_split<384, 7>("split 384: v_mov_b32", T);
  run_split<384, 8>("split 384: v_cmp_gt_f32", T);   run_split<384, 9>("split 384: v_add_u32", T);
  run_split<384, 10>("split 384: ds_read_b32", T);   run_split<384, 11>("split 384: v_max_f32", T);
  run_split<384, 12>("split 384: s_nop 0", T);
  run_split<384, 13>("split 384: s_add_u32", T);     run_split<384, 14>("split 384: s_mul_i32", T);
  run_split<384, 15>("split 384: v_cndmask_b32_e64 (sgpr mask)", T);   run_split<384, 16>("split 384: v_med3_f32", T);
  run_split<384, 17>("split 384: v_cvt_f32_i32", T); run_split<384, 18>("split 384: v_readlane_b32", T);
  run_split<384, 19>("split 384: v_writelane_b32", T); run_split<384, 20>("split 384: s_waitcnt lgkmcnt(0)", T);
  run_split<384, 21>("split 384: v_pk_mov_b32", T);  run_split<384, 22>("split 384: v_mul_f32 clamp", T);
  run_split<384, 23>("split 384: ds_write_b32", T);  run_split<384, 24>("split 384: ds_read_b128", T);
  run_split<384, 25>("split 384: global_load_dword", T); run_split<384, 26>("split 384: v_med3_i32", T);
  run_split<384, 27>("split 384: v_cmp_gt_f32_e64 (sgpr)", T); run_split<384, 28>("split 384: v_sub_f32", T);
  return 0;
}
