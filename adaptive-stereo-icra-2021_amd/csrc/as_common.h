// Shared helpers for libadaptive_stereo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/adaptive_stereo_hip.h"

#define AS_OK 0
#define AS_ERR_ARG (-1)
#define AS_ERR_LAUNCH (-2)

void as_set_error(const char* fmt, ...);
// slab reduction behind every 32->32 weight-gradient kernel (conv32_mfma.hip): dW[o][i][t] (+)= sum_chunks partial[chunk][t][i][o],
// db[o] (+)= sum_chunks partial_db[chunk][o]; recorded instead of launched while a deferral region is open (as_wgrad_defer)
void as_wgrad_reduce_enqueue(hipStream_t st, const float* partial, const float* partial_db, int nchunks, int T, float* dW,
                             float* db, int accumulate);
// measurement hook (optim.hip): event pair around a kernel launch; no-ops unless enabled
void as_prof_mark(int kernel_id, hipStream_t st, int begin, double flops);
// kernel ids of the measurement hook (bench.py reads them back with as_prof_read); the `flops` argument carries FLOPs for
// the matrix-core kernels and ALGORITHMIC BYTES (every tensor read once + written once) for the HBM-bound rows
enum {
  AS_PROF_CONV32 = 0, AS_PROF_WGRAD32 = 1, AS_PROF_CONV32_LDS = 2, AS_PROF_WGRAD32_LDS = 3, AS_PROF_BN_ACT_FWD = 4,
  AS_PROF_BN_BWD = 5, AS_PROF_CONV32_LDS_BNBWD = 6, AS_PROF_AGG3D = 7, AS_PROF_AGG_TAIL = 8, AS_PROF_WGRAD3D_LDS = 9,
  AS_PROF_COSTVOL_FWD = 10, AS_PROF_COSTVOL_BWD = 11, AS_PROF_OUTCONV_BWD = 12, AS_PROF_SOFTARGMAX_BWD = 13,
  AS_PROF_UPSAMPLE_FWD = 14, AS_PROF_UPSAMPLE_BWD = 15, AS_PROF_WARP_FWD = 16, AS_PROF_WARP_BWD = 17,
  AS_PROF_LOSS_FWD = 18, AS_PROF_LOSS_BWD = 19, AS_PROF_OUTCONV_FWD = 20, AS_PROF_SOFTARGMAX_FWD = 21, AS_PROF_BWD_FUSED = 22, AS_PROF_CONV_ACT = 23,
  AS_PROF_WINO_DGRAD = 24, AS_PROF_WINO_WGRAD = 25, AS_PROF_WINO_FWD = 26,    // the minimal-filtering kernels (conv32_wino*.hip)
  AS_PROF_WINO_BWD = 27,                                                       // ... both gradients in one launch (conv32_wino_bwd.hip)
  AS_PROF_IDS = 28
};

#define AS_CHECK_ARG(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      as_set_error(__VA_ARGS__);           \
      return AS_ERR_ARG;                   \
    }                                      \
  } while (0)

#define AS_CHECK_LAUNCH(name)                                              \
  do {                                                                     \
    hipError_t e_ = hipGetLastError();                                     \
    if (e_ != hipSuccess) {                                                \
      as_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));  \
      return AS_ERR_LAUNCH;                                                \
    }                                                                      \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Device-side view of a PCL tensor's geometry (all in voxels; a voxel = 32 floats).
struct PclDev {
  int B, D, H, W;
  int pd, ph, pw;
  int Dp, Hp, Wp;
  __host__ __device__ inline long vox(int b, int d, int y, int x) const {
    return (((long)b * Dp + d + pd) * Hp + y + ph) * (long)Wp + x + pw;
  }
};

static inline PclDev as_make_dev(const as_pcl* g) {
  PclDev p;
  p.B = g->B; p.D = g->D; p.H = g->H; p.W = g->W;
  p.pd = g->pd; p.ph = g->ph; p.pw = g->pw;
  p.Dp = g->D + 2 * g->pd; p.Hp = g->H + 2 * g->ph; p.Wp = g->W + 2 * g->pw;
  return p;
}

static inline bool as_pcl_ok(const as_pcl* g) {
  if (!g) return false;
  if (g->B <= 0 || g->D <= 0 || g->H <= 0 || g->W <= 0) return false;
  if (g->pd < 0 || g->ph < 0 || g->pw < 0) return false;
  const int64_t n = (int64_t)g->B * (g->D + 2 * g->pd) * (g->H + 2 * g->ph) * (g->W + 2 * g->pw) * 32;
  return n < ((int64_t)1 << 31);   // kernels index with 32-bit element offsets
}

// A "done once" flag PER HIP DEVICE: hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the current device only, so a
// process that drives a second GPU must set it there too (a process-wide flag skipped it: its launches asking for more than
// 64 KB of LDS then failed).
struct AsPerDevice {
  bool done[32] = {};
  static int dev() { int d = 0; return hipGetDevice(&d) == hipSuccess && d >= 0 && d < 32 ? d : -1; }
  bool get() const { const int d = dev(); return d >= 0 && done[d]; }
  void set() { const int d = dev(); if (d >= 0) done[d] = true; }
};

static inline int as_div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Wave-level sum over all 64 lanes (result in every lane).
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
