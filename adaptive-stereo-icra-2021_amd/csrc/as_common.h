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
// measurement hook (optim.hip): event pair around a kernel launch; no-ops unless enabled
void as_prof_mark(int kernel_id, hipStream_t st, int begin, double flops);

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
