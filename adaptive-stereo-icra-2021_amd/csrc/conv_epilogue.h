// Shared epilogue of the MFMA convolution kernels (conv32_mfma.hip, conv4_mfma.hip).
// After the K loop a wave holds a 32-voxel x 32-channel tile in the v_mfma_f32_32x32x2_f32
// C/D layout: acc[r] = Z[voxel row(r,h)][channel li], row(r,h) = (r&3) + 8*(r>>2) + 4*h.
// Stores are 128-byte voxel lines (32 lanes x 4 B), two lines per store instruction.
#pragma once
#include "as_common.h"

// 64 floats that swallow stores a kernel issues only to keep its per-wave store count exact (conv4_mfma.hip)
static __device__ float as_store_dump[64];     // one copy per translation unit (no relocatable device code)

struct EpilogueArgs {
  const float* bias;        // [32] or null — folded into the accumulator before the K loop
  float* z;                 // PCL output
  const float* ep_scale;    // epilogue 1: per-channel affine (eval BatchNorm) ...
  const float* ep_shift;
  const float* residual;    // ... + optional residual in the OUTPUT geometry (both epilogues)
  float* stat_mean;         // epilogue 0: per-partial (mean, M2, count) BatchNorm moments, or null
  float* stat_m2;
  float* stat_cnt;
  int epilogue;             // 0 raw (+residual), 1 lrelu(acc*scale+shift) (+residual)
  float slope;
};

__device__ inline void conv_init_acc(f32x16& acc, const float* bias, int li) {
  const float bv = bias ? bias[li] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = bv;
}

// Moments of one workgroup tile, valid in threads 0..31 (channel = threadIdx.x) after conv_epilogue.
struct TileStats { float n, mean, m2; };

// Chan's pairwise merge (fp32; used by persistent kernels to fold their tiles into one partial).
__device__ inline void stats_merge(TileStats& run, const TileStats& t) {
  if (t.n <= 0.f) return;
  const float tot = run.n + t.n, delta = t.mean - run.mean;
  run.mean += delta * (t.n / tot);
  run.m2 += t.m2 + delta * delta * (run.n * t.n / tot);
  run.n = tot;
}

// red: [4][32] floats, bmean: [32] floats of LDS.  All 256 threads of the workgroup must call.
// nvalid = number of valid voxels in this workgroup's tile.  ts (may be null) receives the tile's moments.
__device__ inline void conv_epilogue(const f32x16& acc, const EpilogueArgs& e, int out_vox, bool valid, int nvalid,
                                     float (*red)[32], float* bmean, TileStats* ts) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, li = lane & 31;
  // Row addresses and the residual (if any) are fetched for all 16 rows BEFORE the store loop: a load
  // inside the per-row "if (row valid)" branch is not hoisted by hipcc and costs one dependent L2
  // round trip per row.  Invalid rows carry a clamped (in-bounds) address, so the loads are safe.
  int ov[16], rv[16];
  float res[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    ov[r] = __shfl(out_vox, row, 64);
    rv[r] = __shfl((int)valid, row, 64);
  }
  if (e.residual) {
#pragma unroll
    for (int r = 0; r < 16; ++r) res[r] = e.residual[(long)ov[r] * 32 + li];
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) res[r] = 0.f;
  }

  if (e.epilogue == 1) {
    const float sc = e.ep_scale[li], sh = e.ep_shift[li];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float yv = acc[r] * sc + sh;
      yv = yv > 0.f ? yv : yv * e.slope;
      if (rv[r]) e.z[(long)ov[r] * 32 + li] = yv + res[r];
    }
    return;
  }

  float s1 = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (rv[r]) {
      e.z[(long)ov[r] * 32 + li] = acc[r] + res[r];
      s1 += acc[r];
    }
  }
  if (e.stat_mean == nullptr) return;

  // Per-tile (mean, M2) over its valid voxels, per channel: exact two-pass on the register-resident
  // tile; merged across tiles/workgroups by Chan's formula (stats_merge, then as_bn_finalize in fp64).
  s1 += __shfl_xor(s1, 32, 64);
  if (h == 0) red[wave][li] = s1;
  __syncthreads();
  if (threadIdx.x < 32)
    bmean[li] = (red[0][li] + red[1][li] + red[2][li] + red[3][li]) / (float)nvalid;
  __syncthreads();
  const float mu = bmean[li];
  float s2 = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float dv = acc[r] - mu;
    if (rv[r]) s2 += dv * dv;
  }
  s2 += __shfl_xor(s2, 32, 64);
  __syncthreads();
  if (h == 0) red[wave][li] = s2;
  __syncthreads();
  if (threadIdx.x < 32 && ts) {
    ts->n = (float)nvalid; ts->mean = mu;
    ts->m2 = red[0][li] + red[1][li] + red[2][li] + red[3][li];
  }
}

// One partial per workgroup (flattened kernels): partial index = blockIdx.x.
__device__ inline void stats_write(const EpilogueArgs& e, int part, const TileStats& ts) {
  if (e.stat_mean == nullptr || threadIdx.x >= 32) return;
  e.stat_mean[part * 32 + threadIdx.x] = ts.mean;
  e.stat_m2[part * 32 + threadIdx.x] = ts.m2;
  if (threadIdx.x == 0) e.stat_cnt[part] = ts.n;
}

// Output-voxel decode shared by the forward kernels: flattened index over the launch's logical
// extent (B, D, Hl, Wl) -> PCL voxel index of the output and of the input anchor.
//   forward conv : input anchor = (y*in_stride, x*in_stride), output = (y, x)
//   phase of a stride-2 data gradient: input anchor = (y, x), output = (y*out_stride+oy, x*out_stride+ox)
struct ConvMap {
  int Hl, Wl;                 // logical extent of this launch (H and W)
  int in_stride, out_stride, out_oy, out_ox;
};

__device__ inline void conv_decode(int vc, const PclDev& gin, const PclDev& gout, const ConvMap& m, int& in_vox, int& out_vox) {
  const int D = gout.D;
  int t = vc;
  const int x = t % m.Wl; t /= m.Wl;
  const int y = t % m.Hl; t /= m.Hl;
  const int d = t % D;
  const int b = t / D;
  in_vox = (int)gin.vox(b, d, y * m.in_stride, x * m.in_stride);
  out_vox = (int)gout.vox(b, d, y * m.out_stride + m.out_oy, x * m.out_stride + m.out_ox);
}

static inline int epilogue_args_ok(int epilogue, const float* sc, const float* sh, const float* sm, const float* s2,
                                   const float* cnt) {
  if (!(epilogue == 0 || (epilogue == 1 && sc && sh))) return 0;
  if ((sm == nullptr) != (s2 == nullptr) || (sm == nullptr) != (cnt == nullptr)) return 0;
  return 1;
}
