// Shared epilogue of the MFMA convolution kernels (conv32_mfma.hip, conv4_mfma.hip).
// After the K loop a wave holds a 32-voxel x 32-channel tile in the v_mfma_f32_32x32x2_f32
// C/D layout: acc[r] = Z[voxel row(r,h)][channel li], row(r,h) = (r&3) + 8*(r>>2) + 4*h.
// Stores are 128-byte voxel lines (32 lanes x 4 B), two lines per store instruction.
#pragma once
#include "as_common.h"

struct EpilogueArgs {
  const float* bias;        // [32] or null — folded into the accumulator before the K loop
  float* z;                 // PCL output
  const float* ep_scale;    // epilogue 1: per-channel affine (eval BatchNorm) ...
  const float* ep_shift;
  const float* residual;    // ... + optional residual in the OUTPUT geometry (both epilogues)
  float* stat_mean;         // epilogue 0: per-workgroup (mean, M2) partials, or null
  float* stat_m2;
  int epilogue;             // 0 raw (+residual), 1 lrelu(acc*scale+shift) (+residual)
  float slope;
};

__device__ inline void conv_init_acc(f32x16& acc, const float* bias, int li) {
  const float bv = bias ? bias[li] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = bv;
}

// red: [4][32] floats, bmean: [32] floats of LDS.  All 256 threads of the workgroup must call.
__device__ inline void conv_epilogue(const f32x16& acc, const EpilogueArgs& e, int out_vox, bool valid, int M,
                                     float (*red)[32], float* bmean) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, li = lane & 31;
  if (e.epilogue == 1) {
    const float sc = e.ep_scale[li], sh = e.ep_shift[li];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      const int ov = __shfl(out_vox, row, 64);
      const int rv = __shfl((int)valid, row, 64);
      float yv = acc[r] * sc + sh;
      yv = yv > 0.f ? yv : yv * e.slope;
      if (rv) {
        if (e.residual) yv += e.residual[(long)ov * 32 + li];
        e.z[(long)ov * 32 + li] = yv;
      }
    }
    return;
  }

  float s1 = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    const int ov = __shfl(out_vox, row, 64);
    const int rv = __shfl((int)valid, row, 64);
    if (rv) {
      float v = acc[r];
      if (e.residual) v += e.residual[(long)ov * 32 + li];
      e.z[(long)ov * 32 + li] = v;
      s1 += acc[r];
    }
  }
  if (e.stat_mean == nullptr) return;

  // Per-workgroup (mean, M2) over its valid voxels, per channel: exact two-pass on the
  // register-resident tile; merged across workgroups by as_bn_finalize (Chan, fp64).
  const int first = blockIdx.x * 128;
  const int nvalid = min(128, M - first);
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
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    const int rv = __shfl((int)valid, row, 64);
    const float dv = acc[r] - mu;
    if (rv) s2 += dv * dv;
  }
  s2 += __shfl_xor(s2, 32, 64);
  __syncthreads();
  if (h == 0) red[wave][li] = s2;
  __syncthreads();
  if (threadIdx.x < 32) {
    e.stat_mean[blockIdx.x * 32 + li] = mu;
    e.stat_m2[blockIdx.x * 32 + li] = red[0][li] + red[1][li] + red[2][li] + red[3][li];
  }
}

// Output-voxel decode shared by the forward kernels: flattened index -> (b,d,y,x) -> PCL voxel
// index of the output and of the (strided) input anchor.
__device__ inline void conv_decode(int vc, const PclDev& gin, const PclDev& gout, int stride, int& in_vox, int& out_vox) {
  const int W = gout.W, H = gout.H, D = gout.D;
  int t = vc;
  const int x = t % W; t /= W;
  const int y = t % H; t /= H;
  const int d = t % D;
  const int b = t / D;
  in_vox = (int)gin.vox(b, d, y * stride, x * stride);
  out_vox = (int)gout.vox(b, d, y, x);
}

static inline int epilogue_args_ok(int epilogue, const float* sc, const float* sh, const float* sm, const float* s2) {
  if (!(epilogue == 0 || (epilogue == 1 && sc && sh))) return 0;
  if ((sm == nullptr) != (s2 == nullptr)) return 0;
  return 1;
}
