// a2 — difference cost volume and its adjoint.
// Reference semantics: adaptive_stereo/models/stereo_net.py:173-184
//   cost[b,c,d,y,x] = L[b,c,y,x] - R[b,c,y,x-d] for x >= d, else 0
// (a python loop of Dc slice-assignments into a zero-filled NCDHW tensor there).
//
// MI355X design: one launch; features arrive NCHW (W fastest), the volume leaves
// in PCL (channel fastest).  A workgroup owns a 16-pixel segment of one image row:
// it stages the L segment and the R segment (plus D-1 pixels of left context)
// through LDS, transposing W-major -> channel-major on the way, so global reads
// are coalesced along W and global writes are full 128-byte voxel lines (float4
// per lane).  LDS rows are padded to 33 floats: conflict-free on both sides.
// HBM-bound: algorithmic traffic = 2*F read + V written.
#include "as_common.h"

#define CV_TX 16        // 5 x H x B workgroups at W=78: the volume is small (2.9 MB per pair), parallelism matters more than halo reuse
#define CV_MAXD 64
#define CV_LDS_STRIDE 33

__global__ __launch_bounds__(256) void cost_volume_fwd_kernel(const float* __restrict__ L,
                                                               const float* __restrict__ R,
                                                               float* __restrict__ vol, PclDev g) {
  __shared__ float sL[CV_TX * CV_LDS_STRIDE];
  __shared__ float sR[(CV_TX + CV_MAXD - 1) * CV_LDS_STRIDE];
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * CV_TX, y = blockIdx.y, b = blockIdx.z;
  const int D = g.D, H = g.H, W = g.W;
  const long plane = (long)H * W;
  const float* Lrow = L + (long)b * 32 * plane + (long)y * W;
  const float* Rrow = R + (long)b * 32 * plane + (long)y * W;

  for (int i = tid; i < 32 * CV_TX; i += 256) {
    const int c = i / CV_TX, xx = i % CV_TX, x = x0 + xx;
    sL[xx * CV_LDS_STRIDE + c] = (x < W) ? Lrow[c * plane + x] : 0.f;
  }
  const int rcols = CV_TX + D - 1;
  for (int i = tid; i < 32 * rcols; i += 256) {
    const int c = i / rcols, j = i % rcols, xs = x0 - (D - 1) + j;
    sR[j * CV_LDS_STRIDE + c] = (xs >= 0 && xs < W) ? Rrow[c * plane + xs] : 0.f;
  }
  __syncthreads();

  const int items = D * CV_TX * 8;
  for (int i = tid; i < items; i += 256) {
    const int c4 = i & 7, xx = (i >> 3) % CV_TX, d = (i >> 3) / CV_TX;
    const int x = x0 + xx;
    if (x >= W) continue;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (x >= d) {
      const float* l = sL + xx * CV_LDS_STRIDE + c4 * 4;
      const float* r = sR + (xx + (D - 1) - d) * CV_LDS_STRIDE + c4 * 4;
      v.x = l[0] - r[0]; v.y = l[1] - r[1]; v.z = l[2] - r[2]; v.w = l[3] - r[3];
    }
    *reinterpret_cast<f32x4*>(vol + g.vox(b, d, y, x) * 32 + c4 * 4) = v;
  }
}

// Adjoint: gL[b,c,y,x] = sum_{d<=x} g[b,d,y,x,c];  gR[b,c,y,x'] = -sum_{d, x'+d<W} g[b,d,y,x'+d,c].
__global__ __launch_bounds__(256) void cost_volume_bwd_kernel(const float* __restrict__ gvol,
                                                               float* __restrict__ gL,
                                                               float* __restrict__ gR, PclDev g) {
  __shared__ float sOut[CV_TX * CV_LDS_STRIDE];
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * CV_TX, y = blockIdx.y, b = blockIdx.z;
  const int D = g.D, H = g.H, W = g.W;
  const long plane = (long)H * W;

  for (int pass = 0; pass < 2; ++pass) {
    for (int i = tid; i < CV_TX * 8; i += 256) {
      const int c4 = i & 7, xx = i >> 3, x = x0 + xx;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (x < W) {
        if (pass == 0) {
          const int dmax = (x < D - 1) ? x : D - 1;
          for (int d = 0; d <= dmax; ++d)
            acc += *reinterpret_cast<const f32x4*>(gvol + g.vox(b, d, y, x) * 32 + c4 * 4);
        } else {
          for (int d = 0; d < D && x + d < W; ++d)
            acc -= *reinterpret_cast<const f32x4*>(gvol + g.vox(b, d, y, x + d) * 32 + c4 * 4);
        }
      }
      float* o = sOut + xx * CV_LDS_STRIDE + c4 * 4;
      o[0] = acc.x; o[1] = acc.y; o[2] = acc.z; o[3] = acc.w;
    }
    __syncthreads();
    float* dst = (pass == 0 ? gL : gR) + (long)b * 32 * plane + (long)y * W;
    for (int i = tid; i < 32 * CV_TX; i += 256) {
      const int c = i / CV_TX, xx = i % CV_TX, x = x0 + xx;
      if (x < W) dst[c * plane + x] = sOut[xx * CV_LDS_STRIDE + c];
    }
    __syncthreads();
  }
}

static int check_cv(const as_pcl* g, const char* who) {
  AS_CHECK_ARG(as_pcl_ok(g), "%s: bad geometry", who);
  AS_CHECK_ARG(g->D <= CV_MAXD, "%s: D=%d exceeds %d", who, g->D, CV_MAXD);
  AS_CHECK_ARG(g->H <= 65535 && g->B <= 65535, "%s: H or B exceeds grid limits", who);
  return AS_OK;
}

extern "C" int as_cost_volume_fwd(const float* L, const float* R, float* vol, const as_pcl* g, void* stream) {
  if (int e = check_cv(g, "as_cost_volume_fwd")) return e;
  AS_CHECK_ARG(L && R && vol, "as_cost_volume_fwd: null pointer");
  dim3 grid(as_div_up(g->W, CV_TX), g->H, g->B);
  as_prof_mark(AS_PROF_COSTVOL_FWD, (hipStream_t)stream, 1, 0.0);
  hipLaunchKernelGGL(cost_volume_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, L, R, vol, as_make_dev(g));
  as_prof_mark(AS_PROF_COSTVOL_FWD, (hipStream_t)stream, 0, 2.0 * 128.0 * (double)g->B * g->H * g->W + 128.0 * (double)g->B * g->D * g->H * g->W);
  AS_CHECK_LAUNCH("as_cost_volume_fwd");
  return AS_OK;
}

extern "C" int as_cost_volume_bwd(const float* gvol, float* gL, float* gR, const as_pcl* g, void* stream) {
  if (int e = check_cv(g, "as_cost_volume_bwd")) return e;
  AS_CHECK_ARG(gvol && gL && gR, "as_cost_volume_bwd: null pointer");
  dim3 grid(as_div_up(g->W, CV_TX), g->H, g->B);
  as_prof_mark(AS_PROF_COSTVOL_BWD, (hipStream_t)stream, 1, 0.0);
  hipLaunchKernelGGL(cost_volume_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, gvol, gL, gR, as_make_dev(g));
  as_prof_mark(AS_PROF_COSTVOL_BWD, (hipStream_t)stream, 0, 2.0 * 128.0 * (double)g->B * g->H * g->W + 128.0 * (double)g->B * g->D * g->H * g->W);
  AS_CHECK_LAUNCH("as_cost_volume_bwd");
  return AS_OK;
}
