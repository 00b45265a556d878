// a6 / a7-head: bilinear up-sampling (align_corners=False), and a9: LinearWarping.
// Reference semantics:
//   F.interpolate(x, size, mode="bilinear", align_corners=False) * gain     stereo_net.py:106-114, 201-202
//   LinearWarping.forward: grid (x -/+ d, y), normalised 2x/w-1, 2y/h-1, then
//   F.grid_sample(bilinear, padding_mode="border", align_corners=False) and the
//   validity mask -1 <= g <= 1                                              models/linear_warping.py:18-57
// Both are HBM-bound gathers; one output pixel per lane, W-coalesced.  Arithmetic
// follows ATen's order of operations (source index = scale*(dst+0.5)-0.5 clamped at 0;
// grid un-normalisation ((g+1)*size-1)/2; border clip with zero gradient at the clip).
#include "as_common.h"

__device__ inline void bilin_src(float scale, int dst, int in_size, int& i0, int& i1, float& l0, float& l1) {
  float r = scale * ((float)dst + 0.5f) - 0.5f;
  r = r < 0.f ? 0.f : r;
  i0 = (int)r;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = r - (float)i0;
  l0 = 1.f - l1;
}

__global__ __launch_bounds__(256) void upsample_fwd_kernel(const float* __restrict__ src, int B, int h, int w,
                                                            float* __restrict__ dst, int H, int W, float gain) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * H * W) return;
  const int X = i % W, Y = (i / W) % H, b = i / ((long)W * H);
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
  bilin_src(sh, Y, h, y0, y1, ly0, ly1);
  bilin_src(sw, X, w, x0, x1, lx0, lx1);
  const float* s = src + (long)b * h * w;
  const float top = lx0 * s[y0 * w + x0] + lx1 * s[y0 * w + x1];
  const float bot = lx0 * s[y1 * w + x0] + lx1 * s[y1 * w + x1];
  dst[i] = (ly0 * top + ly1 * bot) * gain;
}

// Adjoint in gather form (deterministic): one wave per coarse pixel, lanes sweep the fine
// footprint, then a wavefront-shuffle sum.
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ g_dst, int B, int H, int W,
                                                            float* __restrict__ g_src, int h, int w, float gain) {
  const int lane = threadIdx.x & 63;
  const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= (long)B * h * w) return;
  const int j = pix % w, i = (pix / w) % h, b = pix / ((long)w * h);
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  // fine rows/cols whose two source taps can include (i, j)
  int Y0 = (int)floorf(((float)i - 1.f + 0.5f) / sh - 0.5f) - 1, Y1 = (int)ceilf(((float)i + 1.f + 0.5f) / sh - 0.5f) + 1;
  int X0 = (int)floorf(((float)j - 1.f + 0.5f) / sw - 0.5f) - 1, X1 = (int)ceilf(((float)j + 1.f + 0.5f) / sw - 0.5f) + 1;
  Y0 = max(Y0, 0); X0 = max(X0, 0); Y1 = min(Y1, H - 1); X1 = min(X1, W - 1);
  const int ny = Y1 - Y0 + 1, nx = X1 - X0 + 1;
  const float* g = g_dst + (long)b * H * W;
  float acc = 0.f;
  for (int k = lane; k < ny * nx; k += 64) {
    const int Y = Y0 + k / nx, X = X0 + k % nx;
    int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
    bilin_src(sh, Y, h, y0, y1, ly0, ly1);
    bilin_src(sw, X, w, x0, x1, lx0, lx1);
    const float wy = (y0 == i ? ly0 : 0.f) + (y1 == i ? ly1 : 0.f);
    const float wx = (x0 == j ? lx0 : 0.f) + (x1 == j ? lx1 : 0.f);
    acc += wy * wx * g[(long)Y * W + X];
  }
  acc = wave_sum(acc);
  if (lane == 0) g_src[pix] = acc * gain;
}

// ---- LinearWarping -------------------------------------------------------------------------
struct WarpCoord {
  float ix, iy;       // clipped sample position
  float mx, my;       // d(clipped)/d(unclipped): 0 or 1
  int valid;
};

__device__ inline float clip_border(float v, int size, float& mult) {
  const float hi = (float)(size - 1);
  if (v <= 0.f) { mult = 0.f; return 0.f; }
  if (v >= hi) { mult = 0.f; return hi; }
  mult = 1.f;
  return v;
}

__device__ inline WarpCoord warp_coord(int x, int y, float d, int H, int W, int r2l) {
  WarpCoord c;
  const float fx = r2l ? (float)x - d : (float)x + d;
  const float fy = (float)y;
  const float nx = (2.f * fx) / (float)W - 1.0f;
  const float ny = (2.f * fy) / (float)H - 1.0f;
  c.valid = (nx >= -1.0f && nx <= 1.0f && ny >= -1.0f && ny <= 1.0f) ? 1 : 0;
  const float ux = ((nx + 1.f) * (float)W - 1.f) / 2.f;
  const float uy = ((ny + 1.f) * (float)H - 1.f) / 2.f;
  c.ix = clip_border(ux, W, c.mx);
  c.iy = clip_border(uy, H, c.my);
  return c;
}

__global__ __launch_bounds__(256) void warp_fwd_kernel(const float* __restrict__ img, const float* __restrict__ disp,
                                                        int B, int C, int H, int W, int r2l,
                                                        float* __restrict__ warped, uint8_t* __restrict__ mask) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * H * W) return;
  const int x = i % W, y = (i / W) % H, b = i / ((long)W * H);
  const WarpCoord c = warp_coord(x, y, disp[i], H, W, r2l);
  const float fx0 = floorf(c.ix), fy0 = floorf(c.iy);
  const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
  const float wx1 = c.ix - fx0, wx0 = (fx0 + 1.f) - c.ix;
  const float wy1 = c.iy - fy0, wy0 = (fy0 + 1.f) - c.iy;
  const bool bx1 = x1 <= W - 1, by1 = y1 <= H - 1;   // x0,y0 are always in range after the clip
  const long plane = (long)H * W;
  for (int ch = 0; ch < C; ++ch) {
    const float* p = img + ((long)b * C + ch) * plane;
    const float nw = p[(long)y0 * W + x0];
    const float ne = bx1 ? p[(long)y0 * W + x1] : 0.f;
    const float sw = by1 ? p[(long)y1 * W + x0] : 0.f;
    const float se = (bx1 && by1) ? p[(long)y1 * W + x1] : 0.f;
    warped[((long)b * C + ch) * plane + (long)y * W + x] =
        nw * (wx0 * wy0) + ne * (wx1 * wy0) + sw * (wx0 * wy1) + se * (wx1 * wy1);
  }
  if (mask) mask[i] = (uint8_t)c.valid;
}

__global__ __launch_bounds__(256) void warp_bwd_kernel(const float* __restrict__ g_warped, const float* __restrict__ img,
                                                        const float* __restrict__ disp, const float* __restrict__ add_src,
                                                        int B, int C, int H, int W, int r2l, float* __restrict__ g_disp) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * H * W) return;
  const int x = i % W, y = (i / W) % H, b = i / ((long)W * H);
  const WarpCoord c = warp_coord(x, y, disp[i], H, W, r2l);
  const float fx0 = floorf(c.ix), fy0 = floorf(c.iy);
  const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
  const float wy1 = c.iy - fy0, wy0 = (fy0 + 1.f) - c.iy;
  const bool bx1 = x1 <= W - 1, by1 = y1 <= H - 1;
  const long plane = (long)H * W;
  float gix = 0.f;
  for (int ch = 0; ch < C; ++ch) {
    const float* p = img + ((long)b * C + ch) * plane;
    const float nw = p[(long)y0 * W + x0];
    const float ne = bx1 ? p[(long)y0 * W + x1] : 0.f;
    const float sw = by1 ? p[(long)y1 * W + x0] : 0.f;
    const float se = (bx1 && by1) ? p[(long)y1 * W + x1] : 0.f;
    const float go = g_warped[((long)b * C + ch) * plane + (long)y * W + x];
    gix += go * ((ne - nw) * wy0 + (se - sw) * wy1);
  }
  // d ix / d nx = W/2 (times the clip multiplier); d nx / d fx = 2/W; d fx / d disp = -/+ 1.
  const float g_nx = gix * (c.mx * ((float)W / 2.f));
  const float g_fx = g_nx * (2.f / (float)W);
  g_disp[i] = (r2l ? -g_fx : g_fx) + (add_src ? add_src[i] : 0.f);
}

// ---- host ------------------------------------------------------------------------------------
extern "C" int as_upsample_bilinear_fwd(const float* src, int B, int h, int w, float* dst, int H, int W,
                                        float gain, void* stream) {
  AS_CHECK_ARG(src && dst && B > 0 && h > 0 && w > 0 && H > 0 && W > 0, "as_upsample_bilinear_fwd: bad argument");
  const long n = (long)B * H * W;
  as_prof_mark(AS_PROF_UPSAMPLE_FWD, (hipStream_t)stream, 1, 0.0);
  hipLaunchKernelGGL(upsample_fwd_kernel, dim3(as_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, src, B, h, w,
                     dst, H, W, gain);
  as_prof_mark(AS_PROF_UPSAMPLE_FWD, (hipStream_t)stream, 0, 4.0 * ((double)B * h * w + (double)B * H * W));
  AS_CHECK_LAUNCH("as_upsample_bilinear_fwd");
  return AS_OK;
}

extern "C" int as_upsample_bilinear_bwd(const float* g_dst, int B, int H, int W, float* g_src, int h, int w,
                                        float gain, void* stream) {
  AS_CHECK_ARG(g_dst && g_src && B > 0 && h > 0 && w > 0 && H > 0 && W > 0, "as_upsample_bilinear_bwd: bad argument");
  const long n = (long)B * h * w;
  as_prof_mark(AS_PROF_UPSAMPLE_BWD, (hipStream_t)stream, 1, 0.0);
  hipLaunchKernelGGL(upsample_bwd_kernel, dim3(as_div_up(n, 4)), dim3(256), 0, (hipStream_t)stream, g_dst, B, H, W,
                     g_src, h, w, gain);
  as_prof_mark(AS_PROF_UPSAMPLE_BWD, (hipStream_t)stream, 0, 4.0 * ((double)B * h * w + (double)B * H * W));
  AS_CHECK_LAUNCH("as_upsample_bilinear_bwd");
  return AS_OK;
}

extern "C" int as_warp_fwd(const float* img, const float* disp, int B, int C, int H, int W, int right_to_left,
                           float* warped, uint8_t* mask, void* stream) {
  AS_CHECK_ARG(img && disp && warped && B > 0 && C > 0 && H > 0 && W > 0, "as_warp_fwd: bad argument");
  const long n = (long)B * H * W;
  as_prof_mark(AS_PROF_WARP_FWD, (hipStream_t)stream, 1, 0.0);
  hipLaunchKernelGGL(warp_fwd_kernel, dim3(as_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, img, disp, B, C, H,
                     W, right_to_left, warped, mask);
  as_prof_mark(AS_PROF_WARP_FWD, (hipStream_t)stream, 0, (double)B * H * W * (4.0 * (2 * C + 1) + 1.0));
  AS_CHECK_LAUNCH("as_warp_fwd");
  return AS_OK;
}

extern "C" int as_warp_bwd_add(const float* g_warped, const float* img, const float* disp, const float* add_src, int B, int C,
                               int H, int W, int right_to_left, float* g_disp, void* stream) {
  AS_CHECK_ARG(g_warped && img && disp && g_disp && B > 0 && C > 0 && H > 0 && W > 0, "as_warp_bwd: bad argument");
  const long n = (long)B * H * W;
  as_prof_mark(AS_PROF_WARP_BWD, (hipStream_t)stream, 1, 0.0);
  hipLaunchKernelGGL(warp_bwd_kernel, dim3(as_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, g_warped, img, disp,
                     add_src, B, C, H, W, right_to_left, g_disp);
  as_prof_mark(AS_PROF_WARP_BWD, (hipStream_t)stream, 0, (double)B * H * W * 4.0 * (2 * C + 2 + (add_src ? 1 : 0)));
  AS_CHECK_LAUNCH("as_warp_bwd");
  return AS_OK;
}

extern "C" int as_warp_bwd(const float* g_warped, const float* img, const float* disp, int B, int C, int H, int W,
                           int right_to_left, float* g_disp, void* stream) {
  return as_warp_bwd_add(g_warped, img, disp, nullptr, B, C, H, W, right_to_left, g_disp, stream);
}
