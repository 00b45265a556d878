// a6 / a7-head: bilinear up-sampling (align_corners=False), and a9: LinearWarping.
// Reference semantics:
//   F.interpolate(x, size, mode="bilinear", align_corners=False) * gain     stereo_net.py:106-114, 201-202
//   LinearWarping.forward: grid (x -/+ d, y), normalised 2x/w-1, 2y/h-1, then
//   F.grid_sample(bilinear, padding_mode="border", align_corners=False) and the
//   validity mask -1 <= g <= 1                                              models/linear_warping.py:18-57
// Both are HBM-bound gathers; one output pixel per lane, W-coalesced.  Arithmetic
// follows ATen's order of operations (source index = scale*(dst+0.5)-0.5 clamped at 0;
// grid un-normalisation ((g+1)*size-1)/2; border clip with zero gradient at the clip).
#include "photometric_dev.h"

__device__ inline void bilin_src(float scale, int dst, int in_size, int& i0, int& i1, float& l0, float& l1) {
  float r = scale * ((float)dst + 0.5f) - 0.5f;
  r = r < 0.f ? 0.f : r;
  i0 = (int)r;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = r - (float)i0;
  l0 = 1.f - l1;
}

// UP_ROWS fine rows per workgroup (no per-pixel division), four adjacent pixels per thread and step, one 16-byte store each:
// the kernel is a 4-byte-per-pixel write stream.  The sixteen taps of a step come from the two source rows staged in LDS
// (LDS template flavour, source rows of up to 1024 pixels): read from global memory, 16 L1 loads per 16-byte store put the
// texture-address path, not the stores, in charge (22 us for 59.6 MB at 32 pairs).
#define UP_ROWS 4
template <bool LDS>
__global__ __launch_bounds__(1024) void upsample_fwd_kernel(const float* __restrict__ src, int B, int h, int w,
                                                             float* __restrict__ dst, int H, int W, float gain) {
  __shared__ float srow[2][LDS ? 1024 : 1];
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  const int rows = B * H;
  int staged0 = -1, staged1 = -1;                                            // source rows (b * h + y) now in srow[0], srow[1]
  for (int row = blockIdx.x * UP_ROWS; row < min(rows, (int)(blockIdx.x + 1) * UP_ROWS); ++row) {      // row = b * H + Y
    const int b = row / H, Y = row - b * H;
    int y0, y1; float ly0, ly1;
    bilin_src(sh, Y, h, y0, y1, ly0, ly1);
    const float* s0 = src + ((long)b * h + y0) * w;
    const float* s1 = src + ((long)b * h + y1) * w;
    if (LDS && (staged0 != b * h + y0 || staged1 != b * h + y1)) {         // (workgroup-uniform)
      __syncthreads();
      for (int k = threadIdx.x; k < w; k += blockDim.x) { srow[0][k] = s0[k]; srow[1][k] = s1[k]; }
      __syncthreads();
      staged0 = b * h + y0; staged1 = b * h + y1;
    }
    float* d = dst + (long)row * W;
    const int mis = (int)(((uintptr_t)d >> 2) & 3);       // the row's first pixel relative to a 16-byte boundary
    // pieces start at X = 4 * q - mis: aligned stores for every row (the first piece of a misaligned row starts before it)
    for (int X0 = 4 * (int)threadIdx.x - mis; X0 < W; X0 += 4 * (int)blockDim.x) {
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int X = min(max(X0 + k, 0), W - 1);
        int x0, x1; float lx0, lx1;
        bilin_src(sw, X, w, x0, x1, lx0, lx1);
        const float top = LDS ? lx0 * srow[0][x0] + lx1 * srow[0][x1] : lx0 * s0[x0] + lx1 * s0[x1];
        const float bot = LDS ? lx0 * srow[1][x0] + lx1 * srow[1][x1] : lx0 * s1[x0] + lx1 * s1[x1];
        v[k] = (ly0 * top + ly1 * bot) * gain;
      }
      if (X0 >= 0 && X0 + 3 < W) {
        f32x4 o; o.x = v[0]; o.y = v[1]; o.z = v[2]; o.w = v[3];
        *reinterpret_cast<f32x4*>(d + X0) = o;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (X0 + k >= 0 && X0 + k < W) d[X0 + k] = v[k];
      }
    }
  }
}

// fine rows / columns whose two source taps can include coarse index i
__device__ inline void footprint(float scale, int i, int fine, int& lo, int& hi) {
  lo = (int)floorf(((float)i - 1.f + 0.5f) / scale - 0.5f) - 1;
  hi = (int)ceilf(((float)i + 1.f + 0.5f) / scale - 0.5f) + 1;
  lo = max(lo, 0); hi = min(hi, fine - 1);
}
// weight of coarse index i in fine index D's interpolation
__device__ inline float tap_weight(float scale, int D, int in_size, int i) {
  int i0, i1; float l0, l1;
  bilin_src(scale, D, in_size, i0, i1, l0, l1);
  return (i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f);
}

// Adjoint, separable and in gather form (deterministic).  One workgroup per (image, coarse row i, chunk of coarse columns):
//   pass 1  colsum[X] = sum_Y wy(Y, i) * g[Y][X]   one thread per fine column of the chunk's footprint, rows top to bottom:
//                                                  coalesced row reads, each fine pixel read by the two coarse rows it feeds
//   pass 2  g_src[i][j] = gain * sum_X wx(X, j) * colsum[X]   one wave per coarse column, lanes over its ~2/scale fine columns
// (the first generation spent a whole wave on the ~1,150 fine pixels of ONE coarse pixel and recomputed both interpolations
// for every one of them: 21.6 us at 4 pairs of 375x1242 for 7.4 MB)
#define UPB_SPAN 1024
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ g_dst, int B, int H, int W,
                                                            float* __restrict__ g_src, int h, int w, float gain, int chunk) {
  __shared__ float colsum[UPB_SPAN];
  const int nchunks = (w + chunk - 1) / chunk;
  const int c = blockIdx.x % nchunks;
  const int i = (blockIdx.x / nchunks) % h, b = blockIdx.x / (nchunks * h);
  const int j0 = c * chunk, j1 = min(j0 + chunk, w);
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  int Y0, Y1, X0, X1, t0, t1;
  footprint(sh, i, H, Y0, Y1);
  footprint(sw, j0, W, X0, t1);
  footprint(sw, j1 - 1, W, t0, X1);
  const float* g = g_dst + (long)b * H * W;
  // the row weights are the same for every column: once per workgroup (up to 256 footprint rows), not once per thread and row
  __shared__ float wy_tab[256];
  const int ny = Y1 - Y0 + 1;
  const bool tab = ny <= 256;
  if (tab) wy_tab[threadIdx.x] = (int)threadIdx.x < ny ? tap_weight(sh, Y0 + threadIdx.x, h, i) : 0.f;     // (zeros behind the footprint)
  __syncthreads();
  for (int X = X0 + threadIdx.x; X <= X1; X += 256) {
    float acc = 0.f;
    if (tab) {
      // eight rows requested together (rows behind the footprint: the last row again, weight zero) — one load latency per
      // eight rows instead of one per row; the sum stays in row order
      for (int Yb = Y0; Yb <= Y1; Yb += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = g[(long)min(Yb + u, Y1) * W + X];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += wy_tab[min(Yb + u - Y0, 255)] * v[u];
      }
    } else {
      for (int Y = Y0; Y <= Y1; ++Y) acc += tap_weight(sh, Y, h, i) * g[(long)Y * W + X];
    }
    colsum[X - X0] = acc;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  for (int j = j0 + (threadIdx.x >> 6); j < j1; j += 4) {
    int a0, a1;
    footprint(sw, j, W, a0, a1);
    float acc = 0.f;
    for (int X = a0 + lane; X <= a1; X += 64) acc += tap_weight(sw, X, w, j) * colsum[X - X0];
    acc = wave_sum(acc);
    if (lane == 0) g_src[((long)b * h + i) * w + j] = acc * gain;
  }
}

// ---- LinearWarping (sample geometry and weights: photometric_dev.h, shared with photometric_rows.hip) ---------------
__global__ __launch_bounds__(256) void warp_fwd_kernel(const float* __restrict__ img, const float* __restrict__ disp,
                                                        int B, int C, int H, int W, int r2l,
                                                        float* __restrict__ warped, uint8_t* __restrict__ mask) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * H * W) return;
  const int x = i % W, y = (i / W) % H, b = i / ((long)W * H);
  const WarpGeom g = warp_geom(x, y, disp[i], H, W, r2l);
  const long plane = (long)H * W;
  for (int ch = 0; ch < C; ++ch) {
    const float* p = img + ((long)b * C + ch) * plane;
    const float nw = p[(long)g.y0 * W + g.x0];      // x0, y0 are always in range after the clip
    const float ne = g.bx1 ? p[(long)g.y0 * W + g.x0 + 1] : 0.f;
    const float sw = g.by1 ? p[(long)(g.y0 + 1) * W + g.x0] : 0.f;
    const float se = (g.bx1 && g.by1) ? p[(long)(g.y0 + 1) * W + g.x0 + 1] : 0.f;
    warped[((long)b * C + ch) * plane + (long)y * W + x] = warp_interp(nw, ne, sw, se, g);
  }
  if (mask) mask[i] = (uint8_t)g.valid;
}

__global__ __launch_bounds__(256) void warp_bwd_kernel(const float* __restrict__ g_warped, const float* __restrict__ img,
                                                        const float* __restrict__ disp, const float* __restrict__ add_src,
                                                        int B, int C, int H, int W, int r2l, float* __restrict__ g_disp) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * H * W) return;
  const int x = i % W, y = (i / W) % H, b = i / ((long)W * H);
  const WarpGeom g = warp_geom(x, y, disp[i], H, W, r2l);
  const long plane = (long)H * W;
  float gix = 0.f;
  for (int ch = 0; ch < C; ++ch) {
    const float* p = img + ((long)b * C + ch) * plane;
    const float nw = p[(long)g.y0 * W + g.x0];
    const float ne = g.bx1 ? p[(long)g.y0 * W + g.x0 + 1] : 0.f;
    const float sw = g.by1 ? p[(long)(g.y0 + 1) * W + g.x0] : 0.f;
    const float se = (g.bx1 && g.by1) ? p[(long)(g.y0 + 1) * W + g.x0 + 1] : 0.f;
    const float go = g_warped[((long)b * C + ch) * plane + (long)y * W + x];
    gix = __builtin_fmaf(go, warp_dix(nw, ne, sw, se, g), gix);
  }
  g_disp[i] = warp_gdisp(gix, g, W, r2l) + (add_src ? add_src[i] : 0.f);
}

// grid_sample(mode="nearest", padding_mode="border", align_corners=False): the tap nearest to the clipped sample position,
// ties to even (std::nearbyint, as ATen).  No gradient reaches the disparity (a piecewise-constant function of the grid).
__global__ __launch_bounds__(256) void warp_nearest_kernel(const float* __restrict__ img, const float* __restrict__ disp,
                                                            int B, int C, int H, int W, int r2l,
                                                            float* __restrict__ warped, uint8_t* __restrict__ mask) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * H * W) return;
  const int x = i % W, y = (i / W) % H, b = i / ((long)W * H);
  const WarpGeom g = warp_geom(x, y, disp[i], H, W, r2l);
  const int xn = min((int)rintf((float)g.x0 + g.wx1), W - 1), yn = min((int)rintf((float)g.y0 + g.wy1), H - 1);
  const long plane = (long)H * W;
  for (int ch = 0; ch < C; ++ch)
    warped[((long)b * C + ch) * plane + (long)y * W + x] = img[((long)b * C + ch) * plane + (long)yn * W + xn];
  if (mask) mask[i] = (uint8_t)g.valid;
}

// ---- host ------------------------------------------------------------------------------------
extern "C" int as_upsample_bilinear_fwd(const float* src, int B, int h, int w, float* dst, int H, int W,
                                        float gain, void* stream) {
  AS_CHECK_ARG(src && dst && B > 0 && h > 0 && w > 0 && H > 0 && W > 0, "as_upsample_bilinear_fwd: bad argument");
  as_prof_mark(AS_PROF_UPSAMPLE_FWD, (hipStream_t)stream, 1, 0.0);
  AS_CHECK_ARG((long)B * H < (1L << 31), "as_upsample_bilinear_fwd: too many rows");
  const long nblk = ((long)B * H + UP_ROWS - 1) / UP_ROWS;
  // whole waves covering one row in one pass where it fits (1242 pixels: 5 waves), at most 1024 threads
  int threads = ((W + 3) / 4 + 1 + 63) / 64 * 64;
  if (threads > 1024) threads = 1024;
  if (w <= 1024) hipLaunchKernelGGL(upsample_fwd_kernel<true>, dim3((unsigned)nblk), dim3(threads), 0, (hipStream_t)stream, src, B, h,
                                    w, dst, H, W, gain);
  else hipLaunchKernelGGL(upsample_fwd_kernel<false>, dim3((unsigned)nblk), dim3(threads), 0, (hipStream_t)stream, src, B, h, w,
                          dst, H, W, gain);
  as_prof_mark(AS_PROF_UPSAMPLE_FWD, (hipStream_t)stream, 0, 4.0 * ((double)B * h * w + (double)B * H * W));
  AS_CHECK_LAUNCH("as_upsample_bilinear_fwd");
  return AS_OK;
}

extern "C" int as_upsample_bilinear_bwd(const float* g_dst, int B, int H, int W, float* g_src, int h, int w,
                                        float gain, void* stream) {
  AS_CHECK_ARG(g_dst && g_src && B > 0 && h > 0 && w > 0 && H > 0 && W > 0, "as_upsample_bilinear_bwd: bad argument");
  as_prof_mark(AS_PROF_UPSAMPLE_BWD, (hipStream_t)stream, 1, 0.0);
  // coarse columns per workgroup: as many as keep the chunk's fine footprint ((chunk + 2) / scale + 4 columns) within a
  // 256-thread row and inside the LDS row of column sums
  const double inv = (double)W / (double)w;
  int chunk = (int)((252.0 / inv)) - 2;
  if (chunk < 1) chunk = 1;
  if (chunk > w) chunk = w;
  AS_CHECK_ARG((chunk + 2) * inv + 6.0 <= (double)UPB_SPAN, "as_upsample_bilinear_bwd: scale factor beyond %d fine columns per coarse column", UPB_SPAN / 3);
  const long nblk = (long)B * h * ((w + chunk - 1) / chunk);
  AS_CHECK_ARG(nblk < (1L << 31), "as_upsample_bilinear_bwd: too many workgroups");
  hipLaunchKernelGGL(upsample_bwd_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, g_dst, B, H, W,
                     g_src, h, w, gain, chunk);
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

extern "C" int as_warp_nearest_fwd(const float* img, const float* disp, int B, int C, int H, int W, int right_to_left,
                                   float* warped, uint8_t* mask, void* stream) {
  AS_CHECK_ARG(img && disp && warped && B > 0 && C > 0 && H > 0 && W > 0, "as_warp_nearest_fwd: bad argument");
  const long n = (long)B * H * W;
  hipLaunchKernelGGL(warp_nearest_kernel, dim3(as_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, img, disp, B, C, H,
                     W, right_to_left, warped, mask);
  AS_CHECK_LAUNCH("as_warp_nearest_fwd");
  return AS_OK;
}
