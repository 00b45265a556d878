// a10: monodepth photometric loss (SSIM + L1 + edge-aware smoothness), forward and
// hand-written backward, plus the masked mean of adapt.py:81-83.
// Reference semantics: adaptive_stereo/utils/loss_functions.py:41-72 (SSIM, 3x3
// avg_pool2d stride 1 pad 1, zero padding counted), :75-103 (smoothness), :106-138
// (0.85*SSIM + 0.15*L1 + w*smooth; disparity normalised by its per-image mean + 1e-7).
//
// Forward is one full-resolution stencil pass (plus a per-image mean reduction);
// backward is two passes: (A) per window-centre partial derivatives w.r.t. the pooled
// moments, (B) a 3x3 gather of those onto each warped pixel.  All HBM/L2-bound.
// fp contraction is disabled in this file so sigma = E[x^2] - mu^2 rounds the way the
// reference's separate ATen multiply / subtract do (the subtraction cancels ~2 digits).
// What bounds the three stencil kernels (62 / 86 / 51 us at 4 pairs for 60-190 MB) is arithmetic, not loads: ~25 IEEE
// divisions (by 9, by 3, by the mean disparity, n/d) and two expf per pixel, ~600 vector instructions.  Staging 32x8 tiles
// with a halo in LDS and taking the 54 / 81 window taps from there (same order, same bits; tried in round 2) changed
// nothing: 73 / 87 / 55 us on a box that runs the step 4 % slower.  Not kept.
#include "photometric_dev.h"
#pragma clang fp contract(off)

#define PH_BLOCKS_PER_IMAGE 512     // x B workgroups: eight per CU at 4 images

// ---- deterministic per-image reductions --------------------------------------------------
// partial[b][blk] (fp64) then a fixed-order finalize.
__global__ __launch_bounds__(256) void image_sum_kernel(const float* __restrict__ v, long n_per_image,
                                                         double* __restrict__ partial) {
  __shared__ double red[4];
  const int b = blockIdx.y;
  const float* p = v + (long)b * n_per_image;
  double s = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_per_image; i += (long)gridDim.x * 256) s += (double)p[i];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[(long)b * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// out[b] = sum_blk partial[b][blk] * mul: one wave per image, lanes stride over the blocks, fixed-order butterfly
__global__ __launch_bounds__(64) void image_sum_finalize_kernel(const double* __restrict__ partial, int nblk, int B,
                                                                double mul, float* __restrict__ out) {
  const int b = blockIdx.x;
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 64) s += partial[(long)b * nblk + i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) out[b] = (float)(s * mul);
}

// ---- shared per-pixel SSIM arithmetic (photometric_dev.h): the 3x3 sums here, the terms there ----------------------
__device__ inline SsimTerms ssim_at(const float* __restrict__ X, const float* __restrict__ Y, int y, int x, int H, int W) {
  float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
  if (y > 0 && y < H - 1 && x > 0 && x < W - 1) {
    // interior (all but the one-pixel frame): three row bases, taps at constant offsets, no predicates; same order
    const float* xr = X + (long)(y - 1) * W + (x - 1);
    const float* yr = Y + (long)(y - 1) * W + (x - 1);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const float a = xr[dx], b = yr[dx];
        sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
      }
      xr += W; yr += W;
    }
  } else {
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const int yy = y + dy;
      if (yy < 0 || yy >= H) continue;
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int xx = x + dx;
        if (xx < 0 || xx >= W) continue;
        const float a = X[(long)yy * W + xx], b = Y[(long)yy * W + xx];
        sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
      }
    }
  }
  return ssim_from_sums(sx, sy, sxx, syy, sxy);
}

// edge weights exp(-mean_c |I(p) - I(p+e)|)
__device__ inline float edge_wx(const float* __restrict__ img, long plane, int y, int x, int W) {
  const long o = (long)y * W + x;
  return edge_weight(img[o], img[o + 1], img[plane + o], img[plane + o + 1], img[2 * plane + o], img[2 * plane + o + 1]);
}
__device__ inline float edge_wy(const float* __restrict__ img, long plane, int y, int x, int W) {
  const long o = (long)y * W + x;
  return edge_weight(img[o], img[o + W], img[plane + o], img[plane + o + W], img[2 * plane + o], img[2 * plane + o + W]);
}

// ---- forward -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void monodepth_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ img,
                                                             const float* __restrict__ warped, const float* __restrict__ mean_disp,
                                                             int B, int H, int W, float sw,
                                                             float* __restrict__ total, float* __restrict__ l1,
                                                             float* __restrict__ ssim, float* __restrict__ smooth) {
  const long plane = (long)H * W;
  const int op = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;      // 32-bit index arithmetic: one image per grid row
  if (op >= H * W) return;
  const int y = op / W, x = op - y * W;
  const long i = (long)b * plane + op;
  const float* I = img + (long)b * 3 * plane;
  const float* Wp = warped + (long)b * 3 * plane;
  float s_acc = 0.f, l_acc = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const SsimTerms t = ssim_at(I + c * plane, Wp + c * plane, y, x, H, W);
    s_acc += fminf(fmaxf(t.raw, 0.f), 1.f);
    l_acc += fabsf(I[c * plane + (long)y * W + x] - Wp[c * plane + (long)y * W + x]);
  }
  const float ps = div3(s_acc), pl = div3(l_acc);
  const float den = mean_disp[b] + 1e-7f;
  const float* P = pred + (long)b * plane;
  const float nd = P[(long)y * W + x] / den;
  float sm = 0.f;
  if (x < W - 1) sm += fabsf(nd - P[(long)y * W + x + 1] / den) * edge_wx(I, plane, y, x, W);
  if (y < H - 1) sm += fabsf(nd - P[(long)(y + 1) * W + x] / den) * edge_wy(I, plane, y, x, W);
  const float photo = 0.85f * ps + 0.15f * pl;
  if (total) total[i] = photo + sw * sm;
  if (l1) l1[i] = pl;
  if (ssim) ssim[i] = ps;
  if (smooth) smooth[i] = sm;
}

// Where the gradient of the `total` loss map comes from: a dense map (autograd handed one over), or — the adaptation step's
// own case, loss = total[mask].mean() (adapt.py:81-83) — the validity mask and two device scalars: every valid pixel carries
// g_sum + g_mean / count (count = valid pixels, read from the forward's masked-sum result), every other pixel 0.  The second
// form needs no g_total plane and none of the element-wise launches that would build it.
struct GtSrc {
  const float* dense;        // [B][H][W] or null
  const uint8_t* mask;       // [B][H][W] or null
  const float* g_sum;        // [1] or null
  const float* g_mean;       // [1] or null
  const float* sum_count;    // [2]: masked sum, count (g_mean only)
};
__device__ inline float gt_scalar(const GtSrc& s) {
  if (s.mask == nullptr) return 0.f;
  float v = s.g_sum ? s.g_sum[0] : 0.f;
  if (s.g_mean) v += s.g_mean[0] / s.sum_count[1];
  return v;
}
__device__ inline float gt_at(const GtSrc& s, float gs, long i) {
  if (s.dense) return s.dense[i];
  return (s.mask && s.mask[i]) ? gs : 0.f;
}

// ---- backward pass A: per-centre SSIM coefficients, g_nd, and partial sums of g_nd*pred -------------
// coef layout: [B][10][H][W]: (a,b,c) x 3 channels, then g_nd.
__global__ __launch_bounds__(256) void monodepth_bwd_a_kernel(
    GtSrc g_total, const float* __restrict__ g_ssim, const float* __restrict__ g_smooth,
    const float* __restrict__ pred, const float* __restrict__ img, const float* __restrict__ warped,
    const float* __restrict__ mean_disp, int B, int H, int W, float sw,
    float* __restrict__ coef, double* __restrict__ partial) {
  __shared__ double red[4];
  const long plane = (long)H * W;
  const int b = blockIdx.y;
  const float* I = img + (long)b * 3 * plane;
  const float* Wp = warped + (long)b * 3 * plane;
  const float* P = pred + (long)b * plane;
  const float den = mean_disp[b] + 1e-7f;
  const float gs = gt_scalar(g_total);
  double s_local = 0.0;
  for (int o = blockIdx.x * 256 + threadIdx.x; o < H * W; o += gridDim.x * 256) {
    const int y = o / W, x = o - y * W;
    const long gi = (long)b * plane + o;
    const float gt = gt_at(g_total, gs, gi);
    const float G_ssim = 0.85f * gt + (g_ssim ? g_ssim[gi] : 0.f);
    float* cf = coef + (long)b * 10 * plane + o;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const SsimTerms t = ssim_at(I + c * plane, Wp + c * plane, y, x, H, W);
      const SsimCoef k = ssim_coef(t, G_ssim);
      cf[(3 * c + 0) * plane] = k.a;
      cf[(3 * c + 1) * plane] = k.b;
      cf[(3 * c + 2) * plane] = k.c;
    }
    // smoothness: derivative w.r.t. the normalised disparity at this pixel
    const float nd = P[o] / den;
    float g_nd = 0.f;
    if (x < W - 1) {
      const float Gs = sw * gt + (g_smooth ? g_smooth[gi] : 0.f);
      g_nd += Gs * edge_wx(I, plane, y, x, W) * sgn(nd - P[o + 1] / den);
    }
    if (x > 0) {
      const float gtl = gt_at(g_total, gs, gi - 1);
      const float Gs = sw * gtl + (g_smooth ? g_smooth[gi - 1] : 0.f);
      g_nd -= Gs * edge_wx(I, plane, y, x - 1, W) * sgn(P[o - 1] / den - nd);
    }
    if (y < H - 1) {
      const float Gs = sw * gt + (g_smooth ? g_smooth[gi] : 0.f);
      g_nd += Gs * edge_wy(I, plane, y, x, W) * sgn(nd - P[o + W] / den);
    }
    if (y > 0) {
      const float gtu = gt_at(g_total, gs, gi - W);
      const float Gs = sw * gtu + (g_smooth ? g_smooth[gi - W] : 0.f);
      g_nd -= Gs * edge_wy(I, plane, y - 1, x, W) * sgn(P[o - W] / den - nd);
    }
    cf[9 * plane] = g_nd;
    s_local += (double)g_nd * (double)P[o];
  }
  s_local = wave_sum_d(s_local);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s_local;
  __syncthreads();
  if (threadIdx.x == 0) partial[(long)b * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// ---- backward pass B: gather ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void monodepth_bwd_b_kernel(
    GtSrc g_total, const float* __restrict__ g_l1,
    const float* __restrict__ pred, const float* __restrict__ img, const float* __restrict__ warped,
    const float* __restrict__ mean_disp, const float* __restrict__ coef, const float* __restrict__ sum_gnd_pred,
    int B, int H, int W, float* __restrict__ g_pred, float* __restrict__ g_warped) {
  const long plane = (long)H * W;
  const int op = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (op >= H * W) return;
  const int y = op / W, x = op - y * W;
  const long o = op;
  const long i = (long)b * plane + op;
  const float* I = img + (long)b * 3 * plane;
  const float* Wp = warped + (long)b * 3 * plane;
  const float* cf = coef + (long)b * 10 * plane;
  const float gt = gt_at(g_total, gt_scalar(g_total), i);
  const float G_l1 = 0.15f * gt + (g_l1 ? g_l1[i] : 0.f);
  if (g_warped) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float xv = I[c * plane + o], yv = Wp[c * plane + o];
      float sa = 0.f, sb = 0.f, sc = 0.f;
      if (y > 0 && y < H - 1 && x > 0 && x < W - 1) {
        const float* ca = cf + (3 * c + 0) * plane + o - W - 1;
        const float* cb = cf + (3 * c + 1) * plane + o - W - 1;
        const float* cc = cf + (3 * c + 2) * plane + o - W - 1;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) { sa += ca[dx]; sb += cb[dx]; sc += cc[dx]; }
          ca += W; cb += W; cc += W;
        }
      } else {
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
          const int yy = y + dy;
          if (yy < 0 || yy >= H) continue;
#pragma unroll
          for (int dx = -1; dx <= 1; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= W) continue;
            const long q = (long)yy * W + xx;
            sa += cf[(3 * c + 0) * plane + q];
            sb += cf[(3 * c + 1) * plane + q];
            sc += cf[(3 * c + 2) * plane + q];
          }
        }
      }
      float g = div9(sa + 2.f * sb * yv + sc * xv);
      g += div3(G_l1) * (-sgn(xv - yv));
      g_warped[((long)b * 3 + c) * plane + o] = g;
    }
  }
  if (g_pred) {
    const float den = mean_disp[b] + 1e-7f;
    const float r = 1.f / den;
    // d/d pred of pred/(mean+eps): direct term plus the term through the mean
    g_pred[i] = cf[9 * plane + o] * r - (sum_gnd_pred[b] * r * r) / (float)plane;
  }
}

// ---- masked sum ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void masked_sum_kernel(const float* __restrict__ v, const uint8_t* __restrict__ m,
                                                          long n, double* __restrict__ partial) {
  __shared__ double red[2][4];
  double s = 0.0, c = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    if (m[i]) { s += (double)v[i]; c += 1.0; }
  }
  s = wave_sum_d(s); c = wave_sum_d(c);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    partial[2 * blockIdx.x + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}
__global__ void masked_sum_finalize_kernel(const double* __restrict__ partial, int nblk, float* __restrict__ out2, int with_mean) {
  double s = 0.0, c = 0.0;                       // one wave; lane-strided, then a shuffle tree: fixed order
  for (int i = threadIdx.x; i < nblk; i += 64) { s += partial[2 * i]; c += partial[2 * i + 1]; }
  s = wave_sum_d(s); c = wave_sum_d(c);
  if (threadIdx.x == 0) {
    out2[0] = (float)s; out2[1] = (float)c;
    if (with_mean) { out2[2] = (float)s / (float)c; out2[3] = (float)c; }     // fp32 sum / fp32 count, as the caller's division was
  }
}

// ---- host ------------------------------------------------------------------------------------------------------
extern "C" int64_t as_monodepth_workspace(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return -1;
  // [mean B | sum B | pad to 16] + fp64 partials [B][PH_BLOCKS] + coef [B][10][H][W]
  return 64 + 2 * (int64_t)((B + 15) / 16 * 16) + 2 * (int64_t)B * PH_BLOCKS_PER_IMAGE + (int64_t)B * 10 * H * W;
}

struct PhWs { float* mean; float* sum; double* partial; float* coef; };
static PhWs carve(float* ws, int B) {
  PhWs w;
  const int Bp = (B + 15) / 16 * 16;
  w.mean = ws; w.sum = ws + Bp;
  w.partial = reinterpret_cast<double*>(ws + 2 * Bp);
  w.coef = ws + 2 * Bp + 2 * (int64_t)B * PH_BLOCKS_PER_IMAGE;
  return w;
}

// per-image mean of pred into mean[B] (partial: B x PH_BLOCKS_PER_IMAGE doubles); photometric_rows.hip uses it too
int as_photometric_image_mean(const float* pred, int B, long plane, double* partial, float* mean, hipStream_t st) {
  hipLaunchKernelGGL(image_sum_kernel, dim3(PH_BLOCKS_PER_IMAGE, B), dim3(256), 0, st, pred, plane, partial);
  AS_CHECK_LAUNCH("monodepth(mean)");
  hipLaunchKernelGGL(image_sum_finalize_kernel, dim3(B), dim3(64), 0, st, partial, PH_BLOCKS_PER_IMAGE, B, 1.0 / (double)plane, mean);
  AS_CHECK_LAUNCH("monodepth(mean finalize)");
  return AS_OK;
}
static int image_mean(const float* pred, int B, long plane, PhWs& w, hipStream_t st) {
  return as_photometric_image_mean(pred, B, plane, w.partial, w.mean, st);
}

extern "C" int as_monodepth_loss_fwd(const float* pred, const float* img, const float* warped, int B, int H, int W,
                                     float smoothness_weight, float* total, float* l1, float* ssim, float* smooth,
                                     float* workspace, void* stream) {
  AS_CHECK_ARG(pred && img && warped && workspace && B > 0 && H > 1 && W > 1 && B <= 65535 && (long)H * W < (1L << 31),
               "as_monodepth_loss_fwd: bad argument");
  AS_CHECK_ARG(((uintptr_t)workspace & 15) == 0, "as_monodepth_loss_fwd: workspace must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  PhWs w = carve(workspace, B);
  const long plane = (long)H * W;
  as_prof_mark(AS_PROF_LOSS_FWD, st, 1, 0.0);
  if (int e = image_mean(pred, B, plane, w, st)) return e;
  hipLaunchKernelGGL(monodepth_fwd_kernel, dim3(as_div_up(plane, 256), B), dim3(256), 0, st, pred, img, warped,
                     w.mean, B, H, W, smoothness_weight, total, l1, ssim, smooth);
  as_prof_mark(AS_PROF_LOSS_FWD, st, 0, 4.0 * (double)B * plane * (1 + 3 + 3 + 4));     // pred, img, warped read; four maps written
  AS_CHECK_LAUNCH("as_monodepth_loss_fwd");
  return AS_OK;
}

static int monodepth_bwd(GtSrc g_total, const float* g_l1, const float* g_ssim, const float* g_smooth,
                         const float* pred, const float* img, const float* warped, int B, int H, int W,
                         float smoothness_weight, float* g_pred, float* g_warped, float* workspace, void* stream,
                         const float* mean_fwd = nullptr);

extern "C" int as_monodepth_loss_bwd(const float* g_total, const float* g_l1, const float* g_ssim, const float* g_smooth,
                                     const float* pred, const float* img, const float* warped, int B, int H, int W,
                                     float smoothness_weight, float* g_pred, float* g_warped,
                                     float* workspace, void* stream) {
  GtSrc src = {g_total, nullptr, nullptr, nullptr, nullptr};
  return monodepth_bwd(src, g_l1, g_ssim, g_smooth, pred, img, warped, B, H, W, smoothness_weight, g_pred, g_warped, workspace,
                       stream);
}

extern "C" int as_monodepth_loss_bwd_masked(const uint8_t* mask, const float* g_sum, const float* g_mean,
                                            const float* sum_count, const float* pred, const float* img, const float* warped,
                                            int B, int H, int W, float smoothness_weight, float* g_pred, float* g_warped,
                                            float* workspace, const float* fwd_workspace, void* stream) {
  AS_CHECK_ARG(mask && (g_sum || g_mean) && (g_mean == nullptr || sum_count != nullptr),
               "as_monodepth_loss_bwd_masked: mask, at least one of g_sum / g_mean, and sum_count with g_mean");
  GtSrc src = {nullptr, mask, g_sum, g_mean, sum_count};
  // fwd_workspace (may be NULL): the workspace as_monodepth_loss_fwd ran with on the SAME pred — its per-image mean disparity
  // (the first B floats) is reused instead of being summed again (two launches)
  return monodepth_bwd(src, nullptr, nullptr, nullptr, pred, img, warped, B, H, W, smoothness_weight, g_pred, g_warped, workspace,
                       stream, fwd_workspace);
}

static int monodepth_bwd(GtSrc g_total, const float* g_l1, const float* g_ssim, const float* g_smooth,
                         const float* pred, const float* img, const float* warped, int B, int H, int W,
                         float smoothness_weight, float* g_pred, float* g_warped, float* workspace, void* stream,
                         const float* mean_fwd) {
  AS_CHECK_ARG(pred && img && warped && workspace && B > 0 && H > 1 && W > 1 && B <= 65535 && (long)H * W < (1L << 31),
               "as_monodepth_loss_bwd: bad argument");
  AS_CHECK_ARG(((uintptr_t)workspace & 15) == 0, "as_monodepth_loss_bwd: workspace must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  PhWs w = carve(workspace, B);
  const long plane = (long)H * W;
  as_prof_mark(AS_PROF_LOSS_BWD, st, 1, 0.0);
  if (mean_fwd != nullptr) w.mean = const_cast<float*>(mean_fwd);      // read-only from here on
  else if (int e = image_mean(pred, B, plane, w, st)) return e;
  hipLaunchKernelGGL(monodepth_bwd_a_kernel, dim3(PH_BLOCKS_PER_IMAGE, B), dim3(256), 0, st, g_total, g_ssim, g_smooth,
                     pred, img, warped, w.mean, B, H, W, smoothness_weight, w.coef, w.partial);
  AS_CHECK_LAUNCH("as_monodepth_loss_bwd(A)");
  hipLaunchKernelGGL(image_sum_finalize_kernel, dim3(B), dim3(64), 0, st, w.partial,
                     PH_BLOCKS_PER_IMAGE, B, 1.0, w.sum);
  AS_CHECK_LAUNCH("as_monodepth_loss_bwd(sum)");
  hipLaunchKernelGGL(monodepth_bwd_b_kernel, dim3(as_div_up(plane, 256), B), dim3(256), 0, st, g_total, g_l1,
                     pred, img, warped, w.mean, w.coef, w.sum, B, H, W, g_pred, g_warped);
  // four incoming gradient maps, pred, img, warped read; g_pred and g_warped written (the 10-plane coefficient
  // workspace between the two passes is the implementation's, not the algorithm's)
  as_prof_mark(AS_PROF_LOSS_BWD, st, 0, 4.0 * (double)B * plane * (4 + 1 + 3 + 3 + 1 + 3));
  AS_CHECK_LAUNCH("as_monodepth_loss_bwd(B)");
  return AS_OK;
}

// ---- evaluation reductions: EPE and D1-all at 2/3/4/5 px (reference train.py:98-106) ------------------------------
// out6 = [sum |pred-gt| over gt>0, count(gt>0), count(gt>0 & err>2), ... >3, >4, >5]
__global__ __launch_bounds__(256) void eval_metrics_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                            long n, double* __restrict__ partial) {
  __shared__ double red[6][4];
  double s[6] = {0, 0, 0, 0, 0, 0};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float g = gt[i];
    if (g > 0.f) {
      const float e = fabsf(pred[i] - g);
      s[0] += (double)e; s[1] += 1.0;
      s[2] += e > 2.f ? 1.0 : 0.0; s[3] += e > 3.f ? 1.0 : 0.0;
      s[4] += e > 4.f ? 1.0 : 0.0; s[5] += e > 5.f ? 1.0 : 0.0;
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    s[k] = wave_sum_d(s[k]);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = s[k];
  }
  __syncthreads();
  if (threadIdx.x < 6) partial[6 * blockIdx.x + threadIdx.x] =
      red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}
__global__ void eval_metrics_finalize_kernel(const double* __restrict__ partial, int nblk, float* __restrict__ out6) {
  const int k = threadIdx.x >> 6 ? -1 : 0;     // one wave
  (void)k;
  for (int m = 0; m < 6; ++m) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 64) s += partial[6 * i + m];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) out6[m] = (float)s;
  }
}

#define EM_BLOCKS 256
extern "C" int64_t as_eval_metrics_workspace(int64_t n) { return n > 0 ? 12 * EM_BLOCKS : -1; }

extern "C" int as_eval_metrics(const float* pred, const float* gt, int64_t n, float* out6, float* workspace, void* stream) {
  AS_CHECK_ARG(pred && gt && out6 && workspace && n > 0, "as_eval_metrics: bad argument");
  AS_CHECK_ARG(((uintptr_t)workspace & 7) == 0, "as_eval_metrics: workspace must be 8-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  long nb = (n + 255) / 256;
  if (nb > EM_BLOCKS) nb = EM_BLOCKS;
  double* partial = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(eval_metrics_kernel, dim3((int)nb), dim3(256), 0, st, pred, gt, (long)n, partial);
  AS_CHECK_LAUNCH("as_eval_metrics");
  hipLaunchKernelGGL(eval_metrics_finalize_kernel, dim3(1), dim3(64), 0, st, partial, (int)nb, out6);
  AS_CHECK_LAUNCH("as_eval_metrics(finalize)");
  return AS_OK;
}

#define MS_BLOCKS 512
extern "C" int64_t as_masked_sum_workspace(int64_t n) { return n > 0 ? 4 * MS_BLOCKS : -1; }

static int masked_sum_impl(const float* v, const uint8_t* mask, int64_t n, float* out2, float* workspace, void* stream, int with_mean);
extern "C" int as_masked_sum(const float* v, const uint8_t* mask, int64_t n, float* out2, float* workspace, void* stream) {
  return masked_sum_impl(v, mask, n, out2, workspace, stream, 0);
}
extern "C" int as_masked_sum_mean(const float* v, const uint8_t* mask, int64_t n, float* out4, float* workspace, void* stream) {
  return masked_sum_impl(v, mask, n, out4, workspace, stream, 1);
}
static int masked_sum_impl(const float* v, const uint8_t* mask, int64_t n, float* out2, float* workspace, void* stream, int with_mean) {
  AS_CHECK_ARG(v && mask && out2 && workspace && n > 0, "as_masked_sum: bad argument");
  AS_CHECK_ARG(((uintptr_t)workspace & 7) == 0, "as_masked_sum: workspace must be 8-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  long nb = (n + 255) / 256;
  if (nb > MS_BLOCKS) nb = MS_BLOCKS;
  double* partial = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(masked_sum_kernel, dim3((int)nb), dim3(256), 0, st, v, mask, (long)n, partial);
  AS_CHECK_LAUNCH("as_masked_sum");
  hipLaunchKernelGGL(masked_sum_finalize_kernel, dim3(1), dim3(64), 0, st, partial, (int)nb, out2, with_mean);
  AS_CHECK_LAUNCH("as_masked_sum(finalize)");
  return AS_OK;
}

// ---- a13: khamis_robust_loss (reference utils/loss_functions.py:6-15; ER modes, adapt.py:339-349) ------------------
// loss = sum_{gt>0} (sqrt((gt-pred)^2 + 4)/2 - 1) / max(count(gt>0), 1).  One streaming pass, per-element value in fp32
// exactly as the reference's element-wise chain forms it, fixed-order fp64 two-stage sum (deterministic).
__global__ __launch_bounds__(256) void khamis_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                          long n, double* __restrict__ partial) {
  __shared__ double red[2][4];
  double s = 0.0, c = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float g = gt[i];
    if (g > 0.f) {
      const float d = g - pred[i];
      const float v = __fsub_rn(__fdiv_rn(__fsqrt_rn(__fadd_rn(__fmul_rn(d, d), 4.f)), 2.f), 1.f);
      s += (double)v; c += 1.0;
    }
  }
  s = wave_sum_d(s); c = wave_sum_d(c);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    partial[2 * blockIdx.x + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}
__global__ void khamis_finalize_kernel(const double* __restrict__ partial, int nblk, float* __restrict__ out2) {
  double s = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 64) { s += partial[2 * i]; c += partial[2 * i + 1]; }
  s = wave_sum_d(s); c = wave_sum_d(c);
  if (threadIdx.x == 0) {
    const double nv = c > 1.0 ? c : 1.0;
    out2[0] = (float)s / (float)nv;      // the reference divides the fp32 sum by the count, in fp32
    out2[1] = (float)nv;
  }
}
// g_pred = g_loss * d loss / d pred = -g_loss/count * (gt-pred) / (2 sqrt((gt-pred)^2+4)) where gt>0, else 0
__global__ __launch_bounds__(256) void khamis_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                          const float* __restrict__ g_loss,
                                                          const float* __restrict__ out2, long n,
                                                          float* __restrict__ g_pred) {
  const float scale = g_loss[0] / out2[1];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float g = gt[i];
    float r = 0.f;
    if (g > 0.f) {
      const float d = g - pred[i];
      r = -scale * d / (2.f * sqrtf(d * d + 4.f));
    }
    g_pred[i] = r;
  }
}

extern "C" int64_t as_khamis_workspace(int64_t n) { return n > 0 ? 4 * MS_BLOCKS : -1; }

extern "C" int as_khamis_fwd(const float* pred, const float* gt, int64_t n, float* out2, float* workspace, void* stream) {
  AS_CHECK_ARG(pred && gt && out2 && workspace && n > 0, "as_khamis_fwd: bad argument");
  AS_CHECK_ARG(((uintptr_t)workspace & 7) == 0, "as_khamis_fwd: workspace must be 8-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  long nb = (n + 255) / 256;
  if (nb > MS_BLOCKS) nb = MS_BLOCKS;
  double* partial = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(khamis_fwd_kernel, dim3((int)nb), dim3(256), 0, st, pred, gt, (long)n, partial);
  AS_CHECK_LAUNCH("as_khamis_fwd");
  hipLaunchKernelGGL(khamis_finalize_kernel, dim3(1), dim3(64), 0, st, partial, (int)nb, out2);
  AS_CHECK_LAUNCH("as_khamis_fwd(finalize)");
  return AS_OK;
}

extern "C" int as_khamis_bwd(const float* pred, const float* gt, const float* g_loss, const float* out2, int64_t n,
                             float* g_pred, void* stream) {
  AS_CHECK_ARG(pred && gt && g_loss && out2 && g_pred && n > 0, "as_khamis_bwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  long nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(khamis_bwd_kernel, dim3((int)nb), dim3(256), 0, st, pred, gt, g_loss, out2, (long)n, g_pred);
  AS_CHECK_LAUNCH("as_khamis_bwd");
  return AS_OK;
}
