// Per-pixel arithmetic shared by the loss-tail kernels (a9 LinearWarping, a10 monodepth loss): photometric.hip / resample.hip
// (one thread per pixel, the dense-gradient API) and photometric_rows.hip (row-walking strips, the adaptation step's chain).
// Every function fixes its own floating-point contraction — products and sums round separately unless an fmaf is written
// out — so the two generations give the same bits whatever the including file's default is.
// Reference semantics: utils/loss_functions.py:41-72 (SSIM), :75-103 (smoothness), models/linear_warping.py:18-57.
#pragma once
#include "as_common.h"

// ---- division by 9 and by 3 ---------------------------------------------------------------------
// The kernels divide ~19 times per pixel by these two constants (the reference's avg_pool2d and means divide, so must we,
// bit for bit) and an IEEE fp32 division is ~12 instructions.  q = x*c, r = fma(-q, y, x), q' = fma(r, c, q) with
// c = RN(1/y) is the correctly rounded x / y for EVERY finite binary32 x for y = 9 and y = 3, -0 excepted (it comes out
// as +0): checked exhaustively over all 2^32 bit patterns (tests/tools/div_const.c).  Three instructions.
__device__ inline float div_const(float x, float y, float c) {
#pragma clang fp contract(off)
  const float q = x * c;
  const float r = __builtin_fmaf(-q, y, x);
  return __builtin_fmaf(r, c, q);
}
__device__ inline float div9(float x) { return div_const(x, 9.f, 1.f / 9.f); }
__device__ inline float div3(float x) { return div_const(x, 3.f, 1.f / 3.f); }

// ---- SSIM at one window centre, from the five 3x3 sums (zero padding counted: always / 9) ----------
struct SsimTerms { float mux, muy, A1, A2, B1, B2, n, d, raw; };

__device__ inline SsimTerms ssim_from_sums(float sx, float sy, float sxx, float syy, float sxy) {
#pragma clang fp contract(off)
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  SsimTerms t;
  t.mux = div9(sx); t.muy = div9(sy);
  const float sigx = div9(sxx) - t.mux * t.mux;
  const float sigy = div9(syy) - t.muy * t.muy;
  const float sigxy = div9(sxy) - t.mux * t.muy;
  t.A1 = 2.f * t.mux * t.muy + C1;
  t.A2 = 2.f * sigxy + C2;
  t.B1 = t.mux * t.mux + t.muy * t.muy + C1;
  t.B2 = sigx + sigy + C2;
  t.n = t.A1 * t.A2;
  t.d = t.B1 * t.B2;
  t.raw = (1.f - t.n / t.d) / 2.f;
  return t;
}

// d (clamped (1 - SSIM)/2 map, already weighted by Gq) / d (the window's pooled moments): the three coefficients pass B gathers
// g_warped(p) = (1/9) * sum over windows containing p of (a + 2 b y(p) + c x(p))
struct SsimCoef { float a, b, c; };
__device__ inline SsimCoef ssim_coef(const SsimTerms& t, float G_ssim) {
#pragma clang fp contract(off)
  // the clamp of (1 - n/d)/2 to [0, 1] passes the gradient where 0 <= raw <= 1, i.e. -d <= n <= d (d > 0: both of its factors
  // carry a positive constant): n/d is correctly rounded and monotone in n, so fl(n/d) <= 1 exactly when n <= d — no division
  const float pass = (t.n <= t.d && t.n >= -t.d) ? 1.f : 0.f;
  const float Gq = div3(G_ssim) * (-0.5f) * pass;
  const float r1 = 1.f / t.d;            // one division for the three coefficients
  const float r2 = r1 * r1;
  SsimCoef k;
  k.a = Gq * ((2.f * t.mux * (t.A2 - t.A1)) * t.d - t.n * (2.f * t.muy * (t.B2 - t.B1))) * r2;
  k.b = Gq * (-(t.n * t.B1)) * r2;
  k.c = Gq * (2.f * t.A1) * r1;
  return k;
}

__device__ inline float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// edge weight exp(-mean_c |I(p) - I(p+e)|) from the three channel pairs
__device__ inline float edge_weight(float a0, float b0, float a1, float b1, float a2, float b2) {
#pragma clang fp contract(off)
  const float m = div3(fabsf(a0 - b0) + fabsf(a1 - b1) + fabsf(a2 - b2));
  return expf(-m);
}

// ---- LinearWarping: sample position, bilinear weights, validity ---------------------------------------
// grid (x -/+ d, y) normalised 2x/w - 1, 2y/h - 1; grid_sample(bilinear, border, align_corners=False) un-normalises
// ((g + 1) * size - 1) / 2 and clips to [0, size - 1] with zero gradient at the clip; valid = -1 <= g <= 1 on both axes.
struct WarpGeom {
  int x0, y0;                 // north-west tap (always inside the image after the clip)
  bool bx1, by1;              // east / south taps inside the image
  float wx0, wx1, wy0, wy1;   // bilinear weights
  float mx;                   // d(clipped x)/d(unclipped x): 0 or 1
  int valid;
};

__device__ inline float clip_border(float v, int size, float& mult) {
  const float hi = (float)(size - 1);
  if (v <= 0.f) { mult = 0.f; return 0.f; }
  if (v >= hi) { mult = 0.f; return hi; }
  mult = 1.f;
  return v;
}

__device__ inline WarpGeom warp_geom(int x, int y, float d, int H, int W, int r2l) {
#pragma clang fp contract(off)
  WarpGeom g;
  const float fx = r2l ? (float)x - d : (float)x + d;
  const float fy = (float)y;
  const float nx = (2.f * fx) / (float)W - 1.0f;
  const float ny = (2.f * fy) / (float)H - 1.0f;
  g.valid = (nx >= -1.0f && nx <= 1.0f && ny >= -1.0f && ny <= 1.0f) ? 1 : 0;
  const float ux = __builtin_fmaf(nx + 1.f, (float)W, -1.f) * 0.5f;
  const float uy = __builtin_fmaf(ny + 1.f, (float)H, -1.f) * 0.5f;
  float my;
  const float ix = clip_border(ux, W, g.mx);
  const float iy = clip_border(uy, H, my);
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  g.x0 = (int)fx0; g.y0 = (int)fy0;
  g.wx1 = ix - fx0; g.wx0 = (fx0 + 1.f) - ix;
  g.wy1 = iy - fy0; g.wy0 = (fy0 + 1.f) - iy;
  g.bx1 = g.x0 + 1 <= W - 1; g.by1 = g.y0 + 1 <= H - 1;
  return g;
}

__device__ inline float warp_interp(float nw, float ne, float sw, float se, const WarpGeom& g) {
#pragma clang fp contract(off)
  return nw * (g.wx0 * g.wy0) + ne * (g.wx1 * g.wy0) + sw * (g.wx0 * g.wy1) + se * (g.wx1 * g.wy1);
}
// d warped / d (sample x)
__device__ inline float warp_dix(float nw, float ne, float sw, float se, const WarpGeom& g) {
#pragma clang fp contract(off)
  return (ne - nw) * g.wy0 + (se - sw) * g.wy1;
}
// gradient w.r.t. the disparity from gix = sum_c g_warped_c * dix_c:  d ix / d nx = W/2 (times the clip multiplier),
// d nx / d fx = 2/W, d fx / d disp = -/+ 1
__device__ inline float warp_gdisp(float gix, const WarpGeom& g, int W, int r2l) {
#pragma clang fp contract(off)
  const float g_nx = gix * (g.mx * ((float)W / 2.f));
  const float g_fx = g_nx * (2.f / (float)W);
  return r2l ? -g_fx : g_fx;
}
