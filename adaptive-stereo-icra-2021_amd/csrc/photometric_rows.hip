// The adaptation step's loss tail (adapt.py:78-86, monodepth_single_loss) as row-walking strips:
//   forward  = LinearWarping of the right image (a9) + monodepth loss map (a10) + masked sum / count      — ONE pass
//   backward = d loss / d disparity through SSIM, L1, smoothness and the warp                               — ONE pass
// Reference semantics: models/linear_warping.py:18-57, utils/loss_functions.py:41-138, adapt.py:81-83; the arithmetic is
// photometric_dev.h's, shared with the one-thread-per-pixel generation (photometric.hip, resample.hip), whose results these
// kernels reproduce bit for bit (tests/test_gpu_kernels.py: the chain in one node against the separate functions).
//
// Shape of the work.  A wave owns 64 adjacent columns and walks down a strip of rows, one row per step.  Everything a 3x3
// window needs from the neighbouring columns comes from the neighbouring LANES by wave-wide DPP shifts (v_mov_b32_dpp
// wave_shr:1 / wave_shl:1 — checked on the chip by tests/tools/scratch/dpp_probe.hip: no LDS, no shuffles, no barriers;
// the compiler keeps each shifted value in a register for its three uses), everything it needs from the rows above
// from registers: each input value is loaded ONCE per strip (the one-thread-per-pixel kernels loaded ~100 values per pixel
// in the forward pass, ~190 in the backward pass, and wrote / re-read ten coefficient planes between the two backward passes).
// The 3x3 sums keep the reference's order — row-major, one addition per tap, (((a00 + a01) + a02) + a10) ... — so the pooled
// moments are the bits avg_pool2d gives: per arriving row the window that ends there, the one in its middle and the one that
// starts take three additions each.  The backward pass chains two such stencils (pooled moments -> coefficients, coefficients ->
// gradient of the warped image) inside the same walk, two rows of delay and two columns / rows of halo, so the ten coefficient
// planes never exist.
// Traffic: forward reads pred, left, right (gathered) and writes warped + mask: 41 B per pixel; backward reads pred, left,
// right and writes two planes (+ the 12 B per pixel of the pass that subtracts the per-image mean term).  What bounds the
// two kernels is their VECTOR work, not HBM: ~500 / ~700 instructions per pixel-row (24 window sums x 10, 9 IEEE divisions,
// 2 expf): 40 / 59 us at 4 pairs of 375 x 1242, 0.24 / 0.14 of the 8 TB/s peak on their algorithmic bytes (0.29 / 0.17 at 32
// pairs) — against 237 us for the launches they replace.
#include "photometric_dev.h"
#pragma clang fp contract(off)

namespace {

// lane i reads lane i-1 / lane i+1 of the wave; the wave's first / last lane reads 0 (they are halo lanes).
// EVERY lane must be active where these execute: they are only called at the top level of the row loop.
__device__ inline float from_left(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ inline float from_right(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// One row of one 3x3 row-major running sum.  `v` is this lane's tap of the arriving row r.  Returns the finished sum of the
// window centred on row r-1; `mid` becomes the window centred on r (two rows in), `start` the one centred on r+1 (one row).
// Taps outside the image are zeros (x + 0 is exact, so skipping them — as the reference's pooling does — gives the same bits).
__device__ inline float win_step(float v, float& mid, float& start) {
  const float l = from_left(v), r = from_right(v);
  float fin = mid + l; fin += v; fin += r;
  float m = start + l; m += v; m += r;
  float s = l; s += v; s += r;
  mid = m; start = s;
  return fin;
}

struct RowsGeom {
  int B, H, W;
  int R;          // output rows per strip
  int nstrips;    // ceil(H / R)
  int ncb;        // column blocks per row: ceil(W / (64 - 2 * halo))
};

// unit (one wave's strip) -> image, first output row, first lane's column
template <int HALO>
__device__ inline bool unit_of(const RowsGeom& g, int unit, int& b, int& y0, int& yend, int& x) {
  const int units = g.B * g.nstrips * g.ncb;
  if (unit >= units) return false;
  const int cb = unit % g.ncb;
  const int t = unit / g.ncb;
  const int s = t % g.nstrips;
  b = t / g.nstrips;
  y0 = s * g.R;
  yend = min(y0 + g.R, g.H);
  x = cb * (64 - 2 * HALO) - HALO + (int)(threadIdx.x & 63);
  return true;
}

// The four taps of the warp for one channel plane.  No branches: every address is inside the image (the sample position is
// clipped to it; a tap beyond the last column / row reads its neighbour instead and is zeroed afterwards, the way
// warp_fwd_kernel's predicates zero it), offsets are 32-bit from the wave-uniform plane base.
struct Taps { float nw, ne, sw, se; };
__device__ inline Taps load_taps(const float* __restrict__ p, int W, const WarpGeom& g) {
  const int o = g.y0 * W + g.x0;
  const int dx = g.bx1 ? 1 : 0, dy = g.by1 ? W : 0;
  Taps t;
  t.nw = p[o];
  t.ne = p[o + dx];
  t.sw = p[o + dy];
  t.se = p[o + dy + dx];
  return t;
}
// ... applied where the taps are consumed, one step after the loads were issued (a select next to the load would make the
// issuing step wait for it)
__device__ inline Taps zero_outside(Taps t, const WarpGeom& g) {
  t.ne = g.bx1 ? t.ne : 0.f;
  t.sw = g.by1 ? t.sw : 0.f;
  t.se = (g.bx1 && g.by1) ? t.se : 0.f;
  return t;
}
// offset of (r, x) clamped into the image: always a valid address; the caller zeroes what lies outside
__device__ inline int clamped_offset(int r, int x, int H, int W) {
  return min(max(r, 0), H - 1) * W + min(max(x, 0), W - 1);
}

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
struct RowsFwdArgs {
  const float* pred;       // [B][H][W]
  const float* img;        // [B][3][H][W]   the true (left) image
  const float* src;        // WARP: the right image, sampled at x - pred;  else: the warped image itself
  const float* mean_disp;  // [B]
  float sw;
  RowsGeom g;
  float* total; float* l1; float* ssim; float* smooth;     // [B][H][W] each, any may be null
  float* warped;           // WARP: [B][3][H][W] out (may be null)
  uint8_t* mask;           // WARP: [B][H][W] out (may be null)
  double* partial;         // SUM: [units][2]  masked sum of `total`, count
};

template <bool WARP, bool SUM>
__global__ __launch_bounds__(256) void photo_rows_fwd_kernel(RowsFwdArgs a) {
  const RowsGeom g = a.g;
  const int unit = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform: rows and bounds live in SGPRs
  int b, y0, yend, x;
  if (!unit_of<1>(g, unit, b, y0, yend, x)) return;       // whole waves only: every lane of a live wave stays active
  const int lane = threadIdx.x & 63;
  const int H = g.H, W = g.W;
  const long plane = (long)H * W;
  const bool col_in = x >= 0 && x < W;
  const bool col_out = col_in && lane >= 1 && lane < 63;   // lanes that own an output column
  const float* P = a.pred + (long)b * plane;
  const float* I = a.img + (long)b * 3 * plane;
  const float* S = a.src + (long)b * 3 * plane;
  const float den = a.mean_disp[b] + 1e-7f;

  // what one row hands to the next steps
  float mid[15], start[15];
#pragma unroll
  for (int q = 0; q < 15; ++q) { mid[q] = 0.f; start[q] = 0.f; }
  float px[3] = {0.f, 0.f, 0.f}, py[3] = {0.f, 0.f, 0.f};       // previous row: true image, warped image
  float pnd = 0.f;                                                // previous row: normalised disparity
  int pvalid = 0;
  double acc_s = 0.0, acc_c = 0.0;

  // software pipeline: row r's values were requested one step earlier (for the warp: its disparity two steps earlier)
  const int r_first = y0 - 1;
  auto in_rows = [&](int r) { return r >= 0 && r < H; };
  const int planei = H * W;
  // (nothing that a load returns is touched in the step that issues it: the selects that zero what lies outside the image
  // are applied one step later, so a step never waits for its own requests)
  float nP = P[clamped_offset(r_first, x, H, W)];             // disparity of row r
  float n2P = P[clamped_offset(r_first + 1, x, H, W)];        // ... of row r + 1, not yet zeroed outside the image
  nP = (col_in && in_rows(r_first)) ? nP : 0.f;
  float nX[3]; Taps nT[3]; float nY[3]; WarpGeom ng;
  {
    const int o = clamped_offset(r_first, x, H, W);
#pragma unroll
    for (int c = 0; c < 3; ++c) nX[c] = I[c * planei + o];
    if (WARP) {
      ng = warp_geom(x, r_first, nP, H, W, 1);
#pragma unroll
      for (int c = 0; c < 3; ++c) nT[c] = load_taps(S + c * planei, W, ng);
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) nY[c] = S[c * planei + o];
    }
  }

  for (int r = r_first; r <= yend; ++r) {
    // ---- this row's values (loaded during the previous step; zero outside the image) ----
    const bool ok = col_in && in_rows(r);
    const float cP = nP;
    float cx[3], cy[3];
    int cvalid = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) cx[c] = ok ? nX[c] : 0.f;
    if (WARP) {
      cvalid = ok ? ng.valid : 0;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const Taps q = zero_outside(nT[c], ng);
        const float t = warp_interp(q.nw, q.ne, q.sw, q.se, ng);      // computed by every lane, then selected
        cy[c] = ok ? t : 0.f;
      }
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) cy[c] = ok ? nY[c] : 0.f;
    }
    // ---- request the next row (one row past the strip's last is requested too: a valid address, never used) ----
    {
      const int rn = r + 1;
      nP = (col_in && in_rows(rn)) ? n2P : 0.f;
      n2P = P[clamped_offset(rn + 1, x, H, W)];
      const int o = clamped_offset(rn, x, H, W);
#pragma unroll
      for (int c = 0; c < 3; ++c) nX[c] = I[c * planei + o];
      if (WARP) {
        ng = warp_geom(x, rn, nP, H, W, 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) nT[c] = load_taps(S + c * planei, W, ng);
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) nY[c] = S[c * planei + o];
      }
    }
    // ---- the 3x3 running sums: the window centred on row r - 1 completes ----
    float fin[15];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float xv = cx[c], yv = cy[c];
      fin[5 * c + 0] = win_step(xv, mid[5 * c + 0], start[5 * c + 0]);
      fin[5 * c + 1] = win_step(yv, mid[5 * c + 1], start[5 * c + 1]);
      fin[5 * c + 2] = win_step(xv * xv, mid[5 * c + 2], start[5 * c + 2]);
      fin[5 * c + 3] = win_step(yv * yv, mid[5 * c + 3], start[5 * c + 3]);
      fin[5 * c + 4] = win_step(xv * yv, mid[5 * c + 4], start[5 * c + 4]);
    }
    const float cnd = cP / den;
    // neighbours of the previous row (DPP: all lanes active here)
    const float pnd_r = from_right(pnd);
    const float pxr0 = from_right(px[0]), pxr1 = from_right(px[1]), pxr2 = from_right(px[2]);
    // ---- emit row r - 1 ----
    const int ye = r - 1;
    if (ye >= y0) {
      float s_acc = 0.f, l_acc = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const SsimTerms t = ssim_from_sums(fin[5 * c], fin[5 * c + 1], fin[5 * c + 2], fin[5 * c + 3], fin[5 * c + 4]);
        s_acc += fminf(fmaxf(t.raw, 0.f), 1.f);
        l_acc += fabsf(px[c] - py[c]);
      }
      const float ps = div3(s_acc), pl = div3(l_acc);
      float sm = 0.f;
      const float tx = fabsf(pnd - pnd_r) * edge_weight(px[0], pxr0, px[1], pxr1, px[2], pxr2);
      const float ty = fabsf(pnd - cnd) * edge_weight(px[0], cx[0], px[1], cx[1], px[2], cx[2]);
      if (x < W - 1) sm += tx;
      if (ye < H - 1) sm += ty;
      const float photo = 0.85f * ps + 0.15f * pl;
      const float tot = photo + a.sw * sm;
      if (col_out) {
        const long i = (long)b * plane + (long)ye * W + x;
        if (a.total) a.total[i] = tot;
        if (a.l1) a.l1[i] = pl;
        if (a.ssim) a.ssim[i] = ps;
        if (a.smooth) a.smooth[i] = sm;
        if (WARP) {
          if (a.warped) {
#pragma unroll
            for (int c = 0; c < 3; ++c) a.warped[((long)b * 3 + c) * plane + (long)ye * W + x] = py[c];
          }
          if (a.mask) a.mask[i] = (uint8_t)pvalid;
          if (SUM && pvalid) { acc_s += (double)tot; acc_c += 1.0; }
        }
      }
    }
    // ---- hand over ----
#pragma unroll
    for (int c = 0; c < 3; ++c) { px[c] = cx[c]; py[c] = cy[c]; }
    pnd = cnd; pvalid = cvalid;
  }
  if (SUM) {
    acc_s = wave_sum_d(acc_s); acc_c = wave_sum_d(acc_c);
    if (lane == 0) { a.partial[2 * (long)unit] = acc_s; a.partial[2 * (long)unit + 1] = acc_c; }
  }
}

// masked sum / count over all units: one 256-thread workgroup, fixed order
__global__ __launch_bounds__(256) void rows_masked_finalize_kernel(const double* __restrict__ partial, int n, float* __restrict__ out4) {
  __shared__ double red[2][4];
  double s = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) { s += partial[2 * (long)i]; c += partial[2 * (long)i + 1]; }
  s = wave_sum_d(s); c = wave_sum_d(c);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    c = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    out4[0] = (float)s; out4[1] = (float)c;
    out4[2] = (float)s / (float)c; out4[3] = (float)c;       // fp32 sum / fp32 count, as the reference's .mean() divides
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------------
struct RowsBwdArgs {
  const float* pred; const float* img; const float* right;
  const float* mean_disp;     // [B]
  const float* g_sum;         // [1] or null   } every valid pixel carries g_sum + g_mean / count, every other pixel 0
  const float* g_mean;        // [1] or null   }
  const float* sum_count;     // out4 of the forward pass ([1] = count)
  float sw;
  RowsGeom g;
  float* g_direct;            // [B][H][W] out: g_nd / (mean + eps)          (the mean term is subtracted by the next pass)
  float* g_warp;              // [B][H][W] out: the gradient through the warp
  double* partial;            // [units]: sum of g_nd * pred
};

__global__ __launch_bounds__(256) void photo_rows_bwd_kernel(RowsBwdArgs a) {
  const RowsGeom g = a.g;
  const int unit = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int b, y0, yend, x;
  if (!unit_of<2>(g, unit, b, y0, yend, x)) return;
  const int lane = threadIdx.x & 63;
  const int H = g.H, W = g.W;
  const long plane = (long)H * W;
  const bool col_in = x >= 0 && x < W;
  const bool col_out = col_in && lane >= 2 && lane < 62;
  const float* P = a.pred + (long)b * plane;
  const float* I = a.img + (long)b * 3 * plane;
  const float* S = a.right + (long)b * 3 * plane;
  const float den = a.mean_disp[b] + 1e-7f;
  const float rden = 1.f / den;
  float gs = a.g_sum ? a.g_sum[0] : 0.f;
  if (a.g_mean) gs += a.g_mean[0] / a.sum_count[1];
  auto in_rows = [&](int r) { return r >= 0 && r < H; };

  float mid[15], start[15], kmid[9], kstart[9];
#pragma unroll
  for (int q = 0; q < 15; ++q) { mid[q] = 0.f; start[q] = 0.f; }
#pragma unroll
  for (int q = 0; q < 9; ++q) { kmid[q] = 0.f; kstart[q] = 0.f; }
  // histories: suffix 1 = row r - 1, 2 = row r - 2, 3 = row r - 3
  float x1[3] = {0.f, 0.f, 0.f}, x2[3] = {0.f, 0.f, 0.f};      // true image
  float w1[3] = {0.f, 0.f, 0.f}, w2[3] = {0.f, 0.f, 0.f};      // warped image
  float d1[3] = {0.f, 0.f, 0.f}, d2[3] = {0.f, 0.f, 0.f};      // d warped / d sample x
  float nd1 = 0.f, nd2 = 0.f, nd3 = 0.f;                        // normalised disparity
  float gt1 = 0.f, gt2 = 0.f, gt3 = 0.f;                        // gradient of the loss map at the pixel (gs where valid)
  float mx1 = 0.f, mx2 = 0.f, p1 = 0.f, p2 = 0.f;
  float wy3 = 0.f;                                              // edge weight between rows r - 3 and r - 2
  double acc = 0.0;

  const int r_first = y0 - 2, r_last = yend + 1;
  const int planei = H * W;
  float nP = P[clamped_offset(r_first, x, H, W)];
  float n2P = P[clamped_offset(r_first + 1, x, H, W)];        // (zeroed outside the image one step later: see the forward pass)
  nP = (col_in && in_rows(r_first)) ? nP : 0.f;
  float nX[3]; Taps nT[3];
  WarpGeom ng = warp_geom(x, r_first, nP, H, W, 1);
  {
    const int o = clamped_offset(r_first, x, H, W);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      nX[c] = I[c * planei + o];
      nT[c] = load_taps(S + c * planei, W, ng);
    }
  }

  for (int r = r_first; r <= r_last; ++r) {
    const bool ok = col_in && in_rows(r);
    const float cP = nP;
    float cx[3], cw[3], cd[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const Taps q = zero_outside(nT[c], ng);
      const float tw = warp_interp(q.nw, q.ne, q.sw, q.se, ng);      // computed by every lane, then selected
      const float td = warp_dix(q.nw, q.ne, q.sw, q.se, ng);
      cx[c] = ok ? nX[c] : 0.f;
      cw[c] = ok ? tw : 0.f;
      cd[c] = ok ? td : 0.f;
    }
    const float cgt = (ok && ng.valid) ? gs : 0.f;
    const float cmx = ng.mx;
    {
      const int rn = r + 1;
      nP = (col_in && in_rows(rn)) ? n2P : 0.f;
      n2P = P[clamped_offset(rn + 1, x, H, W)];
      ng = warp_geom(x, rn, nP, H, W, 1);
      const int o = clamped_offset(rn, x, H, W);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        nX[c] = I[c * planei + o];
        nT[c] = load_taps(S + c * planei, W, ng);
      }
    }
    // ---- stencil 1: pooled moments of the window centred on row r - 1 ----
    float fin[15];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float xv = cx[c], yv = cw[c];
      fin[5 * c + 0] = win_step(xv, mid[5 * c + 0], start[5 * c + 0]);
      fin[5 * c + 1] = win_step(yv, mid[5 * c + 1], start[5 * c + 1]);
      fin[5 * c + 2] = win_step(xv * xv, mid[5 * c + 2], start[5 * c + 2]);
      fin[5 * c + 3] = win_step(yv * yv, mid[5 * c + 3], start[5 * c + 3]);
      fin[5 * c + 4] = win_step(xv * yv, mid[5 * c + 4], start[5 * c + 4]);
    }
    // ---- coefficients at (r - 1, x) (zero outside the image and where the loss map carries no gradient) ----
    const float G_ssim1 = 0.85f * gt1;
    float kfin[9];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const SsimTerms t = ssim_from_sums(fin[5 * c], fin[5 * c + 1], fin[5 * c + 2], fin[5 * c + 3], fin[5 * c + 4]);
      const SsimCoef k = ssim_coef(t, G_ssim1);
      // ---- stencil 2: their 3x3 sums, window centred on row r - 2 ----
      kfin[3 * c + 0] = win_step(k.a, kmid[3 * c + 0], kstart[3 * c + 0]);
      kfin[3 * c + 1] = win_step(k.b, kmid[3 * c + 1], kstart[3 * c + 1]);
      kfin[3 * c + 2] = win_step(k.c, kmid[3 * c + 2], kstart[3 * c + 2]);
    }
    const float cnd = cP / den;
    // neighbours of row r - 2 (DPP at the top level of the loop: all lanes active)
    const float nd2_r = from_right(nd2), nd2_l = from_left(nd2), gt2_l = from_left(gt2);
    const float wx2 = edge_weight(x2[0], from_right(x2[0]), x2[1], from_right(x2[1]), x2[2], from_right(x2[2]));
    const float wx2_l = from_left(wx2);
    const float wy2 = edge_weight(x2[0], x1[0], x2[1], x1[1], x2[2], x1[2]);       // between rows r - 2 and r - 1
    // ---- emit row r - 2 ----
    const int ye = r - 2;
    if (ye >= y0 && ye < yend) {
      // gradient of the warped image, then through the warp
      const float G_l1 = 0.15f * gt2;
      float gix = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float xv = x2[c], yv = w2[c];
        float gw = div9(kfin[3 * c] + 2.f * kfin[3 * c + 1] * yv + kfin[3 * c + 2] * xv);
        gw += div3(G_l1) * (-sgn(xv - yv));
        gix = __builtin_fmaf(gw, d2[c], gix);
      }
      WarpGeom gg; gg.mx = mx2;
      const float g_warp = warp_gdisp(gix, gg, W, 1);
      // smoothness: derivative w.r.t. the normalised disparity at this pixel
      float g_nd = 0.f;
      if (x < W - 1) g_nd += (a.sw * gt2) * wx2 * sgn(nd2 - nd2_r);
      if (x > 0) g_nd -= (a.sw * gt2_l) * wx2_l * sgn(nd2_l - nd2);
      if (ye < H - 1) g_nd += (a.sw * gt2) * wy2 * sgn(nd2 - nd1);
      if (ye > 0) g_nd -= (a.sw * gt3) * wy3 * sgn(nd3 - nd2);
      if (col_out) {
        const long i = (long)b * plane + (long)ye * W + x;
        a.g_direct[i] = g_nd * rden;
        a.g_warp[i] = g_warp;
        acc += (double)g_nd * (double)p2;
      }
    }
    // ---- hand over ----
#pragma unroll
    for (int c = 0; c < 3; ++c) { x2[c] = x1[c]; x1[c] = cx[c]; w2[c] = w1[c]; w1[c] = cw[c]; d2[c] = d1[c]; d1[c] = cd[c]; }
    nd3 = nd2; nd2 = nd1; nd1 = cnd;
    gt3 = gt2; gt2 = gt1; gt1 = cgt;
    mx2 = mx1; mx1 = cmx; p2 = p1; p1 = cP;
    wy3 = wy2;
  }
  acc = wave_sum_d(acc);
  if (lane == 0) a.partial[unit] = acc;
}

// per-image sum of the units' partials: one wave per image, lanes stride over the units, fixed-order butterfly
__global__ __launch_bounds__(64) void rows_image_finalize_kernel(const double* __restrict__ partial, int units_per_image,
                                                                 float* __restrict__ out) {
  const int b = blockIdx.x;
  double s = 0.0;
  for (int i = threadIdx.x; i < units_per_image; i += 64) s += partial[(long)b * units_per_image + i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) out[b] = (float)s;
}

// g_pred = g_warp + (g_direct - (S_b / den^2) / plane): the term of d loss / d pred that goes through the per-image mean.
// Flat over [B * plane] in 16-byte pieces (an image's plane need not be a multiple of four pixels).
__global__ __launch_bounds__(256) void rows_mean_term_kernel(const float* __restrict__ g_warp, const float* __restrict__ mean_disp,
                                                             const float* __restrict__ sum_gnd_pred, long n, long plane,
                                                             float* __restrict__ g_pred) {
  const long o = 4L * ((long)blockIdx.x * 256 + threadIdx.x);
  if (o >= n) return;
  auto mean_term = [&](long b) {
    const float den = mean_disp[b] + 1e-7f;
    const float r = 1.f / den;
    return (sum_gnd_pred[b] * r * r) / (float)plane;
  };
  const long b0 = o / plane;
  const float c0 = mean_term(b0);
  if (o + 3 < n && (o + 3) / plane == b0) {
    const f32x4 w = *reinterpret_cast<const f32x4*>(g_warp + o);
    f32x4 d = *reinterpret_cast<const f32x4*>(g_pred + o);
    d.x = w.x + (d.x - c0); d.y = w.y + (d.y - c0); d.z = w.z + (d.z - c0); d.w = w.w + (d.w - c0);
    *reinterpret_cast<f32x4*>(g_pred + o) = d;
  } else {
    for (int k = 0; k < 4 && o + k < n; ++k) {
      const long bk = (o + k) / plane;
      const float ck = bk == b0 ? c0 : mean_term(bk);
      g_pred[o + k] = g_warp[o + k] + (g_pred[o + k] - ck);
    }
  }
}

// Strip height.  A wave walks R + 2 * halo rows for R rows of output and a SIMD works through its waves at the full vector
// issue rate only with two or more of them resident, so the launch lasts about ceil(waves / 1024 SIMDs) * (R + 2 * halo) row
// steps (twice that while there is at most one wave per SIMD): the candidate with the smallest product, the taller on a tie.
// 4 pairs of 375 x 1242: 16 rows (2,016 waves: two per SIMD, 18 / 20 steps each); 32 pairs: 64; one pair: 8.
int pick_rows(int B, int H, int ncb, int halo) {
  const int cand[] = {64, 48, 32, 24, 20, 16, 12, 8};
  int best = 8;
  long best_cost = -1;
  for (int R : cand) {
    const long waves = (long)B * ((H + R - 1) / R) * ncb;
    const long per = (waves + 1023) / 1024;
    const long cost = per * (R + 2 * halo) * (per <= 1 ? 2 : 1);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = R; }
  }
  return best;
}

RowsGeom make_geom(int B, int H, int W, int halo) {
  RowsGeom g;
  g.B = B; g.H = H; g.W = W;
  g.ncb = (W + (64 - 2 * halo) - 1) / (64 - 2 * halo);
  g.R = pick_rows(B, H, g.ncb, halo);
  g.nstrips = (H + g.R - 1) / g.R;
  return g;
}
inline long units_of(const RowsGeom& g) { return (long)g.B * g.nstrips * g.ncb; }

struct RowsWs { float* mean; float* sum; double* partial; float* plane; };
}  // namespace

// the one-thread-per-pixel generation's per-image mean (photometric.hip)
int as_photometric_image_mean(const float* pred, int B, long plane, double* partial, float* mean, hipStream_t st);

#define ROWS_MEAN_BLOCKS 512
// floats: [mean B | sum B, padded to 16 each] + fp64 partials (max of: the mean's B x 512, the forward's 2 x units, the
// backward's units) + one [B][H][W] plane
extern "C" int64_t as_photometric_chain_workspace(int B, int H, int W) {
  if (B <= 0 || H <= 1 || W <= 1) return -1;
  const int64_t Bp = (B + 15) / 16 * 16;
  const int64_t max_units = (int64_t)B * ((H + 7) / 8) * ((W + 59) / 60);
  int64_t dbl = 2 * max_units;
  if (dbl < (int64_t)B * ROWS_MEAN_BLOCKS) dbl = (int64_t)B * ROWS_MEAN_BLOCKS;
  return 2 * Bp + 2 * dbl + (int64_t)B * H * W + 16;
}

static RowsWs rows_carve(float* ws, int B, int H, int W) {
  RowsWs w;
  const int64_t Bp = (B + 15) / 16 * 16;
  const int64_t max_units = (int64_t)B * ((H + 7) / 8) * ((W + 59) / 60);
  int64_t dbl = 2 * max_units;
  if (dbl < (int64_t)B * ROWS_MEAN_BLOCKS) dbl = (int64_t)B * ROWS_MEAN_BLOCKS;
  w.mean = ws; w.sum = ws + Bp;
  w.partial = reinterpret_cast<double*>(ws + 2 * Bp);
  w.plane = ws + 2 * Bp + 2 * dbl;
  return w;
}

static bool rows_args_ok(int B, int H, int W) {
  return B > 0 && H > 1 && W > 1 && B <= 65535 && 3L * H * W < (1L << 30);
}

// Forward of the whole tail.  warped [B][3][H][W] and mask [B][H][W] (uint8) are outputs (left_warped/<s> of the step's output
// dict; the mask is what loss[mask].mean() indexes with); out4 = {masked sum, count, mean, count}.  The workspace keeps the
// per-image mean disparity for the backward pass (pass it as fwd_workspace).
extern "C" int as_photometric_chain_fwd(const float* pred, const float* left, const float* right, int B, int H, int W,
                                        float smoothness_weight, float* warped, uint8_t* mask, float* out4,
                                        float* workspace, void* stream) {
  AS_CHECK_ARG(pred && left && right && out4 && workspace && rows_args_ok(B, H, W), "as_photometric_chain_fwd: bad argument");
  AS_CHECK_ARG(((uintptr_t)workspace & 15) == 0, "as_photometric_chain_fwd: workspace must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  RowsWs w = rows_carve(workspace, B, H, W);
  const long plane = (long)H * W;
  as_prof_mark(AS_PROF_LOSS_FWD, st, 1, 0.0);
  if (int e = as_photometric_image_mean(pred, B, plane, w.partial, w.mean, st)) return e;
  RowsFwdArgs a;
  a.pred = pred; a.img = left; a.src = right; a.mean_disp = w.mean; a.sw = smoothness_weight;
  a.g = make_geom(B, H, W, 1);
  a.total = nullptr; a.l1 = nullptr; a.ssim = nullptr; a.smooth = nullptr;
  a.warped = warped; a.mask = mask; a.partial = w.partial;
  const long units = units_of(a.g);
  hipLaunchKernelGGL((photo_rows_fwd_kernel<true, true>), dim3(as_div_up(units, 4)), dim3(256), 0, st, a);
  AS_CHECK_LAUNCH("as_photometric_chain_fwd");
  hipLaunchKernelGGL(rows_masked_finalize_kernel, dim3(1), dim3(256), 0, st, w.partial, (int)units, out4);
  // pred, left, right read; warped and the mask written
  as_prof_mark(AS_PROF_LOSS_FWD, st, 0, (double)B * plane * (4.0 * (1 + 3 + 3 + 3) + 1.0));
  AS_CHECK_LAUNCH("as_photometric_chain_fwd(finalize)");
  return AS_OK;
}

// The loss maps alone from a given warped image (utils/loss_functions.py:106-138), any of the four may be NULL.
extern "C" int as_monodepth_loss_rows_fwd(const float* pred, const float* img, const float* warped, int B, int H, int W,
                                          float smoothness_weight, float* total, float* l1, float* ssim, float* smooth,
                                          float* workspace, void* stream) {
  AS_CHECK_ARG(pred && img && warped && workspace && rows_args_ok(B, H, W), "as_monodepth_loss_rows_fwd: bad argument");
  AS_CHECK_ARG(((uintptr_t)workspace & 15) == 0, "as_monodepth_loss_rows_fwd: workspace must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  RowsWs w = rows_carve(workspace, B, H, W);
  const long plane = (long)H * W;
  if (int e = as_photometric_image_mean(pred, B, plane, w.partial, w.mean, st)) return e;
  RowsFwdArgs a;
  a.pred = pred; a.img = img; a.src = warped; a.mean_disp = w.mean; a.sw = smoothness_weight;
  a.g = make_geom(B, H, W, 1);
  a.total = total; a.l1 = l1; a.ssim = ssim; a.smooth = smooth;
  a.warped = nullptr; a.mask = nullptr; a.partial = nullptr;
  hipLaunchKernelGGL((photo_rows_fwd_kernel<false, false>), dim3(as_div_up(units_of(a.g), 4)), dim3(256), 0, st, a);
  AS_CHECK_LAUNCH("as_monodepth_loss_rows_fwd");
  return AS_OK;
}

// Backward of the whole tail: g_pred = d (g_sum * sum + g_mean * mean) / d pred.  fwd_workspace = the forward call's workspace
// (its per-image mean disparity is reused), out4 = the forward call's out4.
extern "C" int as_photometric_chain_bwd(const float* g_sum, const float* g_mean, const float* out4, const float* pred,
                                        const float* left, const float* right, int B, int H, int W, float smoothness_weight,
                                        float* g_pred, float* workspace, const float* fwd_workspace, void* stream) {
  AS_CHECK_ARG(pred && left && right && g_pred && workspace && fwd_workspace && rows_args_ok(B, H, W),
               "as_photometric_chain_bwd: bad argument");
  AS_CHECK_ARG((g_sum || g_mean) && (g_mean == nullptr || out4 != nullptr),
               "as_photometric_chain_bwd: at least one of g_sum / g_mean, and out4 with g_mean");
  AS_CHECK_ARG(((uintptr_t)workspace & 15) == 0, "as_photometric_chain_bwd: workspace must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  RowsWs w = rows_carve(workspace, B, H, W);
  const long plane = (long)H * W;
  as_prof_mark(AS_PROF_LOSS_BWD, st, 1, 0.0);
  RowsBwdArgs a;
  a.pred = pred; a.img = left; a.right = right; a.mean_disp = fwd_workspace;
  a.g_sum = g_sum; a.g_mean = g_mean; a.sum_count = out4; a.sw = smoothness_weight;
  a.g = make_geom(B, H, W, 2);
  a.g_direct = g_pred; a.g_warp = w.plane; a.partial = w.partial;
  const long units = units_of(a.g);
  hipLaunchKernelGGL(photo_rows_bwd_kernel, dim3(as_div_up(units, 4)), dim3(256), 0, st, a);
  AS_CHECK_LAUNCH("as_photometric_chain_bwd");
  hipLaunchKernelGGL(rows_image_finalize_kernel, dim3(B), dim3(64), 0, st, w.partial, a.g.nstrips * a.g.ncb, w.sum);
  AS_CHECK_LAUNCH("as_photometric_chain_bwd(sum)");
  const long n = (long)B * plane;
  hipLaunchKernelGGL(rows_mean_term_kernel, dim3(as_div_up((n + 3) / 4, 256)), dim3(256), 0, st, w.plane, fwd_workspace, w.sum,
                     n, plane, g_pred);
  // pred, left, right read; g_pred written (the two planes between the passes are the implementation's)
  as_prof_mark(AS_PROF_LOSS_BWD, st, 0, (double)B * plane * 4.0 * (1 + 3 + 3 + 1));
  AS_CHECK_LAUNCH("as_photometric_chain_bwd(mean term)");
  return AS_OK;
}
