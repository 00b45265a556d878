// The refinement's output layer (stereo_net.py:102, 116-121): conv2d_out = Conv2d(32, 1, 3, padding 1) on the last BasicBlock's
// output, + the up-sampled disparity, ReLU — with, in a training step, that block's BatchNorm + LeakyReLU + skip connection
// (a6 = lrelu(BN(z6)) + a5, stereo_net.py:44-51) applied ON THE WAY IN instead of in a pass of its own.
//
// Until round 4 the training forward ended with bn_act_fwd (read z6, a5, write a6: 130 us at 4 pairs, HBM-bound) followed by
// conv32to1_2d_fwd_kernel (read a6 again: 79 us — one voxel per thread, eight 16-byte loads 128 bytes apart per lane, 3 TB/s);
// a first fusion on that kernel's access pattern was slower than the two launches (round 3).  Here:
//   * a workgroup walks DOWN a strip of 126 output columns; every input row (128 voxels with the one-voxel halo) is fetched
//     ONCE by coalesced 16-byte chunks (8 lanes per voxel), activated in registers, written back as the by-product a6, and
//     staged into one swizzled LDS row;
//   * "project, then gather" as before, but the projection P_t[u] = sum_c a[u][c] w[c][t] (nine taps per voxel) runs on the
//     matrix cores: M = 32 voxels per wave, K = 32 channels, N = the 9 taps (of 32 columns: 16 v_mfma_f32_32x32x2_f32 per wave
//     and row — 28 us of matrix time per launch where the HBM time is 130+; the vector ALUs only activate and gather);
//   * the projections of three consecutive rows live in a small LDS ring; an output pixel is bias + nine scalars from it.
// HBM: z6, a5 read, a6 written, once each (+ 1.6 % column halo, + 6 % row halo per 32-row strip) and the two dense maps.
// In inference (x already activated) the same kernel is the plain 32 -> 1 layer with coalesced loads.
#include "as_common.h"

#define RO_COLS 126                      // output columns of a strip (128 staged voxels)
// output rows of a strip, per flavour (two more rows are staged): measured inside a step and the inference pass with 16 / 24 /
// 32 / 47 rows (tests/tools/exp_step.sh refine_out.hip refine_out "-DRO_ROWS_ACT=.. -DRO_ROWS_PLAIN=.."): with the activation on
// the way in 160 / 163 / 155 / 183 us, without it (the inference flavour: half the bytes per row, so the launch wants more
// workgroups) 66 / 75 / 79 / 110 us, and 71 / 62.5 with 8 / 12 rows
#ifndef RO_ROWS_ACT
#define RO_ROWS_ACT 32
#endif
#ifndef RO_ROWS_PLAIN
#define RO_ROWS_PLAIN 12
#endif
#define RO_PW 132                        // pitch of a projection row in LDS (floats)

struct RefineOutArgs {
  const float* x;          // ACT: the last block's pre-activation z (PCL); else the activated input a (PCL)
  const float* skip;       // ACT: the block's input (skip connection, PCL) or null
  const float* scale;      // ACT: the block's BatchNorm as an affine
  const float* shift;
  float* a_out;            // ACT: by-product, the block's output a = lrelu(z * scale + shift) + skip (PCL)
  const float* w;          // [32][9] (conv2d_out.weight[0])
  const float* bias;       // [1] or null
  const float* add_src;    // dense [B][H][W] or null
  float* out;              // dense [B][H][W]
  PclDev g;
  int nseg, nstrips, relu;
  float slope;
};

template <bool ACT>
__global__ __launch_bounds__(256) void refine_out_kernel(RefineOutArgs p) {
  __shared__ __attribute__((aligned(16))) char xrow[128 * 128];        // one activated row: slot s of voxel v holds chunk s ^ ((v >> 1) & 7)
  __shared__ float P[3][9][RO_PW];                                     // projections of rows y-1, y, y+1
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int h = lane >> 5, li = lane & 31;
  const int H = p.g.H, W = p.g.W;
  int blk = blockIdx.x;
  const int seg = blk % p.nseg; blk /= p.nseg;
  const int strip = blk % p.nstrips;
  const int b = blk / p.nstrips;
  const int x0 = seg * RO_COLS;
  constexpr int ROWS = ACT ? RO_ROWS_ACT : RO_ROWS_PLAIN;
  const int ya = strip * ROWS, yb = min(H, ya + ROWS);

  // B operand: w[ci][tap] for ci = 16h + 4q + e, column = tap li (columns 9..31 are zero)
  float Bw[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) Bw[i] = li < 9 ? p.w[(16 * h + i) * 9 + li] : 0.f;
  const float bias = p.bias ? p.bias[0] : 0.f;

  // conversion: thread t owns chunk (t & 7) of the staged voxels (t >> 3) + 32 k, k = 0..3 (columns x0 - 1 + voxel)
  const int cq = t & 7, v0 = t >> 3;
  const int lds_w = v0 * 128 + ((cq ^ ((v0 >> 1) & 7)) << 4);           // + k * 4096 (the swizzle repeats every 16 voxels)
  f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (ACT) { sc = *reinterpret_cast<const f32x4*>(p.scale + cq * 4); sh = *reinterpret_cast<const f32x4*>(p.shift + cq * 4); }
  unsigned colmask = 0u, ownmask = 0u;                      // voxel k lies inside the image / is one of the strip's own columns
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int v = v0 + 32 * k, xx = x0 - 1 + v;
    colmask |= ((xx >= 0 && xx < W) ? 1u : 0u) << k;
    ownmask |= ((v >= 1 && v <= RO_COLS && xx < W) ? 1u : 0u) << k;
  }
  const long row0 = (((long)b * p.g.Hp + p.g.ph) * p.g.Wp + (x0 - 1 + p.g.pw)) * 32 + t * 4;      // + y * Wp * 32 + k * 1024
  auto fetch = [&](int yi, f32x4 (&rz)[4], f32x4 (&rs)[4]) {            // rows outside the image: any valid row (zeroed later)
    const long off = row0 + (long)min(max(yi, 0), H - 1) * p.g.Wp * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      rz[k] = *reinterpret_cast<const f32x4*>(p.x + off + k * 1024);
      if (ACT && p.skip) rs[k] = *reinterpret_cast<const f32x4*>(p.skip + off + k * 1024);
    }
  };
  // matrix phase: this wave projects the staged voxels 32 wave + li; A operand chunk 4h + q
  const int mv = 32 * wave + li;
  const char* a_rd = xrow + mv * 128;
  const int a_sw = (mv >> 1) & 7;

  f32x4 rz[4], rs[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) rs[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  fetch(ya - 1, rz, rs);
  for (int yi = ya - 1; yi <= yb; ++yi) {
    const bool row_in = yi >= 0 && yi < H;                 // (uniform)
    const bool row_own = yi >= ya && yi < yb;
    // ---- activate, by-product, stage ----
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f32x4 y = rz[k];
      if (ACT) {                                           // bn_act_fwd_kernel's arithmetic
        y = y * sc + sh;
        y.x = y.x > 0.f ? y.x : y.x * p.slope; y.y = y.y > 0.f ? y.y : y.y * p.slope;
        y.z = y.z > 0.f ? y.z : y.z * p.slope; y.w = y.w > 0.f ? y.w : y.w * p.slope;
        if (p.skip) y += rs[k];
      }
      if (!row_in || !((colmask >> k) & 1u)) y = (f32x4){0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(xrow + lds_w + k * 4096) = y;
      if (ACT && row_own && ((ownmask >> k) & 1u))
        *reinterpret_cast<f32x4*>(p.a_out + row0 + (long)yi * p.g.Wp * 32 + k * 1024) = y;
    }
    if (yi < yb) fetch(yi + 1, rz, rs);                    // the next row: in flight under this row's matrix phase and gather
    __syncthreads();                                       // A: the row is staged; the ring slot of row yi-3 is free
    // ---- project the row onto the nine taps ----
    f32x16 acc;
    {
      f32x4 xa[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) xa[q] = *reinterpret_cast<const f32x4*>(a_rd + (((4 * h + q) ^ a_sw) << 4));
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[0].x, Bw[0], zero, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[0].y, Bw[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[0].z, Bw[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[0].w, Bw[3], acc, 0, 0, 0);
#pragma unroll
      for (int q = 1; q < 4; ++q) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[q].x, Bw[4 * q + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[q].y, Bw[4 * q + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[q].z, Bw[4 * q + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[q].w, Bw[4 * q + 3], acc, 0, 0, 0);
      }
    }
    if (li < 9) {                                          // column li of the tile = tap li; row (r, h) = voxel
      float* pr = &P[(yi + 1) % 3][li][32 * wave + 4 * h];
#pragma unroll
      for (int r = 0; r < 16; ++r) pr[(r & 3) + 8 * (r >> 2)] = acc[r];
    }
    __syncthreads();                                       // B: projections of row yi are complete; the staged row is free
    // ---- output row yi - 1: bias + nine projected scalars, + up-sampled disparity, ReLU ----
    const int yo = yi - 1;
    if (yo >= ya && t < RO_COLS && x0 + t < W) {
      float s = bias;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const float (*pr)[RO_PW] = P[(yo + ky) % 3];       // input row yo + ky - 1 sits in slot (row + 1) % 3
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) s += pr[ky * 3 + kx][t + kx];
      }
      const long o = ((long)b * H + yo) * W + x0 + t;
      if (p.add_src) s += p.add_src[o];
      if (p.relu) s = s > 0.f ? s : 0.f;
      p.out[o] = s;
    }
  }
}

extern "C" int as_refine_out_ok(const as_pcl* g) {
  if (!as_pcl_ok(g)) return 0;
  // the staged row reaches one voxel left of the strip's first column and 126 right of it: for the last strip of a row that
  // can run up to 127 voxels past the row's end — into the rows below, i.e. for the last image row into the bottom halo,
  // which must hold them
  return g->D == 1 && g->pd == 0 && g->ph >= 1 && g->pw >= 1 && (long)g->ph * (g->W + 2 * g->pw) >= 128 ? 1 : 0;
}

// out = relu?(conv2d_out(a) + bias + add_src), a = lrelu(x * scale + shift) + skip (written to a_out) when scale is given,
// a = x otherwise.
extern "C" int as_refine_out_fwd(const float* x, const float* skip, const float* scale, const float* shift, float slope,
                                 float* a_out, const as_pcl* g, const float* w, const float* bias, const float* add_src,
                                 int relu, float* out, void* stream) {
  AS_CHECK_ARG(as_refine_out_ok(g) == 1, "as_refine_out_fwd: geometry not supported (as_refine_out_ok() == 0)");
  AS_CHECK_ARG(x && w && out, "as_refine_out_fwd: null pointer");
  AS_CHECK_ARG((scale == nullptr) == (shift == nullptr) && (scale == nullptr) == (a_out == nullptr),
               "as_refine_out_fwd: scale, shift and a_out come together (the fused BatchNorm + LeakyReLU) or not at all");
  AS_CHECK_ARG(scale != nullptr || skip == nullptr, "as_refine_out_fwd: a skip connection needs the fused activation");
  AS_CHECK_ARG(a_out == nullptr || (a_out != x && a_out != skip), "as_refine_out_fwd: a_out must not alias an input");
  AS_CHECK_ARG(scale == nullptr || (slope > 0.f && slope < 1.f), "as_refine_out_fwd: slope must lie in (0, 1)");
  RefineOutArgs a;
  a.x = x; a.skip = skip; a.scale = scale; a.shift = shift; a.a_out = a_out; a.w = w; a.bias = bias; a.add_src = add_src;
  a.out = out; a.g = as_make_dev(g); a.relu = relu; a.slope = slope;
  a.nseg = as_div_up(g->W, RO_COLS); a.nstrips = as_div_up(g->H, scale ? RO_ROWS_ACT : RO_ROWS_PLAIN);
  const int grid = g->B * a.nseg * a.nstrips;
  if (scale) hipLaunchKernelGGL(refine_out_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(refine_out_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  AS_CHECK_LAUNCH("as_refine_out_fwd");
  return AS_OK;
}
