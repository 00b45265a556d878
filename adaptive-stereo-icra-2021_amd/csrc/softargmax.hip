// a4 + a5 + a8: 32->1 3-D convolution to logits, soft-argmax disparity regression,
// arg-max index and feature-contrast score.
// Reference semantics:
//   conv3d_alone = nn.Conv3d(32,1,3,padding=1)        stereo_net.py:162,187
//   softmax(+cost, dim=1); sum_d d*p_d                 stereo_net.py:190-192,124-134
//   FCS = sorted[0] - mean(sorted[2:]) over d          utils/feature_contrast.py:12-23
//
// The 32->1 convolution has N=1: a matrix-core tile would waste 31/32 of the MFMA, so it
// is a bandwidth kernel: 8 lanes share a voxel (float4 of channels each: one 128-byte
// line per voxel per tap, fully coalesced), 27 taps from the zero-haloed PCL input, then
// a 3-step wavefront-shuffle (xor 1,2,4) reduction across the 8 lanes.
// Soft-argmax keeps one pixel per lane (d-strided, W-coalesced reads of the logits);
// Dc is 8..24, so the per-pixel reduction over d is a short in-register loop.
#include "as_common.h"

struct OutGeom {
  int kd, kh, kw, pad_d, pad_h, pad_w, dil, ntaps;
  int tap_off[27];       // voxel offsets in the PCL input
};

struct OutConvArgs {
  const float* a;
  const float* w;        // [32][ntaps] PyTorch order
  const float* bias;     // [1] or null
  const float* add_src;  // dense [B][D][H][W] added before the optional ReLU, or null
  float* out;            // dense [B][D][H][W]
  int relu;
  PclDev g;
  long M;
  OutGeom k;
};

__global__ __launch_bounds__(256) void conv32to1_fwd_kernel(OutConvArgs p) {
  __shared__ float sw[27 * 32];
  const int NT = p.k.ntaps;
  for (int i = threadIdx.x; i < NT * 32; i += 256) {
    const int t = i >> 5, c = i & 31;
    sw[i] = p.w[c * NT + t];
  }
  __syncthreads();
  const int c4 = threadIdx.x & 7;
  const float bias = p.bias ? p.bias[0] : 0.f;
  const long stride = (long)gridDim.x * 32;
  for (long v = (long)blockIdx.x * 32 + (threadIdx.x >> 3); v < p.M; v += stride) {
    long t = v;
    const int x = t % p.g.W; t /= p.g.W;
    const int y = t % p.g.H; t /= p.g.H;
    const int d = t % p.g.D;
    const int b = t / p.g.D;
    const float* base = p.a + p.g.vox(b, d, y, x) * 32 + c4 * 4;
    float acc = 0.f;
    for (int tp = 0; tp < NT; ++tp) {
      const f32x4 q = *reinterpret_cast<const f32x4*>(base + (long)p.k.tap_off[tp] * 32);
      const float* ww = sw + tp * 32 + c4 * 4;
      acc += q.x * ww[0] + q.y * ww[1] + q.z * ww[2] + q.w * ww[3];
    }
    // the 8 lanes of a voxel are always active together: wavefront-shuffle reduction
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (c4 == 0) {
      float r = acc + bias;
      if (p.add_src) r += p.add_src[v];
      if (p.relu) r = r > 0.f ? r : 0.f;
      p.out[v] = r;
    }
  }
}

// g_a[v][c] = sum_t g_out[v - off(t)] * w[c][t]   (only where v - off(t) is a real voxel)
struct OutConvBwdArgs {
  const float* g_out;
  const float* a;
  const float* w;
  float* g_a;
  float* partial;   // [blocks][ntaps*32 + 1]
  PclDev g;
  long M;
  OutGeom k;
};

__global__ __launch_bounds__(256) void conv32to1_dgrad_kernel(OutConvBwdArgs p) {
  __shared__ float sw[27 * 32];
  const int NT = p.k.ntaps;
  for (int i = threadIdx.x; i < NT * 32; i += 256) {
    const int t = i >> 5, c = i & 31;
    sw[i] = p.w[c * NT + t];
  }
  __syncthreads();
  const int c4 = threadIdx.x & 7;
  const int D = p.g.D, H = p.g.H, W = p.g.W;
  const long stride = (long)gridDim.x * 32;
  for (long v = (long)blockIdx.x * 32 + (threadIdx.x >> 3); v < p.M; v += stride) {
    long t = v;
    const int x = t % W; t /= W;
    const int y = t % H; t /= H;
    const int d = t % D;
    const int b = t / D;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int tp = 0;
    for (int kd = 0; kd < p.k.kd; ++kd)
      for (int kh = 0; kh < p.k.kh; ++kh)
        for (int kw = 0; kw < p.k.kw; ++kw, ++tp) {
          // output voxel u with u + offset(tap) = v
          const int ud = d - ((p.k.kd > 1 ? kd * p.k.dil : 0) - p.k.pad_d);
          const int uy = y - (kh * p.k.dil - p.k.pad_h), ux = x - (kw * p.k.dil - p.k.pad_w);
          if (ud < 0 || ud >= D || uy < 0 || uy >= H || ux < 0 || ux >= W) continue;
          const float gl = p.g_out[(((long)b * D + ud) * H + uy) * W + ux];
          const float* ww = sw + tp * 32 + c4 * 4;
          acc.x += gl * ww[0]; acc.y += gl * ww[1]; acc.z += gl * ww[2]; acc.w += gl * ww[3];
        }
    *reinterpret_cast<f32x4*>(p.g_a + p.g.vox(b, d, y, x) * 32 + c4 * 4) = acc;
  }
}

// g_w[c][t] = sum_v g_out[v] * a[v + off(t)][c]  =  sum_u a[u][c] * g_out[u - off(t)];  g_bias = sum_v g_out[v].
// Input-centric form: every activation voxel u (a 128-byte line, 8 lanes x float4) is read ONCE and multiplied
// by the NT gradient values that reach it; g_out is a 1-channel map (7 MB at 4 pairs) that stays in L2.  The
// output-centric form re-read the 238 MB activation for every tap.
template <int NT>
__global__ __launch_bounds__(256) void conv32to1_wgrad_kernel(OutConvBwdArgs p) {
  __shared__ float red[32][33];
  const int c4 = threadIdx.x & 7, vl = threadIdx.x >> 3;
  const int D = p.g.D, H = p.g.H, W = p.g.W;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float gsum = 0.f;
  const long stride = (long)gridDim.x * 32;
  for (long v = (long)blockIdx.x * 32 + vl; v < p.M; v += stride) {
    long t = v;
    const int x = t % W; t /= W;
    const int y = t % H; t /= H;
    const int d = t % D;
    const int b = t / D;
    gsum += p.g_out[v];
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(p.a + p.g.vox(b, d, y, x) * 32 + c4 * 4);
    const float* gb = p.g_out + (long)b * D * H * W;
    int tp = 0;
#pragma unroll
    for (int kd = 0; kd < (NT == 27 ? 3 : 1); ++kd)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw, ++tp) {
          // output voxel w = u - offset(tap)
          const int wd = d - ((NT == 27 ? kd * p.k.dil : 0) - p.k.pad_d);
          const int wy = y - (kh * p.k.dil - p.k.pad_h), wx = x - (kw * p.k.dil - p.k.pad_w);
          const bool ok = wd >= 0 && wd < D && wy >= 0 && wy < H && wx >= 0 && wx < W;
          const float gl = ok ? gb[((long)wd * H + wy) * W + wx] : 0.f;
          acc[tp] += gl * a4;
        }
  }
  float* out = p.partial + (long)blockIdx.x * (NT * 32 + 1);
#pragma unroll
  for (int tp = 0; tp < NT; ++tp) {
    __syncthreads();
    red[vl][c4 * 4 + 0] = acc[tp].x; red[vl][c4 * 4 + 1] = acc[tp].y;
    red[vl][c4 * 4 + 2] = acc[tp].z; red[vl][c4 * 4 + 3] = acc[tp].w;
    __syncthreads();
    if (threadIdx.x < 32) {
      float s = 0.f;
      for (int j = 0; j < 32; ++j) s += red[j][threadIdx.x];
      out[tp * 32 + threadIdx.x] = s;
    }
  }
  __syncthreads();
  if (c4 == 0) red[vl][0] = gsum;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int j = 0; j < 32; ++j) s += red[j][0];
    out[NT * 32] = s;
  }
}

// one wave per output value: lanes stride over the slabs, shuffle-tree sum (fixed order)
__global__ __launch_bounds__(256) void conv32to1_wgrad_reduce_kernel(const float* __restrict__ partial, int nblocks, int NT,
                                                                      float* __restrict__ g_w, float* __restrict__ g_bias,
                                                                      int accumulate) {
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (idx > NT * 32) return;
  double s = 0.0;
  for (int i = lane; i < nblocks; i += 64) s += (double)partial[(long)i * (NT * 32 + 1) + idx];
  s = wave_sum_d(s);
  if (lane != 0) return;
  if (idx == NT * 32) { if (g_bias) g_bias[0] = accumulate ? g_bias[0] + (float)s : (float)s; }
  else { const int t = idx >> 5, c = idx & 31; if (g_w) g_w[c * NT + t] = accumulate ? g_w[c * NT + t] + (float)s : (float)s; }
}

// =====================================================================================================
// 2-D 3x3 (dilation 1) fast paths of the thin convolutions: the refinement's output layer and the data
// gradient of its 4->32 input layer run on full-resolution maps (238 MB per tensor at 4 pairs), where the
// generic kernels above spend their time on per-voxel 64-bit index arithmetic and on re-reading every
// 128-byte activation line once per tap through L1.
//
// Forward (32->1): "project, then gather".  out[v] = sum_t sum_c a[v+off_t][c] w[c][t], so every input voxel
// u is read ONCE and projected onto the nine taps, P_t[u] = sum_c a[u][c] w[c][t] (288 FMAs, one lane per
// voxel: no cross-lane reduction), the nine planes of a 16x32 voxel patch go to LDS, and each output of the
// patch's 14x30 interior sums nine scalars.  HBM/L2 reads: 1.22x the activation (patch halo), L1: once.
#define TC_PR 16                 // patch rows (with the one-voxel halo)
#define TC_PC 32                 // patch columns
#define TC_OR (TC_PR - 2)
#define TC_OC (TC_PC - 2)

struct Thin2dFwdArgs {
  const float* a;        // PCL
  const float* w;        // [32][9]
  const float* bias;     // [1] or null
  const float* add_src;  // dense [B][H][W] or null
  float* out;            // dense [B][H][W]
  int relu;
  PclDev g;
  int tiles_x, tiles_y;
};

__global__ __launch_bounds__(256) void conv32to1_2d_fwd_kernel(Thin2dFwdArgs p) {
  __shared__ float sp[9][TC_PR][TC_PC + 1];
  __shared__ float sw[32 * 9];
  for (int i = threadIdx.x; i < 288; i += 256) sw[i] = p.w[i];
  int t = blockIdx.x;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int b = t / p.tiles_y;
  const int y0 = ty * TC_OR, x0 = tx * TC_OC;          // first output of the patch
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int idx = threadIdx.x + 256 * pass;
    const int r = idx >> 5, c = idx & 31;
    const int yi = y0 - 1 + r, xi = x0 - 1 + c;         // image coordinates of the staged voxel (-1 .. H, -1 .. W)
    float acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.f;
    if (yi <= p.g.H && xi <= p.g.W) {                   // inside the PCL (halo row/column H, W included); others unused
      const float* src = p.a + (((long)b * p.g.Hp + (yi + p.g.ph)) * p.g.Wp + (xi + p.g.pw)) * 32;
      f32x4 q[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) q[j] = *reinterpret_cast<const f32x4*>(src + 4 * j);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          acc[k] = fmaf(q[j].x, sw[(4 * j + 0) * 9 + k], acc[k]);
          acc[k] = fmaf(q[j].y, sw[(4 * j + 1) * 9 + k], acc[k]);
          acc[k] = fmaf(q[j].z, sw[(4 * j + 2) * 9 + k], acc[k]);
          acc[k] = fmaf(q[j].w, sw[(4 * j + 3) * 9 + k], acc[k]);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) sp[k][r][c] = acc[k];
  }
  __syncthreads();
  const float bias = p.bias ? p.bias[0] : 0.f;
  for (int o = threadIdx.x; o < TC_OR * TC_OC; o += 256) {
    const int ry = o / TC_OC, cx = o - ry * TC_OC;
    const int y = y0 + ry, x = x0 + cx;
    if (y >= p.g.H || x >= p.g.W) continue;
    float s = bias;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) s += sp[ky * 3 + kx][ry + ky][cx + kx];
    const long v = ((long)b * p.g.H + y) * p.g.W + x;
    if (p.add_src) s += p.add_src[v];
    if (p.relu) s = s > 0.f ? s : 0.f;
    p.out[v] = s;
  }
}

// Backward of the same layer over 128-voxel row chunks (thread t owns float4 t, t+256, .. of the chunk: always
// channel group t&7), with the three rows of the 1-channel gradient map that a chunk needs staged in LDS
// (zero outside the image): no per-voxel index arithmetic, no bounds tests in the tap loops.
struct Thin2dBwdArgs {
  const float* g_out;    // dense [B][H][W]
  const float* a;        // PCL (wgrad)
  const float* w;        // [32][9] (dgrad)
  float* g_a;            // PCL (dgrad)
  float* partial;        // [blocks][289] (wgrad)
  PclDev g;
  int nchunks, chunks_per_row;
  // BNSUMS flavour of the data gradient: stage 1 of the BatchNorm backward whose output gradient g_a is (the last
  // refinement block's), from the values in registers: sum g_y and sum g_y*(z - mean) per channel, one fp64 slab [64] per
  // workgroup in the layout bn_bwd_finalize reads
  const float* bn_z; const float* bn_scale; const float* bn_shift; const float* bn_mean; float slope;
  double* bn_partial;
};

__device__ inline void thin_stage_g(const Thin2dBwdArgs& p, float (*sg)[132], int b, int y, int x0) {
  // sg[r][i] = g_out[b][y-1+r][x0-1+i], i = 0..129
  for (int i = threadIdx.x; i < 3 * 130; i += 256) {
    const int r = i / 130, c = i - r * 130;
    const int yy = y - 1 + r, xx = x0 - 1 + c;
    sg[r][c] = (yy >= 0 && yy < p.g.H && xx >= 0 && xx < p.g.W) ? p.g_out[((long)b * p.g.H + yy) * p.g.W + xx] : 0.f;
  }
}

// g_a[v][c] = sum_t g_out[v - off_t] w[c][t]
// BNSUMS: g_a is the output gradient of the last refinement block; its BatchNorm backward starts with per-channel sums over
// g_a and that block's pre-activation z (a pass of its own: read g_a, read z, 89 us at 4 pairs) — taken here from the
// registers that hold g_a anyway, at the price of the z read.
template <bool BNSUMS>
__global__ __launch_bounds__(256) void conv32to1_2d_dgrad_kernel(Thin2dBwdArgs p) {
  __shared__ float sg[3][132];
  __shared__ float red[2][32][33];
  const int c4 = threadIdx.x & 7, vl = threadIdx.x >> 3;
  f32x4 wv[9];
#pragma unroll
  for (int k = 0; k < 9; ++k)
    wv[k] = (f32x4){p.w[(c4 * 4 + 0) * 9 + k], p.w[(c4 * 4 + 1) * 9 + k], p.w[(c4 * 4 + 2) * 9 + k], p.w[(c4 * 4 + 3) * 9 + k]};
  f32x4 bsc, bsh, bmu, s_dy = {0.f, 0.f, 0.f, 0.f}, s_dx = {0.f, 0.f, 0.f, 0.f};
  if (BNSUMS) {
    bsc = *reinterpret_cast<const f32x4*>(p.bn_scale + c4 * 4); bsh = *reinterpret_cast<const f32x4*>(p.bn_shift + c4 * 4);
    bmu = *reinterpret_cast<const f32x4*>(p.bn_mean + c4 * 4);
  }
  for (int ch = blockIdx.x; ch < p.nchunks; ch += gridDim.x) {
    const int rowi = ch / p.chunks_per_row, cx = ch - rowi * p.chunks_per_row;
    const int y = rowi % p.g.H, b = rowi / p.g.H, x0 = cx * 128;
    const int nf4 = min(128, p.g.W - x0) * 8;
    __syncthreads();
    thin_stage_g(p, sg, b, y, x0);
    __syncthreads();
    const long voff = (((long)b * p.g.Hp + (y + p.g.ph)) * p.g.Wp + (x0 + p.g.pw)) * 32;
    float* dst = p.g_a + voff;
    f32x4 zz[4];
    if (BNSUMS) {
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        const int f = threadIdx.x + 256 * k4;
        if (f < nf4) zz[k4] = *reinterpret_cast<const f32x4*>(p.bn_z + voff + f * 4);
      }
    }
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      const int f = threadIdx.x + 256 * k4;
      if (f < nf4) {
        const int vx = f >> 3;                         // voxel within the chunk
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // tap (ky,kx) reaches v from the output voxel v - off = (y - (ky-1), x - (kx-1))
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) acc += sg[2 - ky][vx + 2 - kx] * wv[ky * 3 + kx];
        *reinterpret_cast<f32x4*>(dst + f * 4) = acc;
        if (BNSUMS) {
          const f32x4 yy = zz[k4] * bsc + bsh;
          f32x4 gy;
          gy.x = yy.x > 0.f ? acc.x : acc.x * p.slope; gy.y = yy.y > 0.f ? acc.y : acc.y * p.slope;
          gy.z = yy.z > 0.f ? acc.z : acc.z * p.slope; gy.w = yy.w > 0.f ? acc.w : acc.w * p.slope;
          s_dy += gy; s_dx += gy * (zz[k4] - bmu);
        }
      }
    }
  }
  if (BNSUMS) {
    // 32 threads share a channel group: fixed-order sum through LDS, fp64 slab
    __syncthreads();
    red[0][vl][c4 * 4 + 0] = s_dy.x; red[0][vl][c4 * 4 + 1] = s_dy.y; red[0][vl][c4 * 4 + 2] = s_dy.z; red[0][vl][c4 * 4 + 3] = s_dy.w;
    red[1][vl][c4 * 4 + 0] = s_dx.x; red[1][vl][c4 * 4 + 1] = s_dx.y; red[1][vl][c4 * 4 + 2] = s_dx.z; red[1][vl][c4 * 4 + 3] = s_dx.w;
    __syncthreads();
    if (threadIdx.x < 64) {
      const int which = threadIdx.x >> 5, c = threadIdx.x & 31;
      double sum = 0.0;
      for (int j = 0; j < 32; ++j) sum += (double)red[which][j][c];
      p.bn_partial[(long)blockIdx.x * 64 + which * 32 + c] = sum;
    }
  }
}

// g_w[c][t] = sum_u a[u][c] g_out[u - off_t];  g_bias = sum g_out.  One partial slab per workgroup.
// Round 4: the next chunk's operands — four 16-byte activation loads and up to two gradient-map values per thread — are
// requested BEFORE the current chunk's 144 FMAs instead of behind the barrier that opens it (every chunk then paid a whole HBM
// round trip: 81 us for 245 MB at 4 pairs), and the staged gradient rows alternate between two LDS buffers: one barrier per chunk.
__global__ __launch_bounds__(256) void conv32to1_2d_wgrad_kernel(Thin2dBwdArgs p) {
  __shared__ float sg[2][3][132];
  __shared__ float red[32][33];
  const int c4 = threadIdx.x & 7, vl = threadIdx.x >> 3;
  f32x4 acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float gsum = 0.f;
  // this thread's staging jobs: i = tid, tid + 256 of the 3 x 130 gradient values; its four float4 of the activation chunk
  const int i0 = threadIdx.x, i1 = threadIdx.x + 256;
  const int r0 = i0 / 130, q0 = i0 - r0 * 130, r1 = i1 / 130, q1 = i1 - r1 * 130;
  auto request = [&](int ch, f32x4 (&a4)[4], float& g0, float& g1) {
    const int rowi = ch / p.chunks_per_row, cx = ch - rowi * p.chunks_per_row;
    const int y = rowi % p.g.H, b = rowi / p.g.H, x0 = cx * 128;
    const int nf4 = min(128, p.g.W - x0) * 8;
    const float* src = p.a + (((long)b * p.g.Hp + (y + p.g.ph)) * p.g.Wp + (x0 + p.g.pw)) * 32;
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      const int f = threadIdx.x + 256 * k4;
      a4[k4] = *reinterpret_cast<const f32x4*>(src + min(f, nf4 - 1) * 4);      // (beyond the chunk: re-read, ignored)
    }
    auto gval = [&](int r, int c) {
      const int yy = y - 1 + r, xx = x0 - 1 + c;
      const bool in = yy >= 0 && yy < p.g.H && xx >= 0 && xx < p.g.W;
      const float v = p.g_out[((long)b * p.g.H + min(max(yy, 0), p.g.H - 1)) * p.g.W + min(max(xx, 0), p.g.W - 1)];
      return in ? v : 0.f;
    };
    g0 = gval(r0, q0);
    g1 = i1 < 3 * 130 ? gval(r1, q1) : 0.f;
  };
  f32x4 a4[2][4];
  float g0[2], g1[2];
  int ch = blockIdx.x, it = 0;
  if (ch < p.nchunks) request(ch, a4[0], g0[0], g1[0]);
  for (; ch < p.nchunks; ch += gridDim.x, it ^= 1) {
    const int rowi = ch / p.chunks_per_row, cx = ch - rowi * p.chunks_per_row;
    const int x0 = cx * 128;
    const int nf4 = min(128, p.g.W - x0) * 8;
    float (*sgc)[132] = sg[it];
    sgc[r0][q0] = g0[it];
    if (i1 < 3 * 130) sgc[r1][q1] = g1[it];
    __syncthreads();                                       // (the other buffer's readers passed this barrier a chunk ago)
    if (ch + (int)gridDim.x < p.nchunks) {
      if (it == 0) request(ch + gridDim.x, a4[1], g0[1], g1[1]);
      else request(ch + gridDim.x, a4[0], g0[0], g1[0]);
    }
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      const int f = threadIdx.x + 256 * k4;
      if (f < nf4) {
        const int vx = f >> 3;
        const f32x4 av = it == 0 ? a4[0][k4] : a4[1][k4];
        if (c4 == 0) gsum += sgc[1][vx + 1];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] += sgc[2 - ky][vx + 2 - kx] * av;
      }
    }
  }
  float* out = p.partial + (long)blockIdx.x * (9 * 32 + 1);
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) {
    __syncthreads();
    red[vl][c4 * 4 + 0] = acc[tp].x; red[vl][c4 * 4 + 1] = acc[tp].y;
    red[vl][c4 * 4 + 2] = acc[tp].z; red[vl][c4 * 4 + 3] = acc[tp].w;
    __syncthreads();
    if (threadIdx.x < 32) {
      float s = 0.f;
      for (int j = 0; j < 32; ++j) s += red[j][threadIdx.x];
      out[tp * 32 + threadIdx.x] = s;
    }
  }
  __syncthreads();
  if (c4 == 0) red[vl][0] = gsum;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int j = 0; j < 32; ++j) s += red[j][0];
    out[9 * 32] = s;
  }
}

static bool thin2d_applicable(const as_pcl* g, const as_conv_shape* s) {
  return g->D == 1 && s->kd == 1 && s->kh == 3 && s->kw == 3 && s->dil == 1 && s->stride == 1 && s->pad_h == 1 &&
         s->pad_w == 1 && g->ph >= 1 && g->pw >= 1 && (long)g->B * g->H * ((g->W + 127) / 128) < (1L << 31);
}

// ---- soft-argmax ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void softargmax_fwd_kernel(const float* __restrict__ logits, int B, int D, long HW,
                                                              float* __restrict__ pred, int32_t* __restrict__ argmax,
                                                              float* __restrict__ fcs) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * HW) return;
  const long b = i / HW, pix = i % HW;
  const float* l = logits + b * D * HW + pix;
  float m1 = -INFINITY, m2 = -INFINITY, sum = 0.f;
  int am = 0;
  for (int d = 0; d < D; ++d) {
    const float v = l[d * HW];
    sum += v;
    if (v > m1) { m2 = m1; m1 = v; am = d; }
    else if (v > m2) { m2 = v; }
  }
  float se = 0.f;
  for (int d = 0; d < D; ++d) se += expf(l[d * HW] - m1);
  float acc = 0.f;
  for (int d = 0; d < D; ++d) acc += (expf(l[d * HW] - m1) / se) * (float)d;
  pred[i] = acc;
  if (argmax) argmax[i] = am;
  if (fcs) fcs[i] = (D > 2) ? m1 - (sum - m1 - m2) / (float)(D - 2) : 0.f;
}

__global__ __launch_bounds__(256) void softargmax_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ g_pred,
                                                              const float* __restrict__ g_in, int B, int D, long HW,
                                                              float* __restrict__ g_logits) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * HW) return;
  const long b = i / HW, pix = i % HW;
  const float* l = logits + b * D * HW + pix;
  float m1 = -INFINITY;
  for (int d = 0; d < D; ++d) m1 = fmaxf(m1, l[d * HW]);
  float se = 0.f;
  for (int d = 0; d < D; ++d) se += expf(l[d * HW] - m1);
  float pr = 0.f;
  for (int d = 0; d < D; ++d) pr += (expf(l[d * HW] - m1) / se) * (float)d;
  const float gp = g_pred ? g_pred[i] : 0.f;
  for (int d = 0; d < D; ++d) {
    const float pd = expf(l[d * HW] - m1) / se;
    float gv = pd * ((float)d - pr) * gp;
    if (g_in) gv += g_in[b * D * HW + d * HW + pix];
    g_logits[b * D * HW + d * HW + pix] = gv;
  }
}

// ---- host ------------------------------------------------------------------------------
static int fill_out_geom(const as_pcl* g, const as_conv_shape* s, OutGeom* k, const char* who) {
  AS_CHECK_ARG(as_pcl_ok(g) && s, "%s: bad geometry", who);
  const int T = s->kd * s->kh * s->kw;
  AS_CHECK_ARG((T == 27 && s->kd == 3) || (T == 9 && s->kd == 1), "%s: only 3x3x3 and 3x3 kernels are built", who);
  AS_CHECK_ARG(s->stride == 1 && s->dil >= 1, "%s: stride must be 1", who);
  AS_CHECK_ARG(s->pad_h <= g->ph && s->pad_w <= g->pw && s->dil * (s->kh - 1) - s->pad_h <= g->ph &&
               s->dil * (s->kw - 1) - s->pad_w <= g->pw && (s->kd == 1 || (s->pad_d <= g->pd && s->dil * 2 - s->pad_d <= g->pd)),
               "%s: halo too small", who);
  AS_CHECK_ARG(2 * s->pad_h == s->dil * (s->kh - 1) && 2 * s->pad_w == s->dil * (s->kw - 1) &&
               (s->kd == 1 || 2 * s->pad_d == s->dil * (s->kd - 1)), "%s: only 'same' padding is built", who);
  k->kd = s->kd; k->kh = s->kh; k->kw = s->kw; k->pad_d = s->kd > 1 ? s->pad_d : 0; k->pad_h = s->pad_h; k->pad_w = s->pad_w;
  k->dil = s->dil; k->ntaps = T;
  const int Hp = g->H + 2 * g->ph, Wp = g->W + 2 * g->pw;
  int n = 0;
  for (int i = 0; i < s->kd; ++i) for (int j = 0; j < s->kh; ++j) for (int l = 0; l < s->kw; ++l)
    k->tap_off[n++] = (((s->kd > 1 ? i * s->dil : 0) - k->pad_d) * Hp + (j * s->dil - s->pad_h)) * Wp + (l * s->dil - s->pad_w);
  return AS_OK;
}

static inline int out_blocks(long M) {
  long nb = (M + 31) / 32;
  if (nb > 4096) nb = 4096;
  return (int)nb;
}
#ifndef OUTW_BLOCKS
#define OUTW_BLOCKS 1024                 // one round of four workgroups per CU; 2048 (two rounds): 70.7 + 9.6 us (kernel + slab reduction) against 69.2 + 6.8
#endif

// slab reduction of a 32->1 weight gradient: partial [nblocks][ntaps * 32 + 1] -> g_w [32][ntaps], g_bias [1] (agg_tail_bwd.hip too)
int as_conv32to1_wgrad_reduce(const float* partial, int nblocks, int ntaps, float* g_w, float* g_bias, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(conv32to1_wgrad_reduce_kernel, dim3(as_div_up(ntaps * 32 + 1, 4)), dim3(256), 0, st, partial, nblocks, ntaps,
                     g_w, g_bias, accumulate);
  AS_CHECK_LAUNCH("conv32to1 weight gradient (slab reduction)");
  return AS_OK;
}

extern "C" int as_conv32to1_fwd(const float* a, const as_pcl* g, const as_conv_shape* s, const float* w,
                                const float* bias, const float* add_src, int relu, float* out, void* stream) {
  OutConvArgs p;
  if (int e = fill_out_geom(g, s, &p.k, "as_conv32to1_fwd")) return e;
  AS_CHECK_ARG(a && w && out, "as_conv32to1_fwd: null pointer");
  if (thin2d_applicable(g, s)) {
    Thin2dFwdArgs q;
    q.a = a; q.w = w; q.bias = bias; q.add_src = add_src; q.out = out; q.relu = relu; q.g = as_make_dev(g);
    q.tiles_x = (g->W + TC_OC - 1) / TC_OC; q.tiles_y = (g->H + TC_OR - 1) / TC_OR;
    hipLaunchKernelGGL(conv32to1_2d_fwd_kernel, dim3(g->B * q.tiles_x * q.tiles_y), dim3(256), 0, (hipStream_t)stream, q);
    AS_CHECK_LAUNCH("as_conv32to1_fwd(2d)");
    return AS_OK;
  }
  p.a = a; p.w = w; p.bias = bias; p.add_src = add_src; p.out = out; p.relu = relu; p.g = as_make_dev(g);
  p.M = (long)g->B * g->D * g->H * g->W;
  hipLaunchKernelGGL(conv32to1_fwd_kernel, dim3(out_blocks(p.M)), dim3(256), 0, (hipStream_t)stream, p);
  AS_CHECK_LAUNCH("as_conv32to1_fwd");
  return AS_OK;
}

extern "C" int64_t as_conv32to1_bwd_workspace(const as_pcl* g, const as_conv_shape* s) {
  if (!as_pcl_ok(g) || !s) return -1;
  return (int64_t)OUTW_BLOCKS * (s->kd * s->kh * s->kw * 32 + 1);
}

extern "C" int as_conv32to1_bwd(const float* g_out, const float* a, const as_pcl* g, const as_conv_shape* s,
                                const float* w, float* g_a, float* g_w, float* g_bias, int accumulate, float* workspace,
                                void* stream) {
  OutConvBwdArgs p;
  if (int e = fill_out_geom(g, s, &p.k, "as_conv32to1_bwd")) return e;
  AS_CHECK_ARG(g_out && a && w && workspace, "as_conv32to1_bwd: null pointer");
  p.g_out = g_out; p.a = a; p.w = w; p.g_a = g_a; p.partial = workspace; p.g = as_make_dev(g);
  p.M = (long)g->B * g->D * g->H * g->W;
  hipStream_t st = (hipStream_t)stream;
  if (thin2d_applicable(g, s)) {
    Thin2dBwdArgs q;
    q.g_out = g_out; q.a = a; q.w = w; q.g_a = g_a; q.partial = workspace; q.g = as_make_dev(g);
    q.chunks_per_row = (g->W + 127) / 128; q.nchunks = g->B * g->H * q.chunks_per_row;
    if (g_a) {
      hipLaunchKernelGGL(conv32to1_2d_dgrad_kernel<false>, dim3(q.nchunks > 16384 ? 16384 : q.nchunks), dim3(256), 0, st, q);
      AS_CHECK_LAUNCH("as_conv32to1_bwd(2d dgrad)");
    }
    if (g_w || g_bias) {
      int nb = (q.nchunks + 3) / 4;
      if (nb > OUTW_BLOCKS) nb = OUTW_BLOCKS;
      hipLaunchKernelGGL(conv32to1_2d_wgrad_kernel, dim3(nb), dim3(256), 0, st, q);
      AS_CHECK_LAUNCH("as_conv32to1_bwd(2d wgrad)");
      hipLaunchKernelGGL(conv32to1_wgrad_reduce_kernel, dim3(as_div_up(9 * 32 + 1, 4)), dim3(256), 0, st,
                         workspace, nb, 9, g_w, g_bias, accumulate);
      AS_CHECK_LAUNCH("as_conv32to1_bwd(reduce)");
    }
    return AS_OK;
  }
  if (g_a) {
    hipLaunchKernelGGL(conv32to1_dgrad_kernel, dim3(out_blocks(p.M)), dim3(256), 0, st, p);
    AS_CHECK_LAUNCH("as_conv32to1_bwd(dgrad)");
  }
  if (g_w || g_bias) {
    long nb = (p.M + 31) / 32;
    if (nb > 512) nb = 512;          // every workgroup ends with a 27-round LDS reduction: keep them few
    if (p.k.ntaps == 27) hipLaunchKernelGGL(conv32to1_wgrad_kernel<27>, dim3((int)nb), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(conv32to1_wgrad_kernel<9>, dim3((int)nb), dim3(256), 0, st, p);
    AS_CHECK_LAUNCH("as_conv32to1_bwd(wgrad)");
    hipLaunchKernelGGL(conv32to1_wgrad_reduce_kernel, dim3(as_div_up(p.k.ntaps * 32 + 1, 4)), dim3(256), 0, st,
                       workspace, (int)nb, p.k.ntaps, g_w, g_bias, accumulate);
    AS_CHECK_LAUNCH("as_conv32to1_bwd(reduce)");
  }
  return AS_OK;
}

// Data gradient of the 2-D 32->1 output layer with stage 1 of the BatchNorm backward that consumes it (see
// conv32to1_2d_dgrad_kernel<true>): bn_workspace receives as_conv32to1_bnsums_parts(g) fp64 slabs for as_bn_act_bwd_given.
#define THIN_BNSUMS_BLOCKS 1024
extern "C" int as_conv32to1_bnsums_ok(const as_pcl* g, const as_conv_shape* s) {
  if (!g || !s || !as_pcl_ok(g)) return AS_ERR_ARG;
  return thin2d_applicable(g, s) ? 1 : 0;
}
extern "C" int as_conv32to1_bnsums_parts(const as_pcl* g) {
  if (!g || !as_pcl_ok(g)) return AS_ERR_ARG;
  const long nch = (long)g->B * g->H * ((g->W + 127) / 128);
  return (int)(nch < THIN_BNSUMS_BLOCKS ? nch : THIN_BNSUMS_BLOCKS);
}
extern "C" int as_conv32to1_dgrad_bnsums(const float* g_out, const as_pcl* g, const as_conv_shape* s, const float* w, float* g_a,
                                         const float* bn_z, const float* bn_scale, const float* bn_shift, const float* bn_mean,
                                         float slope, float* bn_workspace, void* stream) {
  AS_CHECK_ARG(g && s && as_pcl_ok(g) && thin2d_applicable(g, s), "as_conv32to1_dgrad_bnsums: configuration not supported");
  AS_CHECK_ARG(g_out && w && g_a && bn_z && bn_scale && bn_shift && bn_mean && bn_workspace, "as_conv32to1_dgrad_bnsums: null pointer");
  AS_CHECK_ARG(((uintptr_t)bn_workspace & 7) == 0, "as_conv32to1_dgrad_bnsums: workspace must be 8-byte aligned");
  Thin2dBwdArgs q;
  q.g_out = g_out; q.a = nullptr; q.w = w; q.g_a = g_a; q.partial = nullptr; q.g = as_make_dev(g);
  q.chunks_per_row = (g->W + 127) / 128; q.nchunks = g->B * g->H * q.chunks_per_row;
  q.bn_z = bn_z; q.bn_scale = bn_scale; q.bn_shift = bn_shift; q.bn_mean = bn_mean; q.slope = slope;
  q.bn_partial = reinterpret_cast<double*>(bn_workspace);
  hipLaunchKernelGGL(conv32to1_2d_dgrad_kernel<true>, dim3(as_conv32to1_bnsums_parts(g)), dim3(256), 0, (hipStream_t)stream, q);
  AS_CHECK_LAUNCH("as_conv32to1_dgrad_bnsums");
  return AS_OK;
}

// The 3-D entry points of a4 are the generic kernels with a 3x3x3 'same' shape.
static const as_conv_shape k333 = {3, 3, 3, 1, 1, 1, 1, 1};

extern "C" int as_conv3d_out_fwd(const float* a, const as_pcl* g, const float* w, const float* bias,
                                 float* logits, void* stream) {
  as_prof_mark(AS_PROF_OUTCONV_FWD, (hipStream_t)stream, 1, 0.0);
  const int e = as_conv32to1_fwd(a, g, &k333, w, bias, nullptr, 0, logits, stream);
  if (e == AS_OK && g) as_prof_mark(AS_PROF_OUTCONV_FWD, (hipStream_t)stream, 0, (128.0 + 4.0) * (double)g->B * g->D * g->H * g->W);
  return e;
}

extern "C" int64_t as_conv3d_out_bwd_workspace(const as_pcl* g) { return as_conv32to1_bwd_workspace(g, &k333); }

extern "C" int as_conv3d_out_bwd(const float* g_logits, const float* a, const as_pcl* g, const float* w,
                                 float* g_a, float* g_w, float* g_bias, int accumulate, float* workspace, void* stream) {
  AS_CHECK_ARG(g_a && g_w, "as_conv3d_out_bwd: null pointer");
  as_prof_mark(AS_PROF_OUTCONV_BWD, (hipStream_t)stream, 1, 0.0);
  const int e = as_conv32to1_bwd(g_logits, a, g, &k333, w, g_a, g_w, g_bias, accumulate, workspace, stream);
  // data gradient: logits gradient read, g_a written; weight gradient: a and the logits gradient read
  if (e == AS_OK && g) as_prof_mark(AS_PROF_OUTCONV_BWD, (hipStream_t)stream, 0, (2 * 128.0 + 2 * 4.0) * (double)g->B * g->D * g->H * g->W);
  return e;
}

extern "C" int as_softargmax_fwd(const float* logits, int B, int D, int H, int W,
                                 float* pred, int32_t* argmax, float* fcs, void* stream) {
  AS_CHECK_ARG(logits && pred, "as_softargmax_fwd: null pointer");
  AS_CHECK_ARG(B > 0 && D > 0 && H > 0 && W > 0, "as_softargmax_fwd: bad shape");
  const long n = (long)B * H * W;
  as_prof_mark(AS_PROF_SOFTARGMAX_FWD, (hipStream_t)stream, 1, 0.0);
  hipLaunchKernelGGL(softargmax_fwd_kernel, dim3(as_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     logits, B, D, (long)H * W, pred, argmax, fcs);
  as_prof_mark(AS_PROF_SOFTARGMAX_FWD, (hipStream_t)stream, 0, 4.0 * (double)n * (D + 3));
  AS_CHECK_LAUNCH("as_softargmax_fwd");
  return AS_OK;
}

extern "C" int as_softargmax_bwd(const float* logits, const float* g_pred, const float* g_logits_in,
                                 int B, int D, int H, int W, float* g_logits, void* stream) {
  AS_CHECK_ARG(logits && g_logits, "as_softargmax_bwd: null pointer");
  AS_CHECK_ARG(B > 0 && D > 0 && H > 0 && W > 0, "as_softargmax_bwd: bad shape");
  const long n = (long)B * H * W;
  as_prof_mark(AS_PROF_SOFTARGMAX_BWD, (hipStream_t)stream, 1, 0.0);
  hipLaunchKernelGGL(softargmax_bwd_kernel, dim3(as_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     logits, g_pred, g_logits_in, B, D, (long)H * W, g_logits);
  as_prof_mark(AS_PROF_SOFTARGMAX_BWD, (hipStream_t)stream, 0, 4.0 * (double)n * (2 * D + 1 + (g_logits_in ? D : 0)));
  AS_CHECK_LAUNCH("as_softargmax_bwd");
  return AS_OK;
}
