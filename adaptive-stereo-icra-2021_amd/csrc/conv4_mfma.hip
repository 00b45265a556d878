// Convolutions with a thin input (Cin <= 4) and 32 output channels on the fp32 matrix cores:
//   EdgeAwareRefinement.conv2d_feature = nn.Conv2d(4, 32, 3, padding=1) over cat([disparity, rgb])
//     (adaptive_stereo/models/stereo_net.py:89-94, 116-118)
//   FeatureExtractorNetwork.downsample[0] = nn.Conv2d(3, 32, 5, stride=2, padding=2)  (:61-69)
//
// Input layout "PCL4": float buf[B][H+2*ph][W+2*pw][4] with a zero halo — one 16-byte pixel.
// GEMM view: Z[pix][co] = sum_{tap} sum_{c<4} X4[pix*stride + off(tap)][c] * W[co][c][tap],
// K = 4*taps (36 or 100).  Per tap a lane loads its pixel's float4 once and issues two
// 32x32x2 MFMAs (k = lane>>5 picks channel 2j+h).  These layers are HBM-bound (16 B in, 128 B out
// per pixel); the matrix core is used because the epilogue (BatchNorm partials, fused affine) and
// the output tile layout are then shared with conv32.
#include "as_common.h"
#include "conv_epilogue.h"

struct Conv4Args {
  const float* x4;
  const float* wp;        // [tap][j][h][co]
  EpilogueArgs ep;
  PclDev gin, gout;       // gin describes the PCL4 tensor (4 floats per pixel)
  int M, stride, ntaps;
  int tap_off[AS_MAX_TAPS];
};

// NT = number of taps at compile time (0: run-time loop).  With a run-time tap loop every tap was a dependent round trip
// (load the pixel, two MFMAs, next tap): 59 us at 4 pairs for the feature extractor's 5x5 stride-2 input layer, 90 MB of
// traffic.  Unrolled, all of a tile's pixel and weight loads are in flight before the first MFMA.
template <int NT>
__global__ __launch_bounds__(256) void conv4_fwd_kernel(Conv4Args p) {
  __shared__ float red[4][32];
  __shared__ float bmean[32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, li = lane & 31;
  const int v = (blockIdx.x * 4 + wave) * 32 + li;
  const bool valid = v < p.M;
  int in_vox, out_vox;
  ConvMap m; m.Hl = p.gout.H; m.Wl = p.gout.W; m.in_stride = p.stride; m.out_stride = 1; m.out_oy = 0; m.out_ox = 0;
  conv_decode(valid ? v : p.M - 1, p.gin, p.gout, m, in_vox, out_vox);
  const float* xa = p.x4 + (long)in_vox * 4;
  f32x16 acc;
  conv_init_acc(acc, p.ep.bias, li);
  if (NT > 0) {
    f32x4 q[NT > 0 ? NT : 1];
    float b0[NT > 0 ? NT : 1], b1[NT > 0 ? NT : 1];
#pragma unroll
    for (int tp = 0; tp < NT; ++tp) {
#ifdef C4F_EXP_NOLOAD                                   // diagnostic builds (tests/tools/exp_step.sh): results are wrong
      q[tp] = (f32x4){(float)p.tap_off[tp], (float)lane, 1.f, 2.f};
      b0[tp] = (float)(tp + lane); b1[tp] = (float)(tp - lane);
#else
      q[tp] = *reinterpret_cast<const f32x4*>(xa + (long)p.tap_off[tp] * 4);
      b0[tp] = p.wp[(tp * 2 + 0) * 64 + lane];
      b1[tp] = p.wp[(tp * 2 + 1) * 64 + lane];
#endif
    }
#pragma unroll
    for (int tp = 0; tp < NT; ++tp) {
      const float a0 = h ? q[tp].y : q[tp].x;
      const float a1 = h ? q[tp].w : q[tp].z;
#ifdef C4F_EXP_NOMFMA
      acc[0] += a0 * b0[tp]; acc[1] += a1 * b1[tp];
#else
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[tp], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[tp], acc, 0, 0, 0);
#endif
    }
  } else {
    for (int tp = 0; tp < p.ntaps; ++tp) {
      const f32x4 q = *reinterpret_cast<const f32x4*>(xa + (long)p.tap_off[tp] * 4);
      const float b0 = p.wp[(tp * 2 + 0) * 64 + lane];
      const float b1 = p.wp[(tp * 2 + 1) * 64 + lane];
      const float a0 = h ? q.y : q.x;
      const float a1 = h ? q.w : q.z;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc, 0, 0, 0);
    }
  }
  TileStats ts;
  conv_epilogue(acc, p.ep, out_vox, valid, min(128, p.M - (int)blockIdx.x * 128), red, bmean, &ts);
  stats_write(p.ep, blockIdx.x, ts);
}


// ---- 5x5 stride-2 instance on staged rows (the feature towers' first layer, downsample[0]: stereo_net.py:61-69) ----------
// conv4_fwd_kernel<25> gives a wave ONE tile: 25 sixteen-byte gathers per lane (stride 32 B: half of every line fetched twice
// over), 50 weight loads, 50 MFMAs, 16 stores, exit.  Diagnostic builds inside a step (tests/tools/exp_step.sh, round 4): 93 us as
// built, 71 without the loads, 56 without the MFMAs, 25 without both — loads and matrix work do not overlap (25 + 45 + 31 = 101).
// Here a wave is persistent (its 50 weight fragments stay in registers) and fetches ROWS: per tile and ky the 72 pixels
// 2 x0 - 2 .. 2 x0 + 69 of input row 2y + ky - 2 as one full and one eight-lane 16-byte load (10 instead of 25 load instructions),
// written into the wave's private LDS split by column parity with the channel order (c0, c2, c1, c3): tap kx reads entry
// li + kx/2 of half kx & 1 as ONE ds_read_b64 = (c_h, c_2+h), the two k-slices of its two MFMAs.  The next tile's rows are
// requested before this tile's matrix phase (the registers are free once the rows sit in LDS).  No barrier; taps and channels
// in conv4_fwd_kernel<25>'s order: bit-identical outputs.  Pixels past a row's padded end are clamped (they feed unstored outputs).
typedef float c4_f32x2 __attribute__((ext_vector_type(2)));
#define C4S_HALF (36 * 16)                    // bytes of one parity half of a staged row: entries 0..35
#define C4S_ROW (2 * C4S_HALF)                // 1,152
#define C4S_WAVE (5 * C4S_ROW)                // 5,760 B per wave

struct Conv4S2Args {
  const float* x4;
  const float* wp;        // [tap][j][h][co] (conv4_pack_kernel)
  const float* bias;
  float* z;
  PclDev gin, gout;
  int nseg, ntiles;
};

__global__ __launch_bounds__(256, 2) void conv4_s2_fwd_kernel(Conv4S2Args p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * C4S_WAVE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, li = lane & 31;
  char* buf = smem + wave * C4S_WAVE;
  float b0[25], b1[25];
#pragma unroll
  for (int tp = 0; tp < 25; ++tp) { b0[tp] = p.wp[(tp * 2 + 0) * 64 + lane]; b1[tp] = p.wp[(tp * 2 + 1) * 64 + lane]; }
  const float bias_v = p.bias ? p.bias[li] : 0.f;
  // staging: load A brings pixel `lane`, load B pixel 64 + (lane & 7); LDS slot: half v & 1, entry v >> 1
  const int va = lane, vb = 64 + (lane & 7);
  char* dst_a = buf + (va & 1) * C4S_HALF + (va >> 1) * 16;
  char* dst_b = buf + (vb & 1) * C4S_HALF + (vb >> 1) * 16;
  const char* rd = buf + li * 16 + 8 * h;                  // + row * C4S_ROW + (kx & 1) * C4S_HALF + (kx >> 1) * 16
  const int nwaves = gridDim.x * 4;
  int tile = blockIdx.x * 4 + wave;
  f32x4 ra[5], rb[5];
  auto request = [&](int t) {
    const int seg = t % p.nseg, row = t / p.nseg;
    const int y = row % p.gout.H, b = row / p.gout.H;
    const int c0 = 2 * seg * 32 - 2 + p.gin.pw;           // first staged column (padded coordinates)
    const int vlim = p.gin.Wp - 1 - c0;
    const float* src = p.x4 + (((long)b * p.gin.Hp + (2 * y - 2 + p.gin.ph)) * p.gin.Wp + c0) * 4;
    const int oa = min(va, vlim) * 4, ob = min(vb, vlim) * 4;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
      ra[ky] = *reinterpret_cast<const f32x4*>(src + (long)ky * p.gin.Wp * 4 + oa);
      rb[ky] = *reinterpret_cast<const f32x4*>(src + (long)ky * p.gin.Wp * 4 + ob);
    }
  };
  if (tile < p.ntiles) request(tile);
  for (; tile < p.ntiles; tile += nwaves) {
    const int seg = tile % p.nseg, row = tile / p.nseg;
    const int y = row % p.gout.H, b = row / p.gout.H;
    const int x0 = seg * 32;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
      *reinterpret_cast<f32x4*>(dst_a + ky * C4S_ROW) = (f32x4){ra[ky].x, ra[ky].z, ra[ky].y, ra[ky].w};
      *reinterpret_cast<f32x4*>(dst_b + ky * C4S_ROW) = (f32x4){rb[ky].x, rb[ky].z, rb[ky].y, rb[ky].w};
    }
    if (tile + nwaves < p.ntiles) request(tile + nwaves);  // in flight under this tile's matrix phase and stores
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias_v;
#pragma unroll
    for (int tp = 0; tp < 25; ++tp) {
      const int ky = tp / 5, kx = tp % 5;
      const c4_f32x2 a = *reinterpret_cast<const c4_f32x2*>(rd + ky * C4S_ROW + (kx & 1) * C4S_HALF + (kx >> 1) * 16);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0[tp], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1[tp], acc, 0, 0, 0);
    }
    float* outp = p.z + ((((long)b * p.gout.Hp + (y + p.gout.ph)) * p.gout.Wp + (x0 + p.gout.pw)) * 32) + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int xo = (r & 3) + 8 * (r >> 2) + 4 * h;
      if (x0 + xo < p.gout.W) outp[xo * 32] = acc[r];
    }
  }
}

static bool conv4_s2_applicable(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s) {
  if (s->kh != 5 || s->kw != 5 || s->stride != 2 || s->dil != 1 || s->pad_h != 2 || s->pad_w != 2) return false;
  if (gin->ph < 2 || gin->pw < 2) return false;            // rows 2y-2 .. 2y+2 and the left edge; the right edge is clamped
  return (long)gout->B * gout->H * ((gout->W + 31) / 32) >= 4096;
}

static bool g_conv4_s2 = true;
extern "C" int as_conv4_s2_enable(int on) {                // 0 / 1; anything else only reads.  Returns the previous setting.
  const int prev = g_conv4_s2 ? 1 : 0;
  if (on == 0 || on == 1) g_conv4_s2 = on == 1;
  return prev;
}

// ---- 3x3 stride-1 instance over row segments (the refinement's 4->32 input layer, full resolution) -------------
// Same tile discipline as conv32_lds.hip: a wave owns 32 consecutive voxels of a row, the last segment of a row
// is shifted left to end at W (duplicates are stored twice, skipped in the moments), vector-memory accesses are
// scalar base + constant lane offset + immediate (inline asm, counted waits), BatchNorm moments are per-lane
// shifted sums, one partial per workgroup.  Waves are independent: no barrier in the tile loop.
// Input: the three 34-pixel rows a wave tile needs are fetched as three 1-KB LDS-DMA instructions (64 pixels
// each) into the wave's private, double-buffered LDS rows; pending data therefore never sits in registers the
// compiler might copy.  A row's window is clamped (scalar) to stay inside the padded row and its LDS
// destination shifted by the same amount, so pixel xw-1 always lands at slot C4_SLOT0.
// A operand: lane (voxel li, k-half h) reads channels h and 2+h of pixel li+kx: ds_read_b32 at a constant
// lane address + immediates (no select instructions).
// Per tile a wave issues exactly 3 DMAs and 16 stores; the next tile's DMAs go out before this tile's MFMAs,
// so "s_waitcnt vmcnt(19)" = this tile's rows have landed (vmcnt is in-order; the 16 stores of the previous
// tile and the 3 DMAs of the next one are the only younger operations).
#define C4_SLOT0 30
#define C4_ROW_BYTES 1536            // (30 + 64) pixels x 16 B, padded
#define C4_WAVE_BYTES (2 * 3 * C4_ROW_BYTES)

struct Conv4RowsArgs {
  const float* x4;
  const float* wp;        // [9][2][64]
  EpilogueArgs ep;
  PclDev gin, gout;       // gin: PCL4
  int wtiles_per_row, nwtiles;
};

__device__ inline void c4_dma(const float* sbase, unsigned voff, unsigned m0) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               :: "s"(m0), "v"(voff), "s"(sbase) : "memory", "m0");
}
template <int IMM> __device__ inline void c4_store(float* sbase, unsigned voff, float v) {
  asm volatile("global_store_dword %0, %1, %2 offset:%3" :: "v"(voff), "v"(v), "s"(sbase), "n"(IMM) : "memory");
}

__device__ inline void c4_tile_coords(const Conv4RowsArgs& p, int wt, int& b, int& y, int& xw, int& x_new) {
  const int row = wt / p.wtiles_per_row;
  x_new = (wt - row * p.wtiles_per_row) * 32;
  xw = min(x_new, p.gout.W - 32);
  b = row / p.gout.H;
  y = row - b * p.gout.H;
}

__device__ inline void c4_issue_dma(const Conv4RowsArgs& p, int wt, unsigned lds_buf, unsigned lane16) {
  int b, y, xw, x_new;
  c4_tile_coords(p, wt, b, y, xw, x_new);
  const int ps = xw - 1 + p.gin.pw;                       // padded column of the tile's first input pixel
  const int pc = min(ps, p.gin.Wp - 64);                  // 64-pixel window clamped into the padded row
  const unsigned dst = lds_buf + (unsigned)((C4_SLOT0 - (ps - pc)) * 16);
  const float* r0 = p.x4 + (((long)b * p.gin.Hp + (y + p.gin.ph - 1)) * p.gin.Wp + pc) * 4;
  c4_dma(r0, lane16, dst);
  c4_dma(r0 + (long)p.gin.Wp * 4, lane16, dst + C4_ROW_BYTES);
  c4_dma(r0 + (long)p.gin.Wp * 8, lane16, dst + 2 * C4_ROW_BYTES);
}

#define C4_ROW_IMM(r) ((((r) & 3) + 8 * ((r) >> 2)) * 128)
#define C4_FOR_ROWS(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

// MODE 0: raw + moments, 1: lrelu(acc*scale+shift), 2: raw
template <int MODE>
__global__ __launch_bounds__(256) void conv4_rows_kernel(Conv4RowsArgs p) {
  __shared__ __attribute__((aligned(16))) char rows[4 * C4_WAVE_BYTES];
  __shared__ float part[8][32][3];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, li = lane & 31;
  const unsigned lane16 = (unsigned)lane * 16u;
  const unsigned io_off = (unsigned)(512 * h + 4 * li);
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((lds_ptr_t)rows)) + wave * C4_WAVE_BYTES;
  const char* rd_base = rows + wave * C4_WAVE_BYTES + (C4_SLOT0 + li) * 16 + 4 * h;     // this lane's (pixel li, channel h)
  float bw[9][2];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) { bw[tp][0] = p.wp[(tp * 2 + 0) * 64 + lane]; bw[tp][1] = p.wp[(tp * 2 + 1) * 64 + lane]; }
  const float bias_v = p.ep.bias ? p.ep.bias[li] : 0.f;
  float sc = 1.f, sh = 0.f;
  if (MODE == 1) { sc = p.ep.ep_scale[li]; sh = p.ep.ep_shift[li]; }
  float st_n = 0.f, st_c = 0.f, st_s1 = 0.f, st_s2 = 0.f;

  const int nw = gridDim.x * 4;
  int wt = blockIdx.x * 4 + wave;
  if (wt < p.nwtiles) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the weight / bias loads above
    c4_issue_dma(p, wt, lds_wave, lane16);
    // 16 dummy stores so that the first tile sees the same queue as every other one
#pragma unroll
    for (int r = 0; r < 16; ++r) c4_store<0>(as_store_dump, (unsigned)(lane * 4), 0.f);
  }
  int cur = 0;
  for (; wt < p.nwtiles; wt += nw, cur ^= 1) {
    const int nxt = min(wt + nw, p.nwtiles - 1);          // past the end: re-fetch a valid tile (keeps the count uniform)
    int b, y, xw, x_new;
    c4_tile_coords(p, wt, b, y, xw, x_new);
    float* z_base = p.ep.z + p.gout.vox(b, 0, y, xw) * 32;
    c4_issue_dma(p, nxt, lds_wave + (unsigned)((cur ^ 1) * 3 * C4_ROW_BYTES), lane16);
    asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
    const char* rd = rd_base + cur * 3 * C4_ROW_BYTES;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias_v;
    float a[9][2];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      a[tp][0] = *reinterpret_cast<const float*>(rd + (tp / 3) * C4_ROW_BYTES + (tp % 3) * 16);
      a[tp][1] = *reinterpret_cast<const float*>(rd + (tp / 3) * C4_ROW_BYTES + (tp % 3) * 16 + 8);
    }
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tp][0], bw[tp][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tp][1], bw[tp][1], acc, 0, 0, 0);
    }
#define C4_ST(r) { float v = acc[r];                                                  \
                   if (MODE == 1) { v = v * sc + sh; v = fmaxf(v, v * p.ep.slope); }   \
                   c4_store<C4_ROW_IMM(r)>(z_base, io_off, v); }
    C4_FOR_ROWS(C4_ST)
#undef C4_ST
    if (MODE == 0) {
      const int dup = x_new - xw;
      if (dup <= 0) {
        st_c = st_n == 0.f ? acc[0] : st_c;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = acc[r] - st_c; st_s1 += d; st_s2 = fmaf(d, d, st_s2); }
        st_n += 16.f;
      } else {
        st_c = st_n == 0.f ? acc[15] : st_c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
          const float d = row >= dup ? acc[r] - st_c : 0.f;
          st_s1 += d; st_s2 = fmaf(d, d, st_s2); st_n += row >= dup ? 1.f : 0.f;
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (MODE == 0 && p.ep.stat_mean != nullptr) {
    const float mean_l = st_n > 0.f ? st_c + st_s1 / st_n : 0.f;
    const float m2_l = st_n > 0.f ? fmaxf(st_s2 - st_s1 * st_s1 / st_n, 0.f) : 0.f;
    part[wave * 2 + h][li][0] = st_n; part[wave * 2 + h][li][1] = mean_l; part[wave * 2 + h][li][2] = m2_l;
    __syncthreads();
    if (threadIdx.x < 32) {
      TileStats run; run.n = 0.f; run.mean = 0.f; run.m2 = 0.f;
      for (int q = 0; q < 8; ++q) {
        TileStats t; t.n = part[q][li][0]; t.mean = part[q][li][1]; t.m2 = part[q][li][2];
        stats_merge(run, t);
      }
      stats_write(p.ep, blockIdx.x, run);
    }
  }
}

static bool conv4_rows_applicable(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s) {
  return s->kh == 3 && s->kw == 3 && s->stride == 1 && s->dil == 1 && s->pad_h == 1 && s->pad_w == 1 &&
         gin->H == gout->H && gin->W == gout->W && gout->W >= 32 && gin->ph >= 1 && gin->pw >= 1 &&
         gin->W + 2 * gin->pw >= 64 &&                      // a 64-pixel DMA window fits into a padded row
         (long)gout->B * gout->H * ((gout->W + 31) / 32) < (1L << 31);
}
static int conv4_rows_grid(const as_pcl* gout) {
  const long nwt = (long)gout->B * gout->H * ((gout->W + 31) / 32);
  long g = (nwt + 3) / 4;
  if (g > 1024) g = 1024;      // 36 KB of LDS per workgroup: four resident per CU, all of them at once
  return (int)g;
}

// packed[t][j][h][co] = w[co][c = 2j+h][t]  (0 for c >= Cin)
__global__ void conv4_pack_kernel(const float* __restrict__ w, float* __restrict__ packed, int T, int Cin) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= T * 128) return;
  const int co = idx & 31, h = (idx >> 5) & 1, j = (idx >> 6) & 1, t = idx >> 7;
  const int c = 2 * j + h;
  packed[idx] = c < Cin ? w[((long)co * Cin + c) * T + t] : 0.f;
}

// x4[b][y][x][:] = (ch0[b,0,y,x] if given), img[b,0..C-1,y,x], zero-filled to 4 channels.
__global__ __launch_bounds__(256) void pack_in4_kernel(const float* __restrict__ ch0, const float* __restrict__ img, int C,
                                                        float* __restrict__ x4, PclDev g) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long plane = (long)g.H * g.W;
  if (i >= (long)g.B * plane) return;
  const int x = i % g.W, y = (i / g.W) % g.H, b = i / plane;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  int n = 0;
  if (ch0) v[n++] = ch0[i];
  for (int c = 0; c < C && n < 4; ++c) v[n++] = img[((long)b * C + c) * plane + (long)y * g.W + x];
  *reinterpret_cast<f32x4*>(x4 + g.vox(b, 0, y, x) * 4) = (f32x4){v[0], v[1], v[2], v[3]};
}

// ---- weight gradient: dW[co][c][t] = sum_pix X4[pix*stride + off(t)][c] * G[pix][co] ------------------------
// MFMA rows i = k = 4*t + c in blocks of 32 (NB blocks cover 4*T), columns j = co, reduction over pixels.
struct Wgrad4Args {
  const float* x4;
  const float* gz;
  float* partial;      // [nchunks][NB][32][32]
  float* partial_db;   // [nchunks][32]
  PclDev gin, gout;
  int rows, rows_per_chunk, ntaps, stride;    // rows = work units (row segments), rows_per_chunk = units per workgroup
  int nseg;                                   // 64-step segments per output row
  int tap_off[AS_MAX_TAPS];
};

template <int NB>
#define W4_SEG_STEPS 32                       // multiple of the 8-step load groups
__global__ __launch_bounds__(256) void conv4_wgrad_kernel(Wgrad4Args p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [3][NB*16][64] + [4][32]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, li = lane & 31;
  const int chunk = blockIdx.x;
  const int W = p.gout.W, H = p.gout.H;
  const int r0 = chunk * p.rows_per_chunk;
  const int r1 = min(p.rows, r0 + p.rows_per_chunk);

  f32x16 acc[NB];
  int koff[NB];          // float offset of this lane's (tap, channel) relative to the pixel anchor
  bool kok[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    const int k = nb * 32 + li;
    kok[nb] = k < 4 * p.ntaps;
    const int t = kok[nb] ? (k >> 2) : 0;
    koff[nb] = p.tap_off[t] * 4 + (k & 3);
  }
  float bsum = 0.f;
  // loads of 8 steps are issued as a group before their MFMAs (see conv32_wgrad_kernel)
  constexpr int U = 8;
  const int nsteps = (W + 1) >> 1;
  for (int unit = r0 + wave; unit < r1; unit += 4) {
    const int row = unit / p.nseg, seg = unit - row * p.nseg;
    const int y = row % H, b = row / H;
    const float* xr = p.x4 + p.gin.vox(b, 0, y * p.stride, 0) * 4;
    const float* gr = p.gz + p.gout.vox(b, 0, y, 0) * 32 + li;
    const int s_end = min(nsteps, (seg + 1) * W4_SEG_STEPS);
    for (int s0 = seg * W4_SEG_STEPS; s0 < s_end; s0 += U) {
      float bv[U], av[U][NB];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int xc = 2 * (s0 + u) + h;
        const bool ok = xc < W;
        const int xcl = ok ? xc : W - 1;
#ifdef C4W_EXP_NOLOAD                                   // diagnostic builds (tests/tools/exp_step.sh): results are wrong
        const float g0 = (float)(xcl + lane);
#else
        const float g0 = gr[xcl * 32];
#endif
        bv[u] = ok ? g0 : 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#ifdef C4W_EXP_NOLOAD
          const float a0 = (float)(xcl * p.stride * 4 + koff[nb]);
#else
          const float a0 = xr[xcl * p.stride * 4 + koff[nb]];
#endif
          av[u][nb] = kok[nb] ? a0 : 0.f;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        bsum += bv[u];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#ifdef C4W_EXP_NOMFMA
          acc[nb][0] += av[u][nb] * bv[u];
#else
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][nb], bv[u], acc[nb], 0, 0, 0);
#endif
      }
    }
  }
  float* slab = lds;
  float* dbs = lds + 3 * NB * 16 * 64;
  bsum += __shfl_xor(bsum, 32, 64);
  if (h == 0) dbs[wave * 32 + li] = bsum;
  if (wave > 0) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) slab[((wave - 1) * NB * 16 + nb * 16 + r) * 64 + lane] = acc[nb][r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      float* out = p.partial + ((long)chunk * NB + nb) * 1024;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[nb][r];
        v += slab[(0 * NB * 16 + nb * 16 + r) * 64 + lane];
        v += slab[(1 * NB * 16 + nb * 16 + r) * 64 + lane];
        v += slab[(2 * NB * 16 + nb * 16 + r) * 64 + lane];
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
        out[i * 32 + li] = v;
      }
    }
    if (h == 0) p.partial_db[chunk * 32 + li] = dbs[li] + dbs[32 + li] + dbs[64 + li] + dbs[96 + li];
  }
}


// ---- weight gradient of the 5x5 stride-2 first layer with the input rows staged in LDS -----------------------------------
// conv4_wgrad_kernel<4> fetches every A operand with a gather of its own: four dword loads per lane and pixel pair, 128 load
// instructions per 32-step unit for data that 25 taps share; with three waves per SIMD their latency is not covered — diagnostic
// builds inside a step (tests/tools/exp_step.sh): 105 us as built, 76 with the operands made up in registers.  Here a wave copies
// the five input rows of its unit — 132 pixels of 16 bytes each, columns 4 s0 - 2 .. 4 s0 + 129 of rows 2y - 2 .. 2y + 2 — into a
// private LDS image with 11 coalesced loads (the NEXT unit's rows are requested as soon as this unit's sit in LDS) and reads the
// A operands with ds_read_b32 at a per-lane constant + an immediate; G stays a coalesced 256-byte load per step, one group of
// eight steps ahead.  Same units, same slabs, same reduction as conv4_wgrad_kernel<4>; the sums over a unit run in the same
// order, so the slabs are bit-identical to it.
#define C4W_ROW_PX 132
#define C4W_WAVE_BYTES (5 * C4W_ROW_PX * 16)          // 10,560

__global__ __launch_bounds__(256, 2) void conv4_s2_wgrad_kernel(Wgrad4Args p) {
  constexpr int NB = 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];   // rows: [4 waves][5][132][4]; later [3][NB*16][64] + [4][32]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, li = lane & 31;
  const int chunk = blockIdx.x;
  const int W = p.gout.W, H = p.gout.H;
  const int r0 = chunk * p.rows_per_chunk;
  const int r1 = min(p.rows, r0 + p.rows_per_chunk);
  char* rows = reinterpret_cast<char*>(lds) + wave * C4W_WAVE_BYTES;

  f32x16 acc[NB];
  unsigned aoff[NB];     // byte offset of this lane's (tap, channel) of step 0 in the staged image: ((ky*132 + kx)*4 + c)*4 + 32 h
  bool kok[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    const int k = nb * 32 + li;
    kok[nb] = k < 100;
    const int t = kok[nb] ? (k >> 2) : 0;
    aoff[nb] = (unsigned)((((t / 5) * C4W_ROW_PX + (t % 5)) * 4 + (k & 3)) * 4 + 32 * h);
  }
  float bsum = 0.f;
  const int nsteps = (W + 1) >> 1;
  // staging jobs of a lane: pixels lane and 64 + lane of each of the five rows; pixels 128..131 of row (lane >> 2) % 5 in one
  // more load (lanes 20..63 repeat rows 0..: same data, same destination)
  f32x4 ra[5], rb[5], rc;
  const int cky = (lane >> 2) % 5, cpx = 128 + (lane & 3);
  auto request = [&](int unit) {
    const int row = unit / p.nseg, seg = unit - row * p.nseg;
    const int y = row % H, b = row / H;
    const int c0 = 4 * seg * W4_SEG_STEPS - 2 + p.gin.pw;                 // first staged column (padded coordinates)
    const int vlim = p.gin.Wp - 1 - c0;                                   // (past the row's padded end: clamped, feeds masked steps)
    const float* src = p.x4 + (((long)b * p.gin.Hp + (2 * y - 2 + p.gin.ph)) * p.gin.Wp + c0) * 4;
    const int oa = min(lane, vlim) * 4, ob = min(64 + lane, vlim) * 4, oc = min(cpx, vlim) * 4;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
      const float* s = src + (long)ky * p.gin.Wp * 4;
      ra[ky] = *reinterpret_cast<const f32x4*>(s + oa);
      rb[ky] = *reinterpret_cast<const f32x4*>(s + ob);
    }
    rc = *reinterpret_cast<const f32x4*>(src + (long)cky * p.gin.Wp * 4 + oc);
  };
  // G of a whole unit (32 steps) in registers, the next unit's requested with its rows: everything a unit needs was asked for
  // a unit (128 MFMAs) earlier.  (With three waves per SIMD and G one group of eight steps ahead the kernel took 103-120 us —
  // 120 when the next rows were requested in front of the G loads: loads retire in order, every unit then began with the rows'
  // round trip — against 105 for the gather kernel and 76 for its matrix instructions alone.)
  float g0[W4_SEG_STEPS], g1[W4_SEG_STEPS];
  auto request_g = [&](float (&g)[W4_SEG_STEPS], int unit) {
    const int row = unit / p.nseg, seg = unit - row * p.nseg;
    const int y = row % H, b = row / H;
    const float* gr = p.gz + p.gout.vox(b, 0, y, 0) * 32 + li;
#pragma unroll
    for (int u = 0; u < W4_SEG_STEPS; ++u) {
      const int s = seg * W4_SEG_STEPS + u, xc = 2 * s + h;
      g[u] = gr[((s < nsteps && xc < W) ? xc : W - 1) * 32];
    }
  };
  auto multiply = [&](const float (&g)[W4_SEG_STEPS], int unit) {
    const int seg = unit % p.nseg;
    // the A operands of step sl + 1 are read from LDS before the MFMAs of step sl (hipcc otherwise waits for every ds_read
    // right in front of the MFMA that uses it)
    float an[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) an[nb] = *reinterpret_cast<const float*>(rows + aoff[nb]);
#pragma unroll
    for (int sl = 0; sl < W4_SEG_STEPS; ++sl) {           // step sl: pixels 2 sl + h, staged column 4 sl + 2 h + kx
      const int s = seg * W4_SEG_STEPS + sl;
      const float gv = (s < nsteps && 2 * s + h < W) ? g[sl] : 0.f;
      float ac[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) ac[nb] = kok[nb] ? an[nb] : 0.f;
      if (sl + 1 < W4_SEG_STEPS) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) an[nb] = *reinterpret_cast<const float*>(rows + aoff[nb] + 64 * (sl + 1));
      }
      __builtin_amdgcn_sched_barrier(0);
      bsum += gv;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[nb], gv, acc[nb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
      *reinterpret_cast<f32x4*>(rows + (ky * C4W_ROW_PX + lane) * 16) = ra[ky];
      *reinterpret_cast<f32x4*>(rows + (ky * C4W_ROW_PX + 64 + lane) * 16) = rb[ky];
    }
    *reinterpret_cast<f32x4*>(rows + (cky * C4W_ROW_PX + cpx) * 16) = rc;
  };
  int unit = r0 + wave;
  if (unit < r1) { request_g(g0, unit); request(unit); }
  while (unit < r1) {
    stage();
    if (unit + 4 < r1) { request_g(g1, unit + 4); request(unit + 4); }
    __builtin_amdgcn_sched_barrier(0);
    multiply(g0, unit);
    __builtin_amdgcn_sched_barrier(0);
    unit += 4;
    if (unit >= r1) break;
    stage();
    if (unit + 4 < r1) { request_g(g0, unit + 4); request(unit + 4); }
    __builtin_amdgcn_sched_barrier(0);
    multiply(g1, unit);
    __builtin_amdgcn_sched_barrier(0);
    unit += 4;
  }
  __syncthreads();                                        // every wave is done with its rows: the slabs may overwrite them
  float* slab = lds;
  float* dbs = lds + 3 * NB * 16 * 64;
  bsum += __shfl_xor(bsum, 32, 64);
  if (h == 0) dbs[wave * 32 + li] = bsum;
  if (wave > 0) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) slab[((wave - 1) * NB * 16 + nb * 16 + r) * 64 + lane] = acc[nb][r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      float* out = p.partial + ((long)chunk * NB + nb) * 1024;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[nb][r];
        v += slab[(0 * NB * 16 + nb * 16 + r) * 64 + lane];
        v += slab[(1 * NB * 16 + nb * 16 + r) * 64 + lane];
        v += slab[(2 * NB * 16 + nb * 16 + r) * 64 + lane];
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
        out[i * 32 + li] = v;
      }
    }
    if (h == 0) p.partial_db[chunk * 32 + li] = dbs[li] + dbs[32 + li] + dbs[64 + li] + dbs[96 + li];
  }
}

// ---- weight gradient, 3x3 stride-1 instance, on the vector ALUs ---------------------------------------------
// With K = 36 useful rows out of 64 and two pixels per v_mfma_f32_32x32x2_f32 the matrix core needs 64 cycles
// per pixel here; the plain FMA formulation needs 1152 FMAs per pixel = 18 wave-cycles per pixel-SIMD and no
// operand gather.  Eight lanes share a pixel (float4 of G = 4 output channels each) and keep dW[36 rows][4 co]
// in registers over all of the workgroup's row chunks; a pixel's nine taps are nine 16-byte loads that the
// eight lanes share.  One slab per workgroup in the layout conv4_wgrad_reduce_kernel reads (NB = 2).
struct Wgrad4RowsArgs {
  const float* x4;
  const float* gz;
  float* partial;      // [blocks][2][32][32]
  float* partial_db;   // [blocks][32]
  PclDev gin, gout;
  int nchunks, chunks_per_row;
  // APPLY flavour: gz is the layer's output gradient g_a; stage 3 of its BatchNorm backward is applied on the fly
  const float* bn_z; const float* bn_scale; const float* bn_shift; const float* bn_mean; const float* bn_coef;
  float* gz_out; float slope;
  // PROJ flavour: instead of g_z (32 channels) the nine per-tap projections h[t] = sum_co g_z[co] * w_proj[t][co] go out
  // ([B][9][H][W] planes): all the thin 32->1 data gradient that follows needs of g_z
  const float* w_proj; float* h_out;
};

// APPLY: g_z = (g_a*lrelu'(z*scale+shift) - k1 - (z-mean)*k2)*k3 is formed in registers from g_a and z (the lane's four
// output channels), used for the products and written out for whoever needs g_z next — the element-wise pass of
// as_bn_act_bwd (read g_a, read z, write g_z) disappears.
template <bool APPLY, bool PROJ>
__global__ __launch_bounds__(256) void conv4_wgrad_rows_kernel(Wgrad4RowsArgs p) {
  __shared__ float red[32][33];
  __shared__ __attribute__((aligned(16))) float wpr[9 * 32];
  if (PROJ) {
    for (int i = threadIdx.x; i < 9 * 32; i += 256) wpr[i] = p.w_proj[i];
    __syncthreads();
  }
  const int c4 = threadIdx.x & 7, vl = threadIdx.x >> 3;
  f32x4 acc[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  f32x4 k1, k2, k3, bsc, bsh, bmu;
  if (APPLY) {
    k1 = *reinterpret_cast<const f32x4*>(p.bn_coef + c4 * 4); k2 = *reinterpret_cast<const f32x4*>(p.bn_coef + 32 + c4 * 4);
    k3 = *reinterpret_cast<const f32x4*>(p.bn_coef + 64 + c4 * 4);
    bsc = *reinterpret_cast<const f32x4*>(p.bn_scale + c4 * 4); bsh = *reinterpret_cast<const f32x4*>(p.bn_shift + c4 * 4);
    bmu = *reinterpret_cast<const f32x4*>(p.bn_mean + c4 * 4);
  }
  for (int ch = blockIdx.x; ch < p.nchunks; ch += gridDim.x) {
    const int rowi = ch / p.chunks_per_row, cx = ch - rowi * p.chunks_per_row;
    const int y = rowi % p.gout.H, b = rowi / p.gout.H, x0 = cx * 128;
    const int nf4 = min(128, p.gout.W - x0) * 8;
    const long goff = p.gout.vox(b, 0, y, x0) * 32;
    const float* grow = p.gz + goff;
    const float* xrow = p.x4 + (((long)b * p.gin.Hp + (y + p.gin.ph - 1)) * p.gin.Wp + (x0 + p.gin.pw - 1)) * 4;
    const long rs = (long)p.gin.Wp * 4;
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      const int f = threadIdx.x + 256 * k4;
      if (f < nf4) {
        const int vx = f >> 3;
        f32x4 g4 = *reinterpret_cast<const f32x4*>(grow + f * 4);
        if (APPLY) {
          const f32x4 zz = *reinterpret_cast<const f32x4*>(p.bn_z + goff + f * 4);
          const f32x4 yy = zz * bsc + bsh;
          f32x4 gy;
          gy.x = yy.x > 0.f ? g4.x : g4.x * p.slope; gy.y = yy.y > 0.f ? g4.y : g4.y * p.slope;
          gy.z = yy.z > 0.f ? g4.z : g4.z * p.slope; gy.w = yy.w > 0.f ? g4.w : g4.w * p.slope;
          g4 = (gy - k1 - (zz - bmu) * k2) * k3;
          // the products below must see the ROUNDED g_z (what the element-wise pass would have stored), not a value
          // hipcc contracts into their FMAs
          asm volatile("" : "+v"(g4.x), "+v"(g4.y), "+v"(g4.z), "+v"(g4.w));
          if (!PROJ) *reinterpret_cast<f32x4*>(p.gz_out + goff + f * 4) = g4;
        }
        if (PROJ) {
          // nine projections of this pixel's g_z: the lane's four channels, then a butterfly over the pixel's 8 lanes
          float hp[9];
#pragma unroll
          for (int tp = 0; tp < 9; ++tp) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(wpr + tp * 32 + c4 * 4);
            hp[tp] = g4.x * w4.x + g4.y * w4.y + g4.z * w4.z + g4.w * w4.w;
          }
#pragma unroll
          for (int tp = 0; tp < 9; ++tp) {
            hp[tp] += __shfl_xor(hp[tp], 1, 64); hp[tp] += __shfl_xor(hp[tp], 2, 64); hp[tp] += __shfl_xor(hp[tp], 4, 64);
          }
          const long plane = (long)p.gout.H * p.gout.W;
          float* hrow = p.h_out + (long)b * 9 * plane + (long)y * p.gout.W + x0 + vx;
          float mine = hp[0];
#pragma unroll
          for (int tp = 1; tp < 8; ++tp) mine = c4 == tp ? hp[tp] : mine;
          hrow[c4 * plane] = mine;                             // lane c4 of the pixel writes tap c4, lane 0 also tap 8
          if (c4 == 0) hrow[8 * plane] = hp[8];
        }
        f32x4 px[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) px[t] = *reinterpret_cast<const f32x4*>(xrow + (t / 3) * rs + (vx + t % 3) * 4);
        bsum += g4;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          acc[4 * t + 0] += px[t].x * g4; acc[4 * t + 1] += px[t].y * g4;
          acc[4 * t + 2] += px[t].z * g4; acc[4 * t + 3] += px[t].w * g4;
        }
      }
    }
  }
  float* out = p.partial + (long)blockIdx.x * 2048;
#pragma unroll
  for (int k = 0; k < 37; ++k) {
    const f32x4 v = k < 36 ? acc[k < 36 ? k : 0] : bsum;
    __syncthreads();
    red[vl][c4 * 4 + 0] = v.x; red[vl][c4 * 4 + 1] = v.y; red[vl][c4 * 4 + 2] = v.z; red[vl][c4 * 4 + 3] = v.w;
    __syncthreads();
    if (threadIdx.x < 32) {
      float sum = 0.f;
      for (int j = 0; j < 32; ++j) sum += red[j][threadIdx.x];
      if (k < 36) out[k * 32 + threadIdx.x] = sum;
      else p.partial_db[blockIdx.x * 32 + threadIdx.x] = sum;
    }
  }
}

// ---- weight gradient, 3x3 stride-1 instance, third form: tile staged in LDS, products on the matrix cores --------------
// The vector-ALU form above keeps dW[36][4 co] per lane (144 accumulators, 246 registers, two waves per SIMD): every load's
// round trip is exposed and 211 us go by for 745 MB.  Here a workgroup stages a 128-pixel row chunk once — g_z formed from
// g_a and z on the way in (stage 3 of the BatchNorm backward), the three 4-channel input rows next to it — and then
//   * dW[k = 4 tap + c][co] += x[p + off_tap][c] * g_z[p][co]: v_mfma_f32_32x32x2_f32 with two PIXELS as the K dimension,
//     the 36 k-rows as M (two blocks: taps 0-7, tap 8 in rows 0-3 of the second), co as N; both operands are ds_read_b32
//     with immediate offsets (the g_z tile has a row pitch of 36 floats: conflict-free);
//   * the nine per-tap projections h[t][p] = sum_co g_z[p][co] * w_proj[t][co] (all the 32->1 data gradient that follows
//     needs of g_z) on the vector ALUs, two threads per pixel;
// 32 accumulator registers per lane, many workgroups per CU: the latency hides behind other workgroups.
// Round 4, measured inside a step (tests/tools/exp_step.sh): 154 us with the projections, 107 us without them
// (EXTRA=-DW4M_EXP_NOPROJ); the same projections as 16 MFMAs per wave and chunk (M = 32 pixels, N = 9 of 32 columns, K = co,
// the result through a wave-private LDS patch to 128-byte stores) passed the parity tests and took 149 us: the matrix form
// executes 32/9 of the products, no faster than 144 FMAs + 36 LDS reads per thread.  Not kept.
#define W4M_PITCH 36                              // floats per pixel of the g_z tile
#define W4M_GZ_BYTES (128 * W4M_PITCH * 4)        // 18,432
#define W4M_X_OFF W4M_GZ_BYTES                    // three input rows of 130 pixels x 4 channels
#define W4M_X_BYTES (3 * 130 * 16)                // 6,240
#define W4M_W_OFF (W4M_X_OFF + W4M_X_BYTES)       // w_proj [9][32]
#define W4M_C_OFF (W4M_W_OFF + 9 * 32 * 4)        // the BatchNorm constants [6][32]
#define W4M_LDS_BYTES (W4M_C_OFF + 6 * 32 * 4)    // 26,592

template <bool PROJ>
__global__ __launch_bounds__(256, 4) void conv4_wgrad_mfma_kernel(Wgrad4RowsArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[W4M_LDS_BYTES];
  float* gzt = reinterpret_cast<float*>(smem);
  float* xt = reinterpret_cast<float*>(smem + W4M_X_OFF);
  float* wpr = reinterpret_cast<float*>(smem + W4M_W_OFF);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5, li = lane & 31;
  const int c4 = t & 7;
  float* cft = reinterpret_cast<float*>(smem + W4M_C_OFF);     // k1, k2, k3, scale, shift, mean [6][32]: re-read per chunk
  if (PROJ) for (int i = t; i < 9 * 32; i += 256) wpr[i] = p.w_proj[i];
  if (t < 96) cft[t] = p.bn_coef[t];
  if (t < 32) { cft[96 + t] = p.bn_scale[t]; cft[128 + t] = p.bn_shift[t]; cft[160 + t] = p.bn_mean[t]; }
  f32x16 acc0;                                     // k rows 0-31 (taps 0-7) on the matrix cores
  f32x4 acc8 = {0.f, 0.f, 0.f, 0.f};               // k rows 32-35 (tap 8) of this lane's co and pixel parity: four FMAs a step
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
  float bsum = 0.f;
  // operand addresses of this lane: A block 0 = x[row tap/3][px + tap%3][c], tap = li >> 2, c = li & 3;
  // A block 1 = x[2][px + 2][li] for li < 4; B = g_z[px][co = li];  px = 32 wave + 2 s + h
  const int tap = li >> 2;
  const float* a0p = xt + ((tap / 3) * 130 + (32 * wave + h + tap % 3)) * 4 + (li & 3);
  const float* a1p = xt + (2 * 130 + (32 * wave + h + 2)) * 4;
  const float* bp = gzt + (32 * wave + h) * W4M_PITCH + li;

  for (int ch = blockIdx.x; ch < p.nchunks; ch += gridDim.x) {
    const int rowi = ch / p.chunks_per_row, cx = ch - rowi * p.chunks_per_row;
    const int y = rowi % p.gout.H, b = rowi / p.gout.H, x0 = cx * 128;
    const int npx = min(128, p.gout.W - x0);
    const long goff = p.gout.vox(b, 0, y, x0) * 32;
    __syncthreads();                               // the previous chunk's tile is no longer read
    // ---- stage: g_z = stage 3 of the BatchNorm backward of (g_a, z), zeros beyond the row's end ----
    const f32x4 k1 = *reinterpret_cast<const f32x4*>(cft + c4 * 4), k2 = *reinterpret_cast<const f32x4*>(cft + 32 + c4 * 4),
                k3 = *reinterpret_cast<const f32x4*>(cft + 64 + c4 * 4);
    const f32x4 bsc = *reinterpret_cast<const f32x4*>(cft + 96 + c4 * 4), bsh = *reinterpret_cast<const f32x4*>(cft + 128 + c4 * 4),
                bmu = *reinterpret_cast<const f32x4*>(cft + 160 + c4 * 4);
#pragma unroll 2
    for (int k4 = 0; k4 < 4; ++k4) {
      const int f = t + 256 * k4, vx = f >> 3;
      f32x4 g4 = {0.f, 0.f, 0.f, 0.f};
      if (vx < npx) {
        g4 = *reinterpret_cast<const f32x4*>(p.gz + goff + f * 4);
        const f32x4 zz = *reinterpret_cast<const f32x4*>(p.bn_z + goff + f * 4);
        const f32x4 yy = zz * bsc + bsh;
        f32x4 gy;
        gy.x = yy.x > 0.f ? g4.x : g4.x * p.slope; gy.y = yy.y > 0.f ? g4.y : g4.y * p.slope;
        gy.z = yy.z > 0.f ? g4.z : g4.z * p.slope; gy.w = yy.w > 0.f ? g4.w : g4.w * p.slope;
        g4 = (gy - k1 - (zz - bmu) * k2) * k3;
      }
      *reinterpret_cast<f32x4*>(gzt + vx * W4M_PITCH + c4 * 4) = g4;
    }
    // the three input rows y-1, y, y+1, pixels x0-1 .. x0+128 (the padded 4-channel layout has the halo)
    for (int i = t; i < 3 * 130; i += 256) {
      const int r = i / 130, xx = i - r * 130;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (xx <= npx + 1)
        v = *reinterpret_cast<const f32x4*>(p.x4 + (((long)b * p.gin.Hp + (y + p.gin.ph - 1 + r)) * p.gin.Wp + (x0 + p.gin.pw - 1 + xx)) * 4);
      *reinterpret_cast<f32x4*>(xt + i * 4) = v;
    }
    __syncthreads();
    // ---- products: 16 steps of two pixels ----
#pragma unroll 4
    for (int s = 0; s < 16; ++s) {
      const float bv = bp[2 * s * W4M_PITCH];
      const float av0 = a0p[2 * s * 4];
      const f32x4 x8 = *reinterpret_cast<const f32x4*>(a1p + 2 * s * 4);     // one address per half-wave: a broadcast
      bsum += bv;
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, bv, acc0, 0, 0, 0);
      acc8 += x8 * bv;
    }
#ifdef W4M_EXP_NOPROJ                                   // diagnostic build (tests/tools/exp_step.sh): results are wrong
    if (false) {
#else
    if (PROJ) {
#endif
      // two threads per pixel (16 channels each), nine projections, one shuffle to combine
      const int px = t >> 1, q = t & 1;
      const float* gp = gzt + px * W4M_PITCH + 16 * q;
      f32x4 gv[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) gv[m] = *reinterpret_cast<const f32x4*>(gp + 4 * m);
      float hp[9];
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        float sum = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const f32x4 w4 = *reinterpret_cast<const f32x4*>(wpr + tp * 32 + 16 * q + 4 * m);
          sum += gv[m].x * w4.x + gv[m].y * w4.y + gv[m].z * w4.z + gv[m].w * w4.w;
        }
        hp[tp] = sum + __shfl_xor(sum, 1, 64);
      }
      if (px < npx) {
        const long plane = (long)p.gout.H * p.gout.W;
        float* hrow = p.h_out + (long)b * 9 * plane + (long)y * p.gout.W + x0 + px;
        if (q == 0) { hrow[0] = hp[0]; hrow[plane] = hp[1]; hrow[2 * plane] = hp[2]; hrow[3 * plane] = hp[3]; hrow[4 * plane] = hp[4]; }
        else { hrow[5 * plane] = hp[5]; hrow[6 * plane] = hp[6]; hrow[7 * plane] = hp[7]; hrow[8 * plane] = hp[8]; }
      }
    }
  }
  // ---- one slab per workgroup in the layout conv4_wgrad_reduce_kernel reads (NB = 2): [k][co], k = 4 tap + c ----
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);     // [4 waves][36][32] floats = 18,432 B
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    red[(wave * 36 + row) * 32 + li] = acc0[r];
  }
  acc8.x += __shfl_xor(acc8.x, 32, 64); acc8.y += __shfl_xor(acc8.y, 32, 64);
  acc8.z += __shfl_xor(acc8.z, 32, 64); acc8.w += __shfl_xor(acc8.w, 32, 64);
  if (h == 0) {
    red[(wave * 36 + 32) * 32 + li] = acc8.x; red[(wave * 36 + 33) * 32 + li] = acc8.y;
    red[(wave * 36 + 34) * 32 + li] = acc8.z; red[(wave * 36 + 35) * 32 + li] = acc8.w;
  }
  bsum += __shfl_xor(bsum, 32, 64);
  __syncthreads();
  float* out = p.partial + (long)blockIdx.x * 2048;
  for (int i = t; i < 36 * 32; i += 256) out[i] = red[i] + red[36 * 32 + i] + red[2 * 36 * 32 + i] + red[3 * 36 * 32 + i];
  __syncthreads();
  if (h == 0) red[wave * 32 + li] = bsum;
  __syncthreads();
  if (t < 32) p.partial_db[blockIdx.x * 32 + t] = red[t] + red[32 + t] + red[64 + t] + red[96 + t];
}

static bool conv4_wgrad_rows_applicable(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s) {
  return s->kh == 3 && s->kw == 3 && s->stride == 1 && s->dil == 1 && s->pad_h == 1 && s->pad_w == 1 &&
         gin->H == gout->H && gin->W == gout->W && gin->ph >= 1 && gin->pw >= 1 &&
         (long)gout->B * gout->H * ((gout->W + 127) / 128) < (1L << 31);
}
static int conv4_wgrad_rows_grid(const as_pcl* gout) {
  const long nch = (long)gout->B * gout->H * ((gout->W + 127) / 128);
  long g = (nch + 3) / 4;
  if (g > 1024) g = 1024;
  return (int)(g < 1 ? 1 : g);
}

// dW[co][c][t] = sum_chunks partial[chunk][k>>5][k&31][co], k = 4t + c
// 1024 threads = 64 consecutive slab elements x 16 slab slices (the scheme of wgrad_reduce_kernel): every thread sums every
// 16th slab of its element — consecutive threads read consecutive floats of one slab — then the slices are added in fixed
// order through LDS (deterministic).  The first form gave each output element to 32 lanes that walked the slabs with a
// stride of a whole slab: 17 us for 8 MB of slabs; this one 5.
__global__ __launch_bounds__(1024) void conv4_wgrad_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ partial_db,
                                                                   int nchunks, int NB, int T, int Cin, float* __restrict__ dW,
                                                                   float* __restrict__ db, int accumulate) {
  __shared__ float red[16][64];
  const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int nk = 4 * T;                                // k rows that exist (the slab holds NB * 32)
  const int total = nk * 32;                           // slab elements [k][co] that matter
  const int idx = blockIdx.x * 64 + o;
  const bool is_w = idx < total, is_b = !is_w && db != nullptr && idx < total + 32;
  float s = 0.f;
  if (is_w) {
    const float* src = partial + idx;                  // element (k, co) sits at k*32 + co of every slab
    const long pitch = (long)NB * 1024;
#pragma unroll 8
    for (int c = sl; c < nchunks; c += 16) s += src[c * pitch];
  } else if (is_b) {
    const float* src = partial_db + (idx - total);
#pragma unroll 8
    for (int c = sl; c < nchunks; c += 16) s += src[c * 32];
  }
  red[sl][o] = s;
  __syncthreads();
  if (sl == 0 && (is_w || is_b)) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][o];
    if (is_w) {
      const int co = idx & 31, k = idx >> 5, tp = k >> 2, c = k & 3;
      if (c < Cin) {
        float* dst = dW + ((long)co * Cin + c) * T + tp;
        *dst = accumulate ? *dst + t : t;
      }
    } else {
      db[idx - total] = accumulate ? db[idx - total] + t : t;
    }
  }
}

// ---- host -------------------------------------------------------------------------------------------------
static int fill_taps4(const as_pcl* gin, const as_conv_shape* s, int* tap_off) {
  const int Wp = gin->W + 2 * gin->pw;
  int n = 0;
  for (int j = 0; j < s->kh; ++j)
    for (int l = 0; l < s->kw; ++l) tap_off[n++] = (j * s->dil - s->pad_h) * Wp + (l * s->dil - s->pad_w);
  return n;
}

static int check4(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s, const char* who) {
  AS_CHECK_ARG(gin && gout && s && as_pcl_ok(gout), "%s: bad geometry", who);
  AS_CHECK_ARG(gin->D == 1 && gout->D == 1 && gin->pd == 0 && gout->pd == 0 && s->kd == 1, "%s: 2-D only", who);
  AS_CHECK_ARG(gin->B == gout->B && gin->B > 0 && gin->H > 0 && gin->W > 0, "%s: bad input extent", who);
  const int T = s->kh * s->kw;
  AS_CHECK_ARG(T >= 1 && T <= AS_MAX_TAPS && s->dil >= 1 && s->stride >= 1, "%s: unsupported kernel", who);
  const int eh = (gin->H + 2 * s->pad_h - s->dil * (s->kh - 1) - 1) / s->stride + 1;
  const int ew = (gin->W + 2 * s->pad_w - s->dil * (s->kw - 1) - 1) / s->stride + 1;
  AS_CHECK_ARG(eh == gout->H && ew == gout->W, "%s: output extent mismatch", who);
  const int hi_h = (gout->H - 1) * s->stride + s->dil * (s->kh - 1) - s->pad_h - (gin->H - 1);
  const int hi_w = (gout->W - 1) * s->stride + s->dil * (s->kw - 1) - s->pad_w - (gin->W - 1);
  AS_CHECK_ARG(s->pad_h <= gin->ph && s->pad_w <= gin->pw && hi_h <= gin->ph && hi_w <= gin->pw,
               "%s: input halo too small", who);
  return AS_OK;
}

extern "C" int64_t as_pcl4_numel(const as_pcl* g) {
  if (!g) return -1;
  return (int64_t)g->B * (g->H + 2 * g->ph) * (g->W + 2 * g->pw) * 4;
}

extern "C" int as_pack_in4(const float* ch0, const float* img, int C, float* x4, const as_pcl* g, void* stream) {
  AS_CHECK_ARG(g && g->D == 1 && g->pd == 0 && g->B > 0 && g->H > 0 && g->W > 0, "as_pack_in4: bad geometry");
  AS_CHECK_ARG(img && x4 && C >= 1 && C + (ch0 ? 1 : 0) <= 4, "as_pack_in4: bad argument");
  const long n = (long)g->B * g->H * g->W;
  hipLaunchKernelGGL(pack_in4_kernel, dim3(as_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, ch0, img, C, x4,
                     as_make_dev(g));
  AS_CHECK_LAUNCH("as_pack_in4");
  return AS_OK;
}

extern "C" int as_conv4_pack_weights(const float* w, int Cin, float* packed, const as_conv_shape* s, void* stream) {
  AS_CHECK_ARG(w && packed && s && Cin >= 1 && Cin <= 4, "as_conv4_pack_weights: bad argument");
  const int T = s->kh * s->kw;
  AS_CHECK_ARG(T >= 1 && T <= AS_MAX_TAPS, "as_conv4_pack_weights: %d taps unsupported", T);
  hipLaunchKernelGGL(conv4_pack_kernel, dim3(as_div_up(T * 128, 256)), dim3(256), 0, (hipStream_t)stream, w, packed, T, Cin);
  AS_CHECK_LAUNCH("as_conv4_pack_weights");
  return AS_OK;
}

// Number of BatchNorm partials as_conv4_fwd writes for this configuration.
extern "C" int as_conv4_stat_parts(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s) {
  if (!gin || !gout || !s || !as_pcl_ok(gout)) return AS_ERR_ARG;
  if (conv4_rows_applicable(gin, gout, s)) return conv4_rows_grid(gout);
  return as_div_up((int64_t)gout->B * gout->H * gout->W, 128);
}

extern "C" int as_conv4_fwd(const float* x4, const as_pcl* gin, const float* packed_w, const float* bias,
                            float* z, const as_pcl* gout, const as_conv_shape* s,
                            int epilogue, const float* ep_scale, const float* ep_shift, float slope,
                            float* stat_mean, float* stat_m2, float* stat_cnt, void* stream) {
  if (int e = check4(gin, gout, s, "as_conv4_fwd")) return e;
  AS_CHECK_ARG(x4 && packed_w && z, "as_conv4_fwd: null pointer");
  AS_CHECK_ARG(epilogue_args_ok(epilogue, ep_scale, ep_shift, stat_mean, stat_m2, stat_cnt), "as_conv4_fwd: bad epilogue arguments");
  if (conv4_rows_applicable(gin, gout, s)) {
    Conv4RowsArgs r;
    r.x4 = x4; r.wp = packed_w;
    r.ep.bias = bias; r.ep.z = z; r.ep.ep_scale = ep_scale; r.ep.ep_shift = ep_shift; r.ep.residual = nullptr;
    r.ep.stat_mean = epilogue == 0 ? stat_mean : nullptr; r.ep.stat_m2 = epilogue == 0 ? stat_m2 : nullptr;
    r.ep.stat_cnt = epilogue == 0 ? stat_cnt : nullptr;
    r.ep.epilogue = epilogue; r.ep.slope = slope;
    r.gin = as_make_dev(gin); r.gout = as_make_dev(gout);
    r.wtiles_per_row = (gout->W + 31) / 32;
    r.nwtiles = gout->B * gout->H * r.wtiles_per_row;
    const int grid = conv4_rows_grid(gout);
    hipStream_t st = (hipStream_t)stream;
    if (epilogue == 1) hipLaunchKernelGGL(conv4_rows_kernel<1>, dim3(grid), dim3(256), 0, st, r);
    else if (r.ep.stat_mean) hipLaunchKernelGGL(conv4_rows_kernel<0>, dim3(grid), dim3(256), 0, st, r);
    else hipLaunchKernelGGL(conv4_rows_kernel<2>, dim3(grid), dim3(256), 0, st, r);
    AS_CHECK_LAUNCH("as_conv4_fwd(rows)");
    return AS_OK;
  }
  if (g_conv4_s2 && epilogue == 0 && !stat_mean && conv4_s2_applicable(gin, gout, s)) {
    Conv4S2Args q;
    q.x4 = x4; q.wp = packed_w; q.bias = bias; q.z = z; q.gin = as_make_dev(gin); q.gout = as_make_dev(gout);
    q.nseg = (gout->W + 31) / 32; q.ntiles = gout->B * gout->H * q.nseg;
    hipLaunchKernelGGL(conv4_s2_fwd_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, q);   // two workgroups per CU (180 registers; three with 168 and two spilled: 78 us against 75)
    AS_CHECK_LAUNCH("as_conv4_fwd(5x5 stride 2, staged rows)");
    return AS_OK;
  }
  Conv4Args a;
  a.x4 = x4; a.wp = packed_w;
  a.ep.bias = bias; a.ep.z = z; a.ep.ep_scale = ep_scale; a.ep.ep_shift = ep_shift; a.ep.residual = nullptr;
  a.ep.stat_mean = epilogue == 0 ? stat_mean : nullptr; a.ep.stat_m2 = epilogue == 0 ? stat_m2 : nullptr;
  a.ep.stat_cnt = epilogue == 0 ? stat_cnt : nullptr;
  a.ep.epilogue = epilogue; a.ep.slope = slope;
  a.gin = as_make_dev(gin); a.gout = as_make_dev(gout);
  const int64_t M = (int64_t)gout->B * gout->H * gout->W;
  a.M = (int)M; a.stride = s->stride; a.ntaps = fill_taps4(gin, s, a.tap_off);
  if (a.ntaps == 25) hipLaunchKernelGGL(conv4_fwd_kernel<25>, dim3(as_div_up(M, 128)), dim3(256), 0, (hipStream_t)stream, a);
  else if (a.ntaps == 9) hipLaunchKernelGGL(conv4_fwd_kernel<9>, dim3(as_div_up(M, 128)), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(conv4_fwd_kernel<0>, dim3(as_div_up(M, 128)), dim3(256), 0, (hipStream_t)stream, a);
  AS_CHECK_LAUNCH("as_conv4_fwd");
  return AS_OK;
}

// A wave's unit of work is one 64-pixel segment of one output row (32 steps of a pixel pair); a workgroup takes
// `rpc` consecutive units, a multiple of four (its waves take every fourth).  Whole rows per wave left three quarters of the
// chip idle on a single image: 188 rows = 188 waves for 1024 SIMDs, each with a 311-step dependent chain.  The launch is ONE
// round of resident workgroups: with 64-step units and 1024 wanted chunks the 5x5 layer at the bench workload was 940
// workgroups of two units per wave on 768 slots (157 registers: three workgroups per CU) — a second round for a fifth of the
// work: 82 us for the matrix instructions alone where 49 are nominal (tests/tools/exp_step.sh, -DC4W_EXP_NOLOAD).
static int plan4_segments(const as_pcl* gout) { return (((gout->W + 1) >> 1) + W4_SEG_STEPS - 1) / W4_SEG_STEPS; }
static bool plan4_staged_shape(const as_conv_shape* s) {     // the 5x5 stride-2 first layer (conv4_s2_wgrad_kernel)
  return s->kh == 5 && s->kw == 5 && s->stride == 2 && s->dil == 1 && s->pad_h == 2 && s->pad_w == 2;
}
static void plan4(const as_pcl* gout, const as_conv_shape* s, bool staged, int* nb, int* rpc, int* nchunks) {
  const int T = s->kh * s->kw;
  *nb = (4 * T + 31) / 32;
  const int units = gout->B * gout->H * plan4_segments(gout);
  // resident workgroups (the register count decides): the staged-row kernel runs two per CU
  const int slots = staged ? 512 : (*nb >= 3 ? 768 : 1024);
  int r = (units + slots - 1) / slots;
  r = (r + 3) / 4 * 4;
  if (r < 4) r = 4;
  *rpc = r;
  *nchunks = (units + r - 1) / r;
}

extern "C" int64_t as_conv4_wgrad_workspace(const as_pcl* gout, const as_conv_shape* s) {
  if (!gout || !s || !as_pcl_ok(gout)) return -1;
  int nb, rpc, nchunks;
  plan4(gout, s, false, &nb, &rpc, &nchunks);               // (the larger of the two plans: as_conv4_s2_enable may change in between)
  int64_t need = (int64_t)nchunks * nb * 1024 + (int64_t)nchunks * 32;
  if (plan4_staged_shape(s)) {
    plan4(gout, s, true, &nb, &rpc, &nchunks);
    const int64_t n2 = (int64_t)nchunks * nb * 1024 + (int64_t)nchunks * 32;
    if (n2 > need) need = n2;
  }
  const int64_t rows_need = (int64_t)conv4_wgrad_rows_grid(gout) * (2048 + 32);     // the 3x3 stride-1 instance
  if (s->kh == 3 && s->kw == 3 && s->stride == 1 && rows_need > need) need = rows_need;
  return need;
}

template <int NB>
static void launch4(const Wgrad4Args& a, int nchunks, hipStream_t st) {
  const size_t lds = (size_t)(3 * NB * 16 * 64 + 4 * 32) * sizeof(float);
  hipLaunchKernelGGL(conv4_wgrad_kernel<NB>, dim3(nchunks), dim3(256), lds, st, a);
}

static int conv4_wgrad_rows_launch(const float* x4, const as_pcl* gin, const float* gz, const as_pcl* gout, int Cin,
                                   float* dW, float* db, int accumulate, float* workspace, bool apply,
                                   const float* bn_z, const float* scale, const float* shift, const float* mean,
                                   const float* coef, float slope, float* gz_out, void* stream,
                                   const float* w_proj = nullptr, float* h_out = nullptr) {
  Wgrad4RowsArgs r;
  const int grid = conv4_wgrad_rows_grid(gout);
  r.x4 = x4; r.gz = gz; r.partial = workspace; r.partial_db = workspace + (int64_t)grid * 2048;
  r.gin = as_make_dev(gin); r.gout = as_make_dev(gout);
  r.chunks_per_row = (gout->W + 127) / 128; r.nchunks = gout->B * gout->H * r.chunks_per_row;
  r.bn_z = bn_z; r.bn_scale = scale; r.bn_shift = shift; r.bn_mean = mean; r.bn_coef = coef; r.gz_out = gz_out; r.slope = slope;
  hipStream_t st = (hipStream_t)stream;
  r.w_proj = w_proj; r.h_out = h_out;
  if (apply && w_proj) hipLaunchKernelGGL(conv4_wgrad_mfma_kernel<true>, dim3(grid), dim3(256), 0, st, r);   // grid <= 1,024 = 4 per CU
  else if (apply) hipLaunchKernelGGL((conv4_wgrad_rows_kernel<true, false>), dim3(grid), dim3(256), 0, st, r);
  else hipLaunchKernelGGL((conv4_wgrad_rows_kernel<false, false>), dim3(grid), dim3(256), 0, st, r);
  AS_CHECK_LAUNCH("as_conv4_wgrad(rows)");
  hipLaunchKernelGGL(conv4_wgrad_reduce_kernel, dim3(as_div_up(4 * 9 * 32 + 32, 64)), dim3(1024), 0, st, r.partial,
                     r.partial_db, grid, 2, 9, Cin, dW, db, accumulate);
  AS_CHECK_LAUNCH("as_conv4_wgrad(reduce)");
  return AS_OK;
}

// Weight gradient fused with stage 3 of the layer's BatchNorm backward (see as_conv32_wgrad_bnapply).
extern "C" int as_conv4_wgrad_bnapply_ok(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s) {
  if (!gin || !gout || !s || !as_pcl_ok(gout)) return AS_ERR_ARG;
  return conv4_wgrad_rows_applicable(gin, gout, s) ? 1 : 0;
}

extern "C" int as_conv4_wgrad_bnapply(const float* x4, const as_pcl* gin, const float* g_a, const float* z,
                                      const as_pcl* gout, const as_conv_shape* s, int Cin, const float* scale,
                                      const float* shift, const float* mean, const float* coef, float slope, float* g_z,
                                      float* dW, float* db, int accumulate, float* workspace, void* stream) {
  if (int e = check4(gin, gout, s, "as_conv4_wgrad_bnapply")) return e;
  AS_CHECK_ARG(x4 && g_a && z && scale && shift && mean && coef && g_z && dW && workspace && Cin >= 1 && Cin <= 4,
               "as_conv4_wgrad_bnapply: bad argument");
  AS_CHECK_ARG(conv4_wgrad_rows_applicable(gin, gout, s), "as_conv4_wgrad_bnapply: configuration not supported");
  return conv4_wgrad_rows_launch(x4, gin, g_a, gout, Cin, dW, db, accumulate, workspace, true, z, scale, shift, mean, coef,
                                 slope, g_z, stream);
}

// As as_conv4_wgrad_bnapply, but g_z is not written: the nine per-tap projections h[t][p] = sum_co g_z[p][co] * w_proj[t][co]
// are ([B][9][H][W]), which is all a following 3x3 32->1 data gradient needs (as_tap_gather sums the nine shifted planes).
extern "C" int as_conv4_wgrad_bnapply_proj(const float* x4, const as_pcl* gin, const float* g_a, const float* z,
                                           const as_pcl* gout, const as_conv_shape* s, int Cin, const float* scale,
                                           const float* shift, const float* mean, const float* coef, float slope,
                                           const float* w_proj, float* h, float* dW, float* db, int accumulate,
                                           float* workspace, void* stream) {
  if (int e = check4(gin, gout, s, "as_conv4_wgrad_bnapply_proj")) return e;
  AS_CHECK_ARG(x4 && g_a && z && scale && shift && mean && coef && w_proj && h && dW && workspace && Cin >= 1 && Cin <= 4,
               "as_conv4_wgrad_bnapply_proj: bad argument");
  AS_CHECK_ARG(conv4_wgrad_rows_applicable(gin, gout, s), "as_conv4_wgrad_bnapply_proj: configuration not supported");
  return conv4_wgrad_rows_launch(x4, gin, g_a, gout, Cin, dW, db, accumulate, workspace, true, z, scale, shift, mean, coef,
                                 slope, nullptr, stream, w_proj, h);
}

// out[b][y][x] = residual[b][y][x] + sum_t h[b][t][y + t/3 - 1][x + t%3 - 1]   (zero outside the image)
__global__ __launch_bounds__(256) void tap_gather_kernel(const float* __restrict__ h, const float* __restrict__ residual,
                                                          float* __restrict__ out, int H, int W) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (x >= W) return;
  const long plane = (long)H * W;
  const float* hb = h + (long)b * 9 * plane;
  float acc = residual ? residual[(long)b * plane + (long)y * W + x] : 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) acc += hb[t * plane + (long)yy * W + xx];
  }
  out[(long)b * plane + (long)y * W + x] = acc;
}

extern "C" int as_tap_gather(const float* h, const float* residual, float* out, int B, int H, int W, void* stream) {
  AS_CHECK_ARG(h && out && B > 0 && H > 0 && W > 0 && H <= 65535 && B <= 65535, "as_tap_gather: bad argument");
  hipLaunchKernelGGL(tap_gather_kernel, dim3(as_div_up(W, 256), H, B), dim3(256), 0, (hipStream_t)stream, h, residual, out, H, W);
  AS_CHECK_LAUNCH("as_tap_gather");
  return AS_OK;
}

extern "C" int as_conv4_wgrad(const float* x4, const as_pcl* gin, const float* gz, const as_pcl* gout,
                              const as_conv_shape* s, int Cin, float* dW, float* db, int accumulate, float* workspace,
                              void* stream) {
  if (int e = check4(gin, gout, s, "as_conv4_wgrad")) return e;
  AS_CHECK_ARG(x4 && gz && dW && workspace && Cin >= 1 && Cin <= 4, "as_conv4_wgrad: bad argument");
  if (conv4_wgrad_rows_applicable(gin, gout, s))
    return conv4_wgrad_rows_launch(x4, gin, gz, gout, Cin, dW, db, accumulate, workspace, false, nullptr, nullptr, nullptr,
                                   nullptr, nullptr, 0.f, nullptr, stream);
  int nb, rpc, nchunks;
  const bool staged = g_conv4_s2 && plan4_staged_shape(s) && gin->ph >= 2 && gin->pw >= 2;
  plan4(gout, s, staged, &nb, &rpc, &nchunks);
  AS_CHECK_ARG(nb >= 1 && nb <= 4, "as_conv4_wgrad: kernel too large");
  Wgrad4Args a;
  a.x4 = x4; a.gz = gz; a.partial = workspace; a.partial_db = workspace + (int64_t)nchunks * nb * 1024;
  a.gin = as_make_dev(gin); a.gout = as_make_dev(gout);
  a.nseg = plan4_segments(gout);
  a.rows = gout->B * gout->H * a.nseg; a.rows_per_chunk = rpc; a.stride = s->stride;
  a.ntaps = fill_taps4(gin, s, a.tap_off);
  hipStream_t st = (hipStream_t)stream;
  if (staged) {
    // (the 5x5 stride-2 first layer of the feature towers: input rows staged in LDS)
    const size_t lds = (size_t)(3 * 4 * 16 * 64 + 4 * 32) * sizeof(float);      // >= 4 * C4W_WAVE_BYTES
    hipLaunchKernelGGL(conv4_s2_wgrad_kernel, dim3(nchunks), dim3(256), lds, st, a);
  } else
  switch (nb) {
    case 1: launch4<1>(a, nchunks, st); break;
    case 2: launch4<2>(a, nchunks, st); break;
    case 3: launch4<3>(a, nchunks, st); break;
    default: launch4<4>(a, nchunks, st); break;
  }
  AS_CHECK_LAUNCH("as_conv4_wgrad");
  const int T = a.ntaps;
  hipLaunchKernelGGL(conv4_wgrad_reduce_kernel, dim3(as_div_up(4 * T * 32 + 32, 64)), dim3(1024), 0, st, a.partial,
                     a.partial_db, nchunks, nb, T, Cin, dW, db, accumulate);
  AS_CHECK_LAUNCH("as_conv4_wgrad(reduce)");
  return AS_OK;
}
