// The strided head of the feature towers (stereo_net.py:59-72, 81-82): Conv2d(32, 32, 5, stride 2, padding 2), forward and
// data gradient, for maps large enough to fill the chip (the two smallest levels stay on the split-K kernels of
// conv32_mfma.hip).
//
// What the generic direct-load kernel (conv32_fwd_kernel<25>, conv32_dgrad_s2_kernel) pays for: every lane fetches the 16-byte
// K-chunks of ITS voxel per tap — one wave instruction touches 32 different 128-byte lines, 100 such instructions per 32-voxel
// tile, and the texture-address path (one line per cycle), not the matrix pipe, sets the pace (matrix pipe busy 0.41-0.43,
// 0.47 / 0.38 of the fp32 peak: round-3 PMC).  Here a wave fetches whole ROW SEGMENTS instead — 8 lanes per voxel, one
// contiguous KB per wave instruction — into a wave-PRIVATE LDS image and reads its A operands from there:
//   forward        output tile = 32 consecutive x of one output row; per ky the 72 input voxels 2x0-2 .. 2x0+69 of input row
//                  2y+ky-2 (9 instructions instead of 5 taps x 4), stored split by column parity (tap kx reads entry
//                  li + kx/2 of half kx&1: unit stride) with the 16-byte-slot swizzle of conv32_lds.hip (slot s of entry e
//                  holds chunk s ^ ((e>>1)&7): conflict-free ds_read_b128);
//   data gradient  tile = 32 consecutive x' of one row y' of the COARSE gradient; its three rows y'-1 .. y'+1 (40 voxels each)
//                  serve all four parity phases (9 + 6 + 6 + 4 taps) of the output pixels (2y'+py, 2x'+px).
// No workgroup barrier anywhere: a wave's LDS traffic is ordered by the LDS queue itself, the four waves of a workgroup only
// share the launch.  Weights stream from L1/L2 one tap ahead (one coalesced KB per instruction, as before).
// Same arithmetic per output as the generic kernels (taps in the same order, K in the same order): bit-identical results.
#include "as_common.h"
#include "conv32_s2.h"

struct S2Args {
  const float* x;        // forward: the input (PCL, halo >= 2); data gradient: the coarse gradient gz (PCL, halo >= 1)
  const float* wp;       // forward: [25][4][64][4] (as_conv32_pack_weights); data gradient: as_conv32_dgrad_s2_pack's 25 taps
  const float* bias;     // forward: [32] or null
  float* out;            // forward: z (PCL); data gradient: gx (PCL)
  PclDev gin, gout;      // geometry of x / of out
  int nseg, ntiles;      // 32-voxel segments per row; tiles of the launch
};

__device__ inline void s2_loadw(f32x4 (&r)[4], const float* p) {
  r[0] = *reinterpret_cast<const f32x4*>(p);
  r[1] = *reinterpret_cast<const f32x4*>(p + 256);
  r[2] = *reinterpret_cast<const f32x4*>(p + 512);
  r[3] = *reinterpret_cast<const f32x4*>(p + 768);
}
__device__ inline void s2_mfma16(f32x16& acc, const f32x4 (&a)[4], const f32x4 (&b)[4]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, b[q].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, b[q].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, b[q].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, b[q].w, acc, 0, 0, 0);
  }
}

// ---- forward -------------------------------------------------------------------------------------------------------
#define S2F_HALF (36 * 128)                  // bytes of one parity half: entries 0..35
#define S2F_BUF (2 * S2F_HALF)               // one staged row
#define S2F_WAVE (2 * S2F_BUF)               // two rows per wave (double buffer): 18,432 B

__global__ __launch_bounds__(256, 2) void conv32_s2_fwd_kernel(S2Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem_s2[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, li = lane & 31;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= p.ntiles) return;                             // (no barrier in this kernel: a wave may leave)
  const int seg = tile % p.nseg, row = tile / p.nseg;
  const int y = row % p.gout.H, b = row / p.gout.H;
  const int x0 = seg * 32;
  char* buf = smem_s2 + wave * S2F_WAVE;

  // staging: instruction i (0..8) brings staged voxels 8i + (lane >> 3), chunk lane & 7; staged voxel v = input column
  // 2 x0 - 2 + v; LDS: half v & 1, entry v >> 1
  const int sv = lane >> 3, sq = lane & 7;
  const int s_half = (sv & 1) * S2F_HALF;
  // input row 2y + ky - 2, first staged column 2 x0 - 2 (padded coordinates: + ph, + pw; the halo of 2 covers both).  The
  // last segment of a row runs past the row's padded end: those voxels only feed outputs that are not stored, and they are
  // CLAMPED to the row's last (halo) voxel — the last padded row of the last image has nothing behind it to read.
  const float* src0 = p.x + ((((long)b * p.gin.Hp + (2 * y - 2 + p.gin.ph)) * p.gin.Wp + (2 * x0 - 2 + p.gin.pw)) * 32) + sq * 4;
  const int vlim = p.gin.Wp - 1 - (2 * x0 - 2 + p.gin.pw);
  int voff[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) voff[i] = min(8 * i + sv, vlim) * 32;
  auto fetch = [&](f32x4 (&r)[9], int ky) {
    const float* src = src0 + (long)ky * p.gin.Wp * 32;
#pragma unroll
    for (int i = 0; i < 9; ++i) r[i] = *reinterpret_cast<const f32x4*>(src + voff[i]);
  };
  auto stage = [&](const f32x4 (&r)[9], char* dst) {
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int e = 4 * i + (sv >> 1);
      *reinterpret_cast<f32x4*>(dst + s_half + e * 128 + ((sq ^ ((e >> 1) & 7)) << 4)) = r[i];
    }
  };
  const float* wb = p.wp + lane * 4;
  const float bias_v = p.bias ? p.bias[li] : 0.f;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = bias_v;

  // (operands one tap ahead and the next row staged under this row's MFMAs were tried: 146 / 140 us against 136 / 137 at the
  //  bench workload — 34 more registers, no gain; the two resident workgroups per CU already cover each other)
  f32x4 ra[9], rb[9];
  f32x4 bw[2][4];
  fetch(ra, 0);
  s2_loadw(bw[0], wb);
#pragma unroll
  for (int ky = 0; ky < 5; ++ky) {
    char* cur = buf + (ky & 1) * S2F_BUF;
    if (ky & 1) stage(rb, cur); else stage(ra, cur);
    if (ky + 1 < 5) { if (ky & 1) fetch(ra, ky + 1); else fetch(rb, ky + 1); }    // the next row: in flight under this row's MFMAs
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) {
      const int tp = ky * 5 + kx;
      if (tp + 1 < 25) s2_loadw(bw[(tp + 1) & 1], wb + (tp + 1) * 1024);
      f32x4 a[4];
      const int e = li + (kx >> 1);
      const char* ap = cur + (kx & 1) * S2F_HALF + e * 128;
      const int sw = (e >> 1) & 7;
#pragma unroll
      for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const f32x4*>(ap + (((4 * h + q) ^ sw) << 4));
      __builtin_amdgcn_sched_barrier(0);
      s2_mfma16(acc, a, bw[tp & 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // epilogue: row (r, h) of the tile = output voxel x0 + (r & 3) + 8 (r >> 2) + 4h
  float* outp = p.out + ((((long)b * p.gout.Hp + (y + p.gout.ph)) * p.gout.Wp + (x0 + p.gout.pw)) * 32) + li;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int xo = (r & 3) + 8 * (r >> 2) + 4 * h;
    if (x0 + xo < p.gout.W) outp[xo * 32] = acc[r];
  }
}

bool conv32_s2_fwd_applicable(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s) {
  if (s->kd != 1 || s->kh != 5 || s->kw != 5 || s->stride != 2 || s->dil != 1 || s->pad_h != 2 || s->pad_w != 2) return false;
  if (gin->D != 1 || gout->D != 1 || gin->pd != 0 || gin->B != gout->B) return false;
  if (gout->H != (gin->H - 1) / 2 + 1 || gout->W != (gin->W - 1) / 2 + 1) return false;
  // the staged rows 2y-2 .. 2y+2 and columns 2x0-2 .. 2x0+69 must lie inside the padded tensor: a halo of 2 covers the rows
  // and the left edge; on the right the kernel clamps to the row's last voxel
  if (gin->ph < 2 || gin->pw < 2) return false;
  return (long)gout->B * gout->H * ((gout->W + 31) / 32) >= 1024;       // below: the split-K / generic kernels
}

int conv32_s2_fwd_launch(const float* x, const as_pcl* gin, const float* packed_w, const float* bias, float* z,
                         const as_pcl* gout, void* stream) {
  static AsPerDevice attr_set;
  if (!attr_set.get()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv32_s2_fwd_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 4 * S2F_WAVE);
    if (e != hipSuccess) { as_set_error("as_conv32_fwd: %s", hipGetErrorString(e)); return AS_ERR_LAUNCH; }
    attr_set.set();
  }
  S2Args a;
  a.x = x; a.wp = packed_w; a.bias = bias; a.out = z; a.gin = as_make_dev(gin); a.gout = as_make_dev(gout);
  a.nseg = (gout->W + 31) / 32; a.ntiles = gout->B * gout->H * a.nseg;
  hipLaunchKernelGGL(conv32_s2_fwd_kernel, dim3(as_div_up(a.ntiles, 4)), dim3(256), 4 * S2F_WAVE, (hipStream_t)stream, a);
  return AS_OK;
}

// ---- data gradient: gx[2y'+py][2x'+px] from gz rows y'-1 .. y'+1 -------------------------------------------------------
#define S2D_ROW (40 * 128)                   // 40 staged voxels: coarse columns x0 - 1 .. x0 + 38
#define S2D_WAVE (3 * S2D_ROW)               // 15,360 B per wave

__global__ __launch_bounds__(256, 2) void conv32_s2_dgrad_kernel(S2Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem_s2[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, li = lane & 31;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= p.ntiles) return;
  const int seg = tile % p.nseg, row = tile / p.nseg;
  const int yc = row % p.gin.H, b = row / p.gin.H;          // coarse row y'
  const int x0 = seg * 32;                                  // first coarse column x'
  char* buf = smem_s2 + wave * S2D_WAVE;

  // stage gz rows y'-1, y', y'+1, coarse columns x0 - 1 + v, v = 0..39 (the halo of 1 covers rows and the left edge)
  const int sv = lane >> 3, sq = lane & 7;
  // (voxels past the row's padded end — the last segment — are clamped to its last voxel, as in the forward kernel)
  const float* src0 = p.x + ((((long)b * p.gin.Hp + (yc - 1 + p.gin.ph)) * p.gin.Wp + (x0 - 1 + p.gin.pw)) * 32) + sq * 4;
  const int vlim = p.gin.Wp - 1 - (x0 - 1 + p.gin.pw);
  int voff[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) voff[i] = min(8 * i + sv, vlim) * 32;
  f32x4 r[3][5];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int i = 0; i < 5; ++i) r[dy][i] = *reinterpret_cast<const f32x4*>(src0 + (long)dy * p.gin.Wp * 32 + voff[i]);
  const float* wb = p.wp + lane * 4;
  f32x4 bw[2][4];
  s2_loadw(bw[0], wb);
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int e = 8 * i + sv;
      *reinterpret_cast<f32x4*>(buf + dy * S2D_ROW + e * 128 + ((sq ^ ((e >> 1) & 7)) << 4)) = r[dy][i];
    }

  // phases in as_conv32_dgrad_s2_pack's order: (py, px) = (0,0) 9 taps, (0,1) 6, (1,0) 6, (1,1) 4; within a phase j = py, py+2,
  // .. outer, l = px, px+2, .. inner; tap (j, l) reads gz[y' + (py+2-j)/2][x' + (px+2-l)/2]
  int tp = 0;
#pragma unroll
  for (int ph = 0; ph < 4; ++ph) {
    const int py = ph >> 1, px = ph & 1;
    f32x16 acc;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) acc[rr] = 0.f;
#pragma unroll
    for (int j = py; j < 5; j += 2)
#pragma unroll
      for (int l = px; l < 5; l += 2, ++tp) {
        if (tp + 1 < 25) s2_loadw(bw[(tp + 1) & 1], wb + (tp + 1) * 1024);
        const int dy = (py + 2 - j) / 2 + 1, dx = (px + 2 - l) / 2;     // staged row 0..2; column offset -1..1
        const int e = li + 1 + dx;
        const char* ap = buf + dy * S2D_ROW + e * 128;
        const int sw = (e >> 1) & 7;
        f32x4 a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const f32x4*>(ap + (((4 * h + q) ^ sw) << 4));
        __builtin_amdgcn_sched_barrier(0);
        s2_mfma16(acc, a, bw[tp & 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    // this phase's outputs: (2y'+py, 2(x0 + xo) + px)
    const int yo = 2 * yc + py;
    if (yo < p.gout.H) {
      float* outp = p.out + ((((long)b * p.gout.Hp + (yo + p.gout.ph)) * p.gout.Wp + (2 * x0 + px + p.gout.pw)) * 32) + li;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int xo = (rr & 3) + 8 * (rr >> 2) + 4 * h;
        if (2 * (x0 + xo) + px < p.gout.W) outp[xo * 64] = acc[rr];
      }
    }
  }
}

bool conv32_s2_dgrad_applicable(const as_pcl* ggz, const as_pcl* ggx) {
  if (ggz->D != 1 || ggx->D != 1 || ggz->B != ggx->B || ggz->pd != 0) return false;
  if (ggz->H != (ggx->H - 1) / 2 + 1 || ggz->W != (ggx->W - 1) / 2 + 1) return false;
  if (ggz->ph < 1 || ggz->pw < 1) return false;          // rows y'-1 .. y'+1 and the left edge; the right edge is clamped
  return (long)ggz->B * ggz->H * ((ggz->W + 31) / 32) >= 1024;
}

int conv32_s2_dgrad_launch(const float* gz, const as_pcl* ggz, const float* packed, float* gx, const as_pcl* ggx, void* stream) {
  static AsPerDevice attr_set;
  if (!attr_set.get()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv32_s2_dgrad_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 4 * S2D_WAVE);
    if (e != hipSuccess) { as_set_error("as_conv32_dgrad_s2: %s", hipGetErrorString(e)); return AS_ERR_LAUNCH; }
    attr_set.set();
  }
  S2Args a;
  a.x = gz; a.wp = packed; a.bias = nullptr; a.out = gx; a.gin = as_make_dev(ggz); a.gout = as_make_dev(ggx);
  a.nseg = (ggz->W + 31) / 32; a.ntiles = ggz->B * ggz->H * a.nseg;
  hipLaunchKernelGGL(conv32_s2_dgrad_kernel, dim3(as_div_up(a.ntiles, 4)), dim3(256), 4 * S2D_WAVE, (hipStream_t)stream, a);
  return AS_OK;
}
