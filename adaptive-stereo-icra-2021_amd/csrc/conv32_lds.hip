// conv32, 2-D 3x3 stride-1 instance (any dilation <= 8): the refinement's and the feature trunk's
// 32->32 layers and their data gradients — 83 % of the model's FLOPs (stereo_net.py:10-18, 33-51, 97).
//
// MI355X design
//   * persistent workgroups, TWO per CU (8 waves = two per SIMD), walk 128-pixel row segments;
//   * the 3 input rows a segment needs (y-d, y, y+d; x0-8 .. x0+135) are DMA'd into LDS
//     (global_load_lds_dwordx4: 1 KB per wave instruction, no VGPR round trip): HBM/L2 sees each
//     input voxel 3 times per layer instead of 9, as full 128-byte lines;
//   * one tile buffer per workgroup: while one workgroup waits for its DMA or runs its epilogue
//     (stores, residual, BatchNorm moments) the other one owns the matrix cores — the overlap comes
//     from the second workgroup, not from double buffering.  (A double-buffered single-workgroup
//     variant with LDS-resident weights measured 69 TFLOP/s: hipcc places vmcnt(0) waits in the
//     compute phase that also drain an in-flight DMA, and the epilogue is fully exposed.)
//   * LDS image is lane-linear (a DMA cannot pad), so bank conflicts are removed by swizzling the
//     SOURCE: LDS slot s of voxel v holds channel chunk s ^ ((v>>1)&7); a ds_read_b128 of one chunk
//     from 16 consecutive voxels then hits 16 distinct 16-byte bank groups;
//   * weights are read from L2 in the [tap][q][lane][4] packing: every wave load is one contiguous KB;
//   * tiles are banded per XCD (workgroup b is on XCD b%8 under round-robin dispatch — speed only):
//     the three uses of an input row happen close in time on one XCD's L2;
//   * BatchNorm moments of a workgroup's tiles are Chan-merged in registers: one partial per
//     workgroup (512 per layer instead of 15,000).
// LDS per workgroup: 55,296 (tile) + 640; two workgroups per CU use 112 KB of the CU's 160 KB.
#include "as_common.h"
#include "conv_epilogue.h"
#include "conv32_lds.h"

#define TL_W 144                       // staged voxels per row: 8 + 128 + 8
#define TL_ROW_BYTES (TL_W * 128)
#define TL_BUF_BYTES (3 * TL_ROW_BYTES)
#define TL_LDS_BYTES (TL_BUF_BYTES + 1024)
#define TL_DMA_PER_ROW (TL_W / 8)      // 18 wave instructions of 8 voxels
#define TL_MAX_WG 512                  // two resident workgroups per CU

struct ConvLdsArgs {
  const float* x;
  const float* wq;                     // [9][4][64][4]
  EpilogueArgs ep;
  PclDev gin, gout;
  int dil, tiles_per_row, ntiles, tiles_per_band, wg_per_xcd;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ inline void tile_coords(const ConvLdsArgs& p, int tile, int& b, int& y, int& x0) {
  const int row = tile / p.tiles_per_row;
  x0 = (tile - row * p.tiles_per_row) * 128;
  b = row / p.gout.H;
  y = row - b * p.gout.H;
}

// Stage the three rows of `tile` into `buf` (LDS byte address).  54 wave instructions, 13-14 per wave.
__device__ inline void issue_tile_dma(const ConvLdsArgs& p, int tile, char* buf, int wave, int lane) {
  int b, y, x0;
  tile_coords(p, tile, b, y, x0);
  const int Wp = p.gin.Wp;
  const int px0 = x0 - 8 + p.gin.pw;             // padded x of LDS column 0
  const int vl = lane >> 3, slot = lane & 7;
  for (int idx = wave; idx < 3 * TL_DMA_PER_ROW; idx += 4) {
    const int r = idx / TL_DMA_PER_ROW, i = idx - r * TL_DMA_PER_ROW;
    const int v = 8 * i + vl;
    const int chunk = slot ^ ((v >> 1) & 7);
    const long rowvox = ((long)b * p.gin.Hp + (y + p.gin.ph + (r - 1) * p.dil)) * Wp;
    const int px = min(px0 + v, Wp - 1);         // the last tile of a row may overhang: stay inside the buffer
    const float* src = p.x + (rowvox + px) * 32 + chunk * 4;
    char* dst = buf + (r * TL_W + 8 * i) * 128;  // wave-uniform; the hardware adds lane*16
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
  }
}

__device__ inline void lds_load_a(f32x4 (&a)[4], const char* buf, int r, int v, int h) {
  const int sw = (v >> 1) & 7;
  const char* base = buf + (r * TL_W + v) * 128;
#pragma unroll
  for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const f32x4*>(base + (((4 * h + q) ^ sw) << 4));
}

__device__ inline void glb_load_w(f32x4 (&r)[4], const float* p) {
  r[0] = *reinterpret_cast<const f32x4*>(p);
  r[1] = *reinterpret_cast<const f32x4*>(p + 256);
  r[2] = *reinterpret_cast<const f32x4*>(p + 512);
  r[3] = *reinterpret_cast<const f32x4*>(p + 768);
}

__device__ inline void mfma16l(f32x16& acc, const f32x4 (&a)[4], const f32x4 (&b)[4]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, b[q].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, b[q].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, b[q].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, b[q].w, acc, 0, 0, 0);
  }
}

__global__ __launch_bounds__(256, 2) void conv32_lds_kernel(ConvLdsArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tile_buf = smem;
  float (*red)[32] = reinterpret_cast<float (*)[32]>(smem + TL_BUF_BYTES);
  float* bmean = reinterpret_cast<float*>(smem + TL_BUF_BYTES + 512);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, li = lane & 31;
  const float* wb = p.wq + lane * 4;

  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int t_begin = xcd * p.tiles_per_band;
  const int t_end = min(t_begin + p.tiles_per_band, p.ntiles);
  TileStats run; run.n = 0.f; run.mean = 0.f; run.m2 = 0.f;
  const float bias_v = p.ep.bias ? p.ep.bias[li] : 0.f;

  int tile = t_begin + j;
  if (tile < t_end) issue_tile_dma(p, tile, tile_buf, wave, lane);
  for (int it = 0; tile < t_end; tile += p.wg_per_xcd, ++it) {
    int b, y, x0;
    tile_coords(p, tile, b, y, x0);
    const int x = x0 + 32 * wave + li;
    const bool valid = x < p.gout.W;
    const int out_vox = (int)p.gout.vox(b, 0, y, valid ? x : p.gout.W - 1);
    const int vbase = 32 * wave + li + 8;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias_v;
    f32x4 a[2][4], w[2][4];
    // Wait for this wave's share of the tile's DMA — but not for the previous tile's 16 output stores,
    // which were issued AFTER that DMA: vmcnt counts stores too, and waiting for their HBM
    // acknowledgement would expose a 1-2 us drain on every tile.  (vmcnt is in-order: "at most 16
    // outstanding" means everything older than the 16 youngest operations has completed.)
    if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else         asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __syncthreads();
    glb_load_w(w[0], wb);
    lds_load_a(a[0], tile_buf, 0, vbase - p.dil, h);
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      if (tp + 1 < 9) {
        const int r = (tp + 1) / 3, kx = (tp + 1) % 3;
        lds_load_a(a[(tp + 1) & 1], tile_buf, r, vbase + (kx - 1) * p.dil, h);
        glb_load_w(w[(tp + 1) & 1], wb + (tp + 1) * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma16l(acc, a[tp & 1], w[tp & 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // every wave has read its operands: the tile buffer is free.  Start the next tile's DMA BEFORE the
    // epilogue's stores, so that it is older than they are in the vmcnt queue.
    __syncthreads();
    const int next_tile = tile + p.wg_per_xcd;
    if (next_tile < t_end) issue_tile_dma(p, next_tile, tile_buf, wave, lane);
    __builtin_amdgcn_sched_barrier(0);
    TileStats ts; ts.n = 0.f; ts.mean = 0.f; ts.m2 = 0.f;
    conv_epilogue<true>(acc, p.ep, out_vox, valid, min(128, p.gout.W - x0), red, bmean, &ts);
    if (p.ep.stat_mean != nullptr && threadIdx.x < 32) stats_merge(run, ts);
  }
  stats_write(p.ep, blockIdx.x, run);
}

// ---- host ---------------------------------------------------------------------------------------
bool conv32_lds_applicable(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s) {
  if (s->kd != 1 || s->kh != 3 || s->kw != 3 || s->stride != 1) return false;
  if (gin->D != 1 || gout->D != 1) return false;
  if (s->dil < 1 || s->dil > 8 || s->pad_h != s->dil || s->pad_w != s->dil) return false;
  if (gin->H != gout->H || gin->W != gout->W) return false;
  if (gin->pw < 8 || gin->ph < s->dil) return false;      // the staged tile always spans x0-8 .. x0+135
  return true;
}

static inline int lds_ntiles(const as_pcl* gout) { return gout->B * gout->H * ((gout->W + 127) / 128); }

int conv32_lds_grid(const as_pcl* gout) {
  const int nt = lds_ntiles(gout);
  int g = (nt + 7) / 8 * 8;
  if (g > TL_MAX_WG) g = TL_MAX_WG;
  return g;
}

int conv32_lds_launch(const float* x, const as_pcl* gin, const float* packed_w, const float* bias,
                      float* z, const as_pcl* gout, const as_conv_shape* s,
                      int epilogue, const float* ep_scale, const float* ep_shift, float slope,
                      const float* residual, float* stat_mean, float* stat_m2, float* stat_cnt, void* stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv32_lds_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, TL_LDS_BYTES);
    if (e != hipSuccess) {
      as_set_error("conv32_lds: cannot reserve %d bytes of LDS: %s", TL_LDS_BYTES, hipGetErrorString(e));
      return AS_ERR_LAUNCH;
    }
    attr_set = true;
  }
  ConvLdsArgs a;
  a.x = x; a.wq = packed_w;
  a.ep.bias = bias; a.ep.z = z; a.ep.ep_scale = ep_scale; a.ep.ep_shift = ep_shift; a.ep.residual = residual;
  a.ep.stat_mean = epilogue == 0 ? stat_mean : nullptr; a.ep.stat_m2 = epilogue == 0 ? stat_m2 : nullptr;
  a.ep.stat_cnt = epilogue == 0 ? stat_cnt : nullptr;
  a.ep.epilogue = epilogue; a.ep.slope = slope;
  a.gin = as_make_dev(gin); a.gout = as_make_dev(gout);
  a.dil = s->dil;
  a.tiles_per_row = (gout->W + 127) / 128;
  a.ntiles = lds_ntiles(gout);
  const int grid = conv32_lds_grid(gout);
  a.tiles_per_band = (a.ntiles + 7) / 8;
  a.wg_per_xcd = grid / 8;
  hipStream_t st = (hipStream_t)stream;
  as_prof_mark(2, st, 1, 0.0);
  hipLaunchKernelGGL(conv32_lds_kernel, dim3(grid), dim3(256), TL_LDS_BYTES, st, a);
  as_prof_mark(2, st, 0, 2.0 * (double)gout->B * gout->H * gout->W * 1024.0 * 9);
  AS_CHECK_LAUNCH("as_conv32_fwd(lds)");
  return AS_OK;
}

// =====================================================================================================
// Weight gradient, same instance:  dW[tap][ci][co] = sum_v X[v + off(tap)][ci] * G[v][co].
// MFMA rows i = ci, columns j = co, reduction k = voxel.  The direct-load kernel (conv32_mfma.hip) issues
// one 256-byte wave load per operand per MFMA (1.33 per MFMA with 3 taps per wave): the CU's vector-memory
// instruction rate, not bytes, limited it to 58 TFLOP/s, and each tap group re-fetched its rows (2.9x the
// algorithmic HBM bytes).  Here a persistent workgroup stages the three X rows and the G row of a
// 128-pixel segment in LDS by DMA (same swizzle as the forward kernel), every wave takes 32 of the
// segment's voxels for ALL NINE taps (9 independent accumulators = 144 registers, so the MFMA chain
// never waits on itself), operands are conflict-free ds_read_b32 (32 lanes = the 32 channels of one
// voxel line), and the accumulators live in registers across all of the workgroup's tiles: one slab per
// workgroup, summed in fixed order by wgrad_reduce_kernel (deterministic, no float atomics).
#define TLG_BYTES (128 * 128)                           // G row segment
#define TLW_LDS_BYTES (TL_BUF_BYTES + TLG_BYTES)        // 71,680 B; two workgroups per CU

struct WgradLdsArgs {
  const float* x;
  const float* gz;
  float* partial;      // [wgs][9][32][32]
  float* partial_db;   // [wgs][32]
  PclDev gin, gout;
  int dil, tiles_per_row, ntiles, tiles_per_band, wg_per_xcd;
};

__device__ inline void issue_wgrad_dma(const WgradLdsArgs& p, int tile, char* xbuf, char* gbuf, int wave, int lane) {
  const int row = tile / p.tiles_per_row;
  const int x0 = (tile - row * p.tiles_per_row) * 128;
  const int b = row / p.gout.H, y = row - b * p.gout.H;
  const int vl = lane >> 3, slot = lane & 7;
  // X: three rows of 144 voxels; G: one row of 128 voxels (16 more instructions)
  for (int idx = wave; idx < 3 * TL_DMA_PER_ROW + 16; idx += 4) {
    const float* src; char* dst;
    if (idx < 3 * TL_DMA_PER_ROW) {
      const int r = idx / TL_DMA_PER_ROW, i = idx - r * TL_DMA_PER_ROW;
      const int v = 8 * i + vl;
      const int chunk = slot ^ ((v >> 1) & 7);
      const long rowvox = ((long)b * p.gin.Hp + (y + p.gin.ph + (r - 1) * p.dil)) * p.gin.Wp;
      const int px = min(x0 - 8 + p.gin.pw + v, p.gin.Wp - 1);
      src = p.x + (rowvox + px) * 32 + chunk * 4;
      dst = xbuf + (r * TL_W + 8 * i) * 128;
    } else {
      const int i = idx - 3 * TL_DMA_PER_ROW;
      const int v = 8 * i + vl;
      const int chunk = slot ^ ((v >> 1) & 7);
      const long rowvox = ((long)b * p.gout.Hp + (y + p.gout.ph)) * p.gout.Wp;
      const int px = min(x0 + p.gout.pw + v, p.gout.Wp - 1);   // beyond W: halo voxels, which are zero
      src = p.gz + (rowvox + px) * 32 + chunk * 4;
      dst = gbuf + (8 * i) * 128;
    }
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
  }
}

// one float of voxel v's 128-byte line (swizzled): channel c
__device__ inline float lds_chan(const char* rowbase, int v, int c) {
  const int slot = (c >> 2) ^ ((v >> 1) & 7);
  return *reinterpret_cast<const float*>(rowbase + v * 128 + slot * 16 + (c & 3) * 4);
}

__global__ __launch_bounds__(256, 2) void conv32_wgrad_lds_kernel(WgradLdsArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xbuf = smem;
  char* gbuf = smem + TL_BUF_BYTES;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, li = lane & 31;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int t_begin = xcd * p.tiles_per_band;
  const int t_end = min(t_begin + p.tiles_per_band, p.ntiles);
  for (int tile = t_begin + j; tile < t_end; tile += p.wg_per_xcd) {
    issue_wgrad_dma(p, tile, xbuf, gbuf, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // this wave's 32 voxels of the segment: 16 steps of one voxel pair (lane half h picks the voxel)
    float bv[2], av[2][9];
    {
      const int v = 32 * wave + h;
      bv[0] = lds_chan(gbuf, v, li);
#pragma unroll
      for (int t = 0; t < 9; ++t) av[0][t] = lds_chan(xbuf + (t / 3) * TL_ROW_BYTES, v + 8 + (t % 3 - 1) * p.dil, li);
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (s + 1 < 16) {
        const int v = 32 * wave + 2 * (s + 1) + h;
        bv[(s + 1) & 1] = lds_chan(gbuf, v, li);
#pragma unroll
        for (int t = 0; t < 9; ++t)
          av[(s + 1) & 1][t] = lds_chan(xbuf + (t / 3) * TL_ROW_BYTES, v + 8 + (t % 3 - 1) * p.dil, li);
      }
      __builtin_amdgcn_sched_barrier(0);
      bsum += bv[s & 1];
#pragma unroll
      for (int t = 0; t < 9; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1][t], bv[s & 1], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();          // all waves are done with the tile before the next DMA overwrites it
  }

  // reduce the four waves' accumulators through LDS, three taps per round (fixed order w0+w1+w2+w3)
  float* slab = reinterpret_cast<float*>(smem);           // [3 waves][3 taps][16][64] floats = 36,864 B
  float* out = p.partial + (long)blockIdx.x * 9 * 1024;
#pragma unroll
  for (int round = 0; round < 3; ++round) {
    if (wave > 0) {
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) slab[(((wave - 1) * 3 + g) * 16 + r) * 64 + lane] = acc[round * 3 + g][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[round * 3 + g][r];
          v += slab[((0 * 3 + g) * 16 + r) * 64 + lane];
          v += slab[((1 * 3 + g) * 16 + r) * 64 + lane];
          v += slab[((2 * 3 + g) * 16 + r) * 64 + lane];
          const int ci = (r & 3) + 8 * (r >> 2) + 4 * h;
          out[(round * 3 + g) * 1024 + ci * 32 + li] = v;
        }
    }
    __syncthreads();
  }
  float* dbs = reinterpret_cast<float*>(smem);
  bsum += __shfl_xor(bsum, 32, 64);
  if (h == 0) dbs[wave * 32 + li] = bsum;
  __syncthreads();
  if (threadIdx.x < 32) p.partial_db[blockIdx.x * 32 + li] = dbs[li] + dbs[32 + li] + dbs[64 + li] + dbs[96 + li];
}

int conv32_wgrad_lds_slabs(const as_pcl* gout) {
  // every workgroup ends with a 36 KB slab (+ its share of the final reduce): give each at least 16 tiles
  const int nt = lds_ntiles(gout);
  int g = ((nt + 15) / 16 + 7) / 8 * 8;
  if (g > 512) g = 512;
  return g;
}

int conv32_wgrad_lds_launch(const float* x, const as_pcl* gin, const float* gz, const as_pcl* gout,
                            const as_conv_shape* s, float* partial, float* partial_db, void* stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv32_wgrad_lds_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, TLW_LDS_BYTES);
    if (e != hipSuccess) {
      as_set_error("conv32_wgrad_lds: cannot reserve %d bytes of LDS: %s", TLW_LDS_BYTES, hipGetErrorString(e));
      return AS_ERR_LAUNCH;
    }
    attr_set = true;
  }
  WgradLdsArgs a;
  a.x = x; a.gz = gz; a.partial = partial; a.partial_db = partial_db;
  a.gin = as_make_dev(gin); a.gout = as_make_dev(gout);
  a.dil = s->dil;
  a.tiles_per_row = (gout->W + 127) / 128;
  a.ntiles = lds_ntiles(gout);
  const int grid = conv32_wgrad_lds_slabs(gout);
  a.tiles_per_band = (a.ntiles + 7) / 8;
  a.wg_per_xcd = grid / 8;
  hipLaunchKernelGGL(conv32_wgrad_lds_kernel, dim3(grid), dim3(256), TLW_LDS_BYTES, (hipStream_t)stream, a);
  AS_CHECK_LAUNCH("as_conv32_wgrad(lds)");
  return AS_OK;
}
