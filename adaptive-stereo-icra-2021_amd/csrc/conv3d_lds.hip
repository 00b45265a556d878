// conv32, 3-D 3x3x3 stride-1 instance: the four cost-aggregation layers (stereo_net.py:21-30, 155-161, 185-186)
// and their data gradients — the path's one genuinely dense contraction over a volume (M = B*Dc*Hc*Wc voxels,
// N = 32, K = 27*32).
//
// Why not the direct-load kernel (conv32_mfma.hip): there every wave fetches its A operand itself, 16 B per lane
// from 32 different 128-byte lines per instruction, 27 taps over — the vector-memory path, not the matrix pipe,
// sets the pace (70-80 TFLOP/s).  Here the operand goes through LDS once per workgroup, fully coalesced:
//   * a PCL plane is ONE contiguous run of (H+2)*(W+2) voxels, so "position" p = y*Wp + x (padded coordinates)
//     turns every (kh, kw) tap into the constant offset (kh-1)*Wp + (kw-1): a tile is 128 consecutive positions of
//     one plane, its operand for one kd is the contiguous run [p0 - Wp - 1, p0 + 128 + Wp + 1) of plane d+kd-1 —
//     a plain 1-KB-per-instruction LDS-DMA stream (global_load_lds_dwordx4), no row bookkeeping at all;
//   * positions in the two halo columns are computed like any other and simply not stored (the halo must stay
//     zero); tiles run from the first to the last interior voxel of a plane, the last one shifted back to end there
//     (as in conv32_lds.hip), so the staged run never leaves the plane: no out-of-bounds read even for the first
//     and last plane of the buffer;
//   * three stages (kd = 0, 1, 2) of 9 taps x 16 MFMAs accumulate into the same registers; LDS image and
//     source-side swizzle as in conv32_lds.hip (slot s of voxel v holds chunk s ^ ((v>>1)&7)): conflict-free
//     ds_read_b128 for any tap offset;
//   * weights (27 x 4 KB) are streamed from L2 two taps ahead, one coalesced KB per wave instruction; the LDS operand
//     is read one tap ahead;
//   * one buffer per workgroup, 37 KB at W = 78: three workgroups per CU cover each other's DMA waits.
// Measured (12x24x78 per pair): 57 us = 88 TFLOP/s at 4 pairs (the whole launch is one round of 720 workgroups; the
// direct-load kernel took 68 us), 100 TFLOP/s at 16 pairs, 110 at 64 (direct-load: 80).  A start-up stagger between
// co-resident workgroups changes nothing.
#include "as_common.h"
#include "conv_epilogue.h"
#include "conv3d_lds.h"

struct Conv3dLdsArgs {
  const float* x;
  const float* wq;                 // packed [27][4][64][4]
  EpilogueArgs ep;
  PclDev g;                        // input and output share the padded geometry
  int tiles_per_plane, npos;       // npos = (H-1)*Wp + W positions from the first to the last interior voxel
  int run, groups;                 // staged voxels (130 + 2*Wp) and 8-voxel DMA groups (the last one may overlap)
};

typedef __attribute__((address_space(3))) void* lds3_ptr_t;

// One 1-KB LDS-DMA instruction (see conv32_lds.hip for why this is inline asm).
__device__ inline void dma3_1kb(const float* sbase, unsigned voff, unsigned m0) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               :: "s"(m0), "v"(voff), "s"(sbase) : "memory", "m0");
}

__device__ inline f32x4 lds3_chunk(const char* buf, int v, int h, int q) {
  return *reinterpret_cast<const f32x4*>(buf + v * 128 + (((4 * h + q) ^ ((v >> 1) & 7)) << 4));
}

__device__ inline void load_w3(f32x4 (&r)[4], const float* p) {
  r[0] = *reinterpret_cast<const f32x4*>(p);
  r[1] = *reinterpret_cast<const f32x4*>(p + 256);
  r[2] = *reinterpret_cast<const f32x4*>(p + 512);
  r[3] = *reinterpret_cast<const f32x4*>(p + 768);
}

__global__ __launch_bounds__(256) void conv3d_lds_kernel(Conv3dLdsArgs p) {
  extern __shared__ __attribute__((aligned(16))) char tile_buf[];
  __shared__ float red[4][32];
  __shared__ float bmean[32];
  __shared__ int s_valid[4];
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((lds3_ptr_t)tile_buf));
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, li = lane & 31;
  const int Wp = p.g.Wp;

  const int tile = blockIdx.x;
  const int plane = tile / p.tiles_per_plane, t = tile - plane * p.tiles_per_plane;
  const int b = plane / p.g.D, d = plane - b * p.g.D;
  const int first = p.g.ph * Wp + p.g.pw;                        // position of the first interior voxel
  const int pos_new = first + 128 * t;                            // first position that is this tile's alone
  const int pos0 = min(pos_new, first + p.npos - 128);            // the last tile is shifted back to end at the last voxel
  const long plane_vox = (long)Wp * p.g.Hp;
  const long out_plane = ((long)b * p.g.Dp + d + p.g.pd) * plane_vox;

  // DMA lane constants.  A regular group i covers LDS voxels [8i, 8i+8): voxel vl = lane>>3 of it, slot s = lane&7,
  // source chunk s ^ ((v>>1)&7) with (v>>1)&7 = ((i&1)<<2) | (vl>>1); wave w takes groups i = w, w+4, .. (same parity).
  // The last group is placed to END at the run's end (it overlaps its predecessor with identical data when the run
  // is not a multiple of 8): its swizzle phase is its own.
  const unsigned vl = (unsigned)(lane >> 3), sl = (unsigned)(lane & 7);
  const unsigned off_reg = vl * 128u + ((sl ^ ((((unsigned)wave & 1u) << 2) | (vl >> 1))) << 4);
  const unsigned tail_v0 = (unsigned)(p.run - 8);
  const unsigned off_tail = vl * 128u + ((sl ^ ((((tail_v0 + vl) >> 1)) & 7u)) << 4);

  const float* wb = p.wq + lane * 4;
  const float bias_v = p.ep.bias ? p.ep.bias[li] : 0.f;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = bias_v;

  // Software pipeline over the 27 taps: weights two taps ahead (a ring of three register sets, streamed from L2), the
  // LDS operand one tap ahead (two sets).  Stage boundaries only interrupt the operand side.
  f32x4 bw[3][4], a[2][4];
  load_w3(bw[0], wb);
  load_w3(bw[1], wb + 1024);
  const int vbase = 32 * wave + li;                               // LDS voxel of tap (kh=0, kw=0) for this lane's position
#pragma unroll
  for (int kd = 0; kd < 3; ++kd) {
    __syncthreads();                                              // the previous stage's operand reads are done
    const float* src = p.x + (((long)b * p.g.Dp + d + p.g.pd + kd - 1) * plane_vox + (pos0 - Wp - 1)) * 32;
    for (int i = wave; i < p.groups - 1; i += 4)
      dma3_1kb(src + i * 256, off_reg, lds0 + (unsigned)(i * 1024));
    if (((p.groups - 1) & 3) == wave)
      dma3_1kb(src + (long)tail_v0 * 32, off_tail, lds0 + tail_v0 * 128u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) a[0][q] = lds3_chunk(tile_buf, vbase, h, q);
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      const int tap = kd * 9 + tp;
      if (tap + 2 < 27) load_w3(bw[(tap + 2) % 3], wb + (tap + 2) * 1024);
      if (tp + 1 < 9) {
        const int v = vbase + ((tp + 1) / 3) * Wp + ((tp + 1) % 3);
#pragma unroll
        for (int q = 0; q < 4; ++q) a[(tp + 1) & 1][q] = lds3_chunk(tile_buf, v, h, q);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 av = a[tp & 1][q], bv = bw[tap % 3][q];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // Epilogue: lane li of each half owns position pos0 + 32*wave + li.  Stored: interior columns that are not a
  // duplicate of the previous tile's (shifted last tile); the BatchNorm moments cover exactly the stored voxels.
  const int pos = pos0 + 32 * wave + li;
  const int yp = pos / Wp, xp = pos - yp * Wp;
  const bool valid = xp >= p.g.pw && xp < p.g.pw + p.g.W && pos >= pos_new;
  const unsigned long long m = __ballot(valid && h == 0);
  if (lane == 0) s_valid[wave] = __popcll(m);
  __syncthreads();
  const int nvalid = s_valid[0] + s_valid[1] + s_valid[2] + s_valid[3];
  TileStats ts; ts.n = 0.f; ts.mean = 0.f; ts.m2 = 0.f;
  conv_epilogue(acc, p.ep, (int)(out_plane + pos), valid, max(nvalid, 1), red, bmean, &ts);
  if (p.ep.epilogue == 0 && p.ep.stat_mean != nullptr) {
    if (nvalid == 0) { ts.n = 0.f; ts.mean = 0.f; ts.m2 = 0.f; }
    stats_write(p.ep, blockIdx.x, ts);
  }
}

// ---- host ---------------------------------------------------------------------------------------
static int conv3d_run(const as_pcl* g) { return 130 + 2 * (g->W + 2 * g->pw); }

bool conv3d_lds_applicable(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s) {
  if (!(s->kd == 3 && s->kh == 3 && s->kw == 3 && s->stride == 1 && s->dil == 1 && s->pad_d == 1 && s->pad_h == 1 && s->pad_w == 1))
    return false;
  if (gin->B != gout->B || gin->D != gout->D || gin->H != gout->H || gin->W != gout->W) return false;
  if (gin->pd != gout->pd || gin->ph != gout->ph || gin->pw != gout->pw) return false;    // shared padded positions
  if (gin->pd < 1 || gin->ph < 1 || gin->pw < 1) return false;
  const int Wp = gin->W + 2 * gin->pw;
  if (Wp < 34) return false;                                     // a wave tile (32 positions) spans at most two rows
  if ((long)(gin->H - 1) * Wp + gin->W < 128) return false;     // at least one full tile per plane
  return (long)((conv3d_run(gin) + 7) / 8) * 1024 <= 64 * 1024;  // >= 2 workgroups per CU
}

static int conv3d_tiles_per_plane(const as_pcl* g) {
  const int Wp = g->W + 2 * g->pw;
  return as_div_up((int64_t)(g->H - 1) * Wp + g->W, 128);
}

int conv3d_lds_grid(const as_pcl* gout) { return gout->B * gout->D * conv3d_tiles_per_plane(gout); }

int conv3d_lds_launch(const float* x, const as_pcl* gin, const float* packed_w, const float* bias, float* z,
                      const as_pcl* gout, int epilogue, const float* ep_scale, const float* ep_shift, float slope,
                      const float* residual, float* stat_mean, float* stat_m2, float* stat_cnt, void* stream) {
  Conv3dLdsArgs a;
  a.x = x; a.wq = packed_w;
  a.ep.bias = bias; a.ep.z = z; a.ep.ep_scale = ep_scale; a.ep.ep_shift = ep_shift; a.ep.residual = residual;
  a.ep.stat_mean = epilogue == 0 ? stat_mean : nullptr; a.ep.stat_m2 = epilogue == 0 ? stat_m2 : nullptr;
  a.ep.stat_cnt = epilogue == 0 ? stat_cnt : nullptr;
  a.ep.epilogue = epilogue; a.ep.slope = slope;
  a.g = as_make_dev(gin);
  a.tiles_per_plane = conv3d_tiles_per_plane(gin);
  a.npos = (gin->H - 1) * a.g.Wp + gin->W;
  a.run = conv3d_run(gin);
  a.groups = (a.run + 7) / 8;
  const int lds_bytes = a.groups * 1024;
  static AsPerDevice attr_set;
  if (!attr_set.get()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_lds_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    if (e != hipSuccess) { as_set_error("as_conv32_fwd(3-D LDS): %s", hipGetErrorString(e)); return AS_ERR_LAUNCH; }
    attr_set.set();
  }
  hipLaunchKernelGGL(conv3d_lds_kernel, dim3(conv3d_lds_grid(gout)), dim3(256), lds_bytes, (hipStream_t)stream, a);
  AS_CHECK_LAUNCH("as_conv32_fwd(3-D LDS)");
  return AS_OK;
}


// =====================================================================================================
// Weight gradient of the same instance:  dW[tap][ci][co] = sum_p X[p + off(tap)][ci] * G[p][co]  over all positions.
// Same flattening: for one kd the nine (kh, kw) taps of a 128-position tile read ONE staged run of plane d+kd-1
// (plain copy, as in conv32_wgrad_lds_kernel: a ds_read_b32 of one voxel's 32 channels by 32 lanes is conflict-free)
// against the tile's G run.  G's halo is zero, so halo-column positions and whatever a clamped DMA group fetches
// beyond a plane's last voxel contribute nothing: no masks, no ragged-edge code.
// Workgroups come in triples (chunk c = blockIdx/3, kd = blockIdx%3): the three of a chunk walk the same tiles, each
// for its own kd with nine accumulators (144 registers) that live across all of its tiles, and write the taps
// 9*kd .. 9*kd+8 of slab c: the slabs have the [chunk][27][32][32] layout wgrad_reduce_kernel sums in fixed order.
// The direct-load kernel needed one 256-byte wave load per operand per MFMA (57-65 TFLOP/s).
struct Wgrad3dLdsArgs {
  const float* x;
  const float* gz;
  float* partial;      // [chunks][27][32][32]
  float* partial_db;   // [chunks][32]
  PclDev g;
  int tiles_per_plane, npos, ntiles, nchunks;
  int run, xgroups;    // staged X voxels (130 + 2*Wp, rounded up to 8) and its 8-voxel DMA groups
};

__global__ __launch_bounds__(256, 2) void conv3d_wgrad_lds_kernel(Wgrad3dLdsArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xbuf = smem;
  char* gbuf = smem + p.xgroups * 1024;
  const unsigned lds_x = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((lds3_ptr_t)xbuf));
  const unsigned lds_g = lds_x + (unsigned)(p.xgroups * 1024);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, li = lane & 31;
  const unsigned lane16 = (unsigned)lane * 16u;
  const int Wp = p.g.Wp;
  // XCD-aware (chunk, kd) assignment (block ids congruent mod 8 share an XCD and its 4-MB L2): an XCD owns a contiguous range
  // of chunks — at any time its workgroups walk ADJACENT tiles, whose runs overlap by a row above and below — and all three kd
  // of a chunk, which read the same G run and (one plane = tiles_per_plane chunks apart) the same X runs.  With blockIdx = 3 *
  // chunk + kd the three landed on three XCDs and every run was fetched from the fabric by each: 5.0x the algorithmic bytes
  // on the counters (profiles/r04_pmc_by_pairs.json).  Placement affects speed and traffic only.
  // Tiles: `full` rounds of tile = chunk + k * nchunks, and the remaining tiles one each to the chunks that are dispatched FIRST
  // on every XCD (they land on different CUs: no CU then hosts two workgroups with an extra tile — the launch is matrix-bound, a
  // CU's time is the sum of its two workgroups' tiles; with the extra tiles on chunks 0..rem-1 = one XCD's neighbours: 55 us for 50).
  int chunk, kd, extra_tile;
  const int full = p.ntiles / p.nchunks;
  if (!conv3d_wgrad_assign(blockIdx.x, p.ntiles, p.nchunks, &chunk, &kd, &extra_tile)) return;     // padding blocks (uniform per workgroup)
  const long plane_vox = (long)Wp * p.g.Hp;
  const int first = p.g.ph * Wp + p.g.pw;
  const int last_group = (int)plane_vox - 8;                       // last 8-voxel group that lies inside a plane
  const int zero_group = (p.g.Hp - 1) * Wp;                        // 8 voxels of the bottom pad row: zeros

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  // operand addresses of step 0 (tile-invariant): position 32*wave + h, tap (kh, kw) at run voxel + kh*Wp + kw
  const char* gaddr = gbuf + (32 * wave + h) * 128 + li * 4;
  const char* xaddr[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) xaddr[t] = xbuf + (32 * wave + h + (t / 3) * Wp + (t % 3)) * 128 + li * 4;

  for (int k = 0; k < full + (extra_tile >= 0 ? 1 : 0); ++k) {
    const int tile = k < full ? chunk + k * p.nchunks : extra_tile;
    const int plane = tile / p.tiles_per_plane, tt = tile - plane * p.tiles_per_plane;
    const int b = plane / p.g.D, d = plane - b * p.g.D;
    const int pos0 = first + 128 * tt;
    const float* xplane = p.x + ((long)b * p.g.Dp + d + p.g.pd + kd - 1) * plane_vox * 32;
    const float* gplane = p.gz + ((long)b * p.g.Dp + d + p.g.pd) * plane_vox * 32;
    const int xs = pos0 - Wp - 1;
    for (int i = wave; i < p.xgroups; i += 4)                       // clamped groups only ever multiply zero G
      dma3_1kb(xplane + (long)min(xs + 8 * i, last_group) * 32, lane16, lds_x + (unsigned)(i * 1024));
    const int g_end = first + p.npos;                               // first position past the last interior voxel
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = wave + 4 * k;
      const int gp = pos0 + 8 * i;
      dma3_1kb(gplane + (long)(gp < g_end ? gp : zero_group) * 32, lane16, lds_g + (unsigned)(i * 1024));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float bv[2], av[2][9];
    bv[0] = *reinterpret_cast<const float*>(gaddr);
#pragma unroll
    for (int t = 0; t < 9; ++t) av[0][t] = *reinterpret_cast<const float*>(xaddr[t]);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (s + 1 < 16) {
        bv[(s + 1) & 1] = *reinterpret_cast<const float*>(gaddr + (s + 1) * 256);
#pragma unroll
        for (int t = 0; t < 9; ++t) av[(s + 1) & 1][t] = *reinterpret_cast<const float*>(xaddr[t] + (s + 1) * 256);
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
  float* out = p.partial + ((long)chunk * 27 + 9 * kd) * 1024;
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
  if (kd == 1) {                                            // the bias gradient once per chunk
    float* dbs = reinterpret_cast<float*>(smem);
    bsum += __shfl_xor(bsum, 32, 64);
    if (h == 0) dbs[wave * 32 + li] = bsum;
    __syncthreads();
    if (threadIdx.x < 32) p.partial_db[chunk * 32 + li] = dbs[li] + dbs[32 + li] + dbs[64 + li] + dbs[96 + li];
  }
}

bool conv3d_wgrad_lds_applicable(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s) {
  if (!conv3d_lds_applicable(gin, gout, s)) return false;
  const int Wp = gin->W + 2 * gin->pw;
  return Wp >= 8 && (long)(conv3d_run(gin) + 7) / 8 * 1024 + 16384 <= 80 * 1024;     // two workgroups per CU
}

int conv3d_wgrad_lds_slabs(const as_pcl* gout) {
  const int ntiles = conv3d_lds_grid(gout);
  return ntiles < 168 ? ntiles : 168;                      // 21 chunks x 3 kd = 63 workgroups per XCD: one round of two per CU
                                                           // (22 per XCD = 66 workgroups on 64 slots: a second round, 82 us for 50)
}

int conv3d_wgrad_lds_launch(const float* x, const as_pcl* gin, const float* gz, const as_pcl* gout,
                            float* partial, float* partial_db, void* stream) {
  Wgrad3dLdsArgs a;
  a.x = x; a.gz = gz; a.partial = partial; a.partial_db = partial_db;
  a.g = as_make_dev(gin);
  a.tiles_per_plane = conv3d_tiles_per_plane(gin);
  a.npos = (gin->H - 1) * a.g.Wp + gin->W;
  a.ntiles = conv3d_lds_grid(gout);
  a.nchunks = conv3d_wgrad_lds_slabs(gout);
  a.run = conv3d_run(gin);
  a.xgroups = (a.run + 7) / 8;
  const int lds_bytes = a.xgroups * 1024 + 16384;
  static AsPerDevice attr_set;
  if (!attr_set.get()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_wgrad_lds_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (e != hipSuccess) { as_set_error("as_conv32_wgrad(3-D LDS): %s", hipGetErrorString(e)); return AS_ERR_LAUNCH; }
    attr_set.set();
  }
  hipLaunchKernelGGL(conv3d_wgrad_lds_kernel, dim3(8 * ((a.nchunks + 7) / 8) * 3), dim3(256), lds_bytes, (hipStream_t)stream, a);
  AS_CHECK_LAUNCH("as_conv32_wgrad(3-D LDS)");
  return AS_OK;
}
