// Weight / bias gradient of a full-resolution refinement layer (3x3, dilation 1/2/4/8, 32->32: stereo_net.py:10-18, 33-51,
// 97) by the minimal-filtering algorithm F(3x3, 2x2) — the second half of the layer's backward pass (conv32_wino.hip MODE 2 is
// the first and leaves g_z behind):
//   dW[a][b] = sum over 2x2 tiles of g_z of  A^T [ (G g G^T) .* (B^T x B) ] A ,   g = the g_z tile, x = the 4x4 input tile
//   A^T = [1 1 1 0; 0 1 -1 0; 0 1 1 1],  G = [1 0; 1/2 1/2; 1/2 -1/2; 0 1],  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 -1 0 1]
// 16 products per tile (4 output pixels) and channel pair instead of 36; the SUM OVER TILES is the K dimension of the matrix
// instruction:  M[r][c][ci][co] += sum_tiles (B^T x B)[r][c][tile][ci] * (G g G^T)[r][c][tile][co], so the lane is the channel
// (ci for the A operand, co for B) and one v_mfma_f32_32x32x2_f32 consumes two tiles.  The two factors 1/2 of G are applied once,
// to M, at the end.
//
// Nothing here is element-wise work on a tile's way in, nothing leaves per tile: rows of x (with halo) and of g_z arrive by
// LDS-DMA one tile ahead, the matrix waves read them with ds_read_b32 (32 lanes = one voxel's 32 channels; the two half-waves
// read two tiles whose voxels lie in different bank halves: rows are stored in 8-voxel chunks one voxel apart), 12 vector
// operations per 4 MFMAs form both transforms.  So ONE workgroup per CU (4 waves, the 16 accumulators split by the transformed
// row r as in conv32_wino.hip) keeps the matrix pipe fed with one barrier per tile; 147 KB of LDS: a ring of 8 x rows and three
// buffers of a g_z row pair — the rows of TWO tiles are in flight (with one, 9 MB in flight chip-wide against ~3 us of loaded
// HBM latency capped the launch at ~3 TB/s: 190-210 us where the matrix work is 111).  At the end the waves exchange (M A) through LDS, apply A^T and write the workgroup's slab
// [9][32][32] (+ bias partial) for the batch reduce that every weight gradient of the step shares.
//
// Diagnostic builds (tests/tools/wino_exp.sh rebuilds with EXTRA=-D...; results are then WRONG, only the time means something):
// WW_EXP_NODMA (no row traffic in the tile loop), WW_EXP_NOLOAD (no operand reads), WW_EXP_NOVALU (operands fed to the MFMAs as
// they come), WW_EXP_NOMFMA (data movement only) — they gave the floors quoted in DESIGN 4: matrix work alone 121-127 us, + loads
// +15, + transforms +15, both +55, data movement alone 100 us.
#include "as_common.h"
#include "conv32_wino.h"

#define WW_CHUNK 1152                        // 8 voxels of 128 B + one voxel of padding
#define WW_XROW (10 * WW_CHUNK)              // 11,520: 80 staged voxels (8 + 64 + 8)
#define WW_XSLOTS 8
#define WW_GROW (8 * WW_CHUNK)               // 9,216: the segment's own 64 voxels
#define WW_G_OFF (WW_XSLOTS * WW_XROW)       // 92,160
#define WW_GBUF (2 * WW_GROW)                // 18,432: a pair of g_z rows
#define WW_LDS_BYTES (WW_G_OFF + 3 * WW_GBUF)    // 147,456
#ifndef WW_GRID
#define WW_GRID 256
#endif

struct WgradWinoArgs {
  const float* x;          // layer input (PCL, zero halo)
  const float* gz;         // gradient w.r.t. the layer's pre-activation (PCL, zero halo)
  float* partial;          // [WW_GRID][9][32][32]
  float* partial_db;       // [WW_GRID][32]
  PclDev g;
  int nseg, pairs;
};

typedef __attribute__((address_space(3))) void* ww_lds_t;
typedef float ww_f32x2 __attribute__((ext_vector_type(2)));

__device__ inline void ww_dma_1kb(const float* sbase, unsigned voff, unsigned m0) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               :: "s"(m0), "v"(voff), "s"(sbase) : "memory", "m0");
}

template <int L> __host__ __device__ constexpr int ww_c0(int t) { return ((t >> L) << (L + 1)) | (t & ((1 << L) - 1)); }
__host__ __device__ constexpr int ww_off(int v) { return (v >> 3) * WW_CHUNK + (v & 7) * 128; }
// tile of matrix step s, half 0 (half 1: + 1 for d > 1, + 4 for d = 1 — a voxel in the other bank half either way)
template <int L> __host__ __device__ constexpr int ww_tile(int s) { return L == 0 ? (s & 3) + 8 * (s >> 2) : 2 * s; }

// RW: the wave's row of the transformed tiles; PART: which half of a tile's 16 matrix steps it takes (two waves per SIMD: one
// wave's operand reads and transforms under the other's MFMAs — with one wave per SIMD the launch took 195-210 us where the
// matrix work alone is 120-127)
template <int RW, int PART, int L>
__device__ __forceinline__ void conv32_wino_wgrad_role(const WgradWinoArgs& p, char* smem) {
  constexpr int d = 1 << L;
  constexpr int W8 = PART * 4 + RW;                         // wave index: DMA jobs i = W8, W8 + 8, ..
  constexpr int NJOB = W8 < 4 ? 5 : 4;                      // of a tile's 36 DMA jobs
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((ww_lds_t)smem));
  const int lane = threadIdx.x & 63;
  const int h = lane >> 5, li = lane & 31;
  const unsigned lane16 = (unsigned)lane * 16u;
  const int H = p.g.H, W = p.g.W, Wp = p.g.Wp;
  // this wave's row of the transformed tiles: R = x[ra] + sg * x[rb]  (B^T rows: x0-x2, x1+x2, x2-x1, x3-x1)
  constexpr int ra = RW == 0 ? 0 : (RW == 1 ? 1 : (RW == 2 ? 2 : 3));
  constexpr int rb = RW == 0 ? 2 : (RW == 1 ? 2 : 1);
  [[maybe_unused]] constexpr float sg = RW == 1 ? 1.f : -1.f;
  const int lane_off = li * 4 + h * (L == 0 ? WW_CHUNK : 128);

  f32x16 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float bsum = 0.f;

  const long t_total = (long)p.g.B * p.nseg * p.pairs;
  long t_next = t_total * blockIdx.x / gridDim.x;
  const long t_end = t_total * (blockIdx.x + 1) / gridDim.x;
  while (t_next < t_end) {
    const int blk = (int)(t_next / p.pairs);
    int pj0 = (int)(t_next - (long)blk * p.pairs);
    int r0 = 0, nrow = (H + d - 1) / d;                   // rows of comb r0
    while (pj0 >= (nrow + 1) / 2) { pj0 -= (nrow + 1) / 2; ++r0; nrow = (H - r0 + d - 1) / d; }
    const int pj1 = (int)min((long)((nrow + 1) / 2), pj0 + (t_end - t_next));
    t_next += pj1 - pj0;
    const int j0 = 2 * pj0, j1 = min(2 * pj1, nrow);
    const int seg = blk % p.nseg;
    const int b = blk / p.nseg;
    // The last segment is NOT shifted back to end at the image edge (conv32_wino.hip does that and masks the duplicated
    // columns): its chunks that start beyond the image come from the zero halo row, the one that straddles the edge ends inside
    // the halo (pw >= 8) — so every g_z pixel is met exactly once and no lane-level mask is needed in the matrix loop.
    const int x0 = 64 * seg;
    const long img = (long)b * p.g.Hp;

    // DMA job i of a tile whose first comb row is jt: i < 20: chunk i % 10 of x row jt + 1 + i / 10 (the two NEW rows of the
    // tile: its input rows are jt-1 .. jt+2); else chunk (i-20) % 8 of g_z row jt + (i-20) / 8.  Rows outside the image (or, for
    // g_z, the piece) and chunks beyond the image's right edge come from the zero halo row above the image.
    auto issue = [&](int jt, int i) {
      if (i < 20) {
        const int jj = jt + 1 + i / 10, c = i % 10;
        const int yy = r0 + jj * d;
        const int xs = x0 - 8 + 8 * c;                      // first column of the chunk
        const int y = (yy >= 0 && yy < H && xs < W + 8) ? yy : -1;
        const float* src = p.x + ((img + y + p.g.ph) * Wp + (y < 0 ? 0 : xs) + p.g.pw) * 32;
        ww_dma_1kb(src, lane16, lds0 + (unsigned)(((jj + 1) & (WW_XSLOTS - 1)) * WW_XROW + c * WW_CHUNK));
      } else {
        const int i2 = i - 20, jj = jt + i2 / 8, c = i2 % 8;
        const int yy = r0 + jj * d;
        const int xs = x0 + 8 * c;
        const int y = (jj < j1 && yy < H && xs < W) ? yy : -1;
        const float* src = p.gz + ((img + y + p.g.ph) * Wp + (y < 0 ? 0 : xs) + p.g.pw) * 32;
        ww_dma_1kb(src, lane16, lds0 + (unsigned)(WW_G_OFF + ((jt >> 1) % 3) * WW_GBUF + (i2 / 8) * WW_GROW + c * WW_CHUNK));
      }
    };
    // ---- run-in: x rows j0-1 .. j0+2 and the g_z pair of tile j0 ----
    // (x rows of issue(jt, .) are jt+1, jt+2: jt = j0-2 brings j0-1, j0; jt = j0 brings j0+1, j0+2)
    for (int i = W8; i < 20; i += 8) issue(j0 - 2, i);
    for (int i = W8; i < 36; i += 8) issue(j0, i);
    if (j0 + 2 < j1) {                                     // the second tile's rows stay in flight (NJOB jobs per wave)
      for (int i = W8; i < 36; i += 8) issue(j0 + 2, i);
      if constexpr (NJOB == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    for (int j = j0; j < j1; j += 2) {
      // the rows of the tile after next: x rows j+5, j+6 into the two slots nobody reads or awaits, g_z pair into the third buffer
      const bool ahead = j + 4 < j1;
#ifndef WW_EXP_NODMA
      if (ahead) for (int i = W8; i < 36; i += 8) issue(j + 4, i);
#endif
      // input row m of the tile = comb row j-1+m = slot (j+m) & 7
      const char* xa_row = smem + ((j + ra) & (WW_XSLOTS - 1)) * WW_XROW + lane_off;
      const char* xb_row = smem + ((j + rb) & (WW_XSLOTS - 1)) * WW_XROW + lane_off;
      const char* g_row = smem + WW_G_OFF + ((j >> 1) % 3) * WW_GBUF + lane_off;
      float xa[2][4], xb[2][4], g0[2][2], g1[2][2];
      auto load_step = [&](int s, float (&xa)[4], float (&xb)[4], float (&g0)[2], float (&g1)[2]) {
        const int t0 = ww_tile<L>(s);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int off = ww_off(8 + ww_c0<L>(t0) + (m - 1) * d);
          xa[m] = *reinterpret_cast<const float*>(xa_row + off);
          xb[m] = *reinterpret_cast<const float*>(xb_row + off);
        }
#pragma unroll
        for (int jc = 0; jc < 2; ++jc) {
          const int off = ww_off(ww_c0<L>(t0) + jc * d);
          if (RW != 3) g0[jc] = *reinterpret_cast<const float*>(g_row + off);
          if (RW != 0) g1[jc] = *reinterpret_cast<const float*>(g_row + WW_GROW + off);
        }
      };
#ifndef WW_EXP_NOMFMA
      load_step(8 * PART, xa[0], xb[0], g0[0], g1[0]);
#pragma unroll
      for (int s = 8 * PART; s < 8 * PART + 8; ++s) {
#ifndef WW_EXP_NOLOAD
        if (s + 1 < 8 * PART + 8) load_step(s + 1, xa[(s + 1) & 1], xb[(s + 1) & 1], g0[(s + 1) & 1], g1[(s + 1) & 1]);
#endif
        __builtin_amdgcn_sched_barrier(0);
#ifdef WW_EXP_NOVALU
        {
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[s & 1][c], xb[s & 1][c], acc[c], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          continue;
        }
#endif
#ifdef WW_SCALAR_TRANSFORMS
        float Gr[2];
#pragma unroll
        for (int jc = 0; jc < 2; ++jc) {
          const float u0 = RW != 3 ? g0[s & 1][jc] : 0.f, u1 = RW != 0 ? g1[s & 1][jc] : 0.f;
          if (RW == 1) bsum += u0 + u1;
          Gr[jc] = RW == 0 ? u0 : (RW == 3 ? u1 : (RW == 1 ? u0 + u1 : u0 - u1));   // (the factor 1/2 of rows 1, 2: at the end)
        }
        const float Gt[4] = {Gr[0], Gr[0] + Gr[1], Gr[0] - Gr[1], Gr[1]};          // (likewise for columns 1, 2)
        float Rt[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) Rt[m] = xa[s & 1][m] + sg * xb[s & 1][m];
        const float V[4] = {Rt[0] - Rt[2], Rt[1] + Rt[2], Rt[2] - Rt[1], Rt[3] - Rt[1]};
#else
        // The same signed sums on PACKED instructions (round 4): two values per v_pk_add_f32 with per-half negation and half
        // selection (a - b = a + (-b): the same IEEE operation, bit-identical sums) — 6 vector instructions per 4 MFMAs instead
        // of 12.  Inline asm, because hipcc turns every vector difference into one v_sub_f32 per element and splits packed
        // instructions that it does emit next to MFMAs back into two; asm is NOT padded for the matrix pipe's hazards: two wait
        // states behind the last of them (wn_before_mfma's rule: conv32_wino_dev.h; tests/tools/check_async_loads.py checks it).
        const ww_f32x2 u0 = {g0[s & 1][0], g0[s & 1][1]}, u1 = {g1[s & 1][0], g1[s & 1][1]};
        ww_f32x2 Gr, Gm, Rt01, Rt23, V01, V23;
        if constexpr (RW == 0) Gr = u0;
        else if constexpr (RW == 3) Gr = u1;
        else if constexpr (RW == 1) { asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(Gr) : "v"(u0), "v"(u1)); bsum += Gr.x; bsum += Gr.y; }
        else asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(Gr) : "v"(u0), "v"(u1));
        // (Gr0 + Gr1, Gr0 - Gr1): src0 = Gr.lo twice, src1 = Gr.hi twice, the high half negated
        asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(Gm) : "v"(Gr));
        const ww_f32x2 xa01 = {xa[s & 1][0], xa[s & 1][1]}, xa23 = {xa[s & 1][2], xa[s & 1][3]};
        const ww_f32x2 xb01 = {xb[s & 1][0], xb[s & 1][1]}, xb23 = {xb[s & 1][2], xb[s & 1][3]};
        if constexpr (RW == 1) {
          asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(Rt01) : "v"(xa01), "v"(xb01));
          asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(Rt23) : "v"(xa23), "v"(xb23));
        } else {
          asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(Rt01) : "v"(xa01), "v"(xb01));
          asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(Rt23) : "v"(xa23), "v"(xb23));
        }
        // (Rt0 - Rt2, Rt1 + Rt2): src1 = Rt2 twice, negated in the low half;  (Rt2 - Rt1, Rt3 - Rt1): src1 = Rt1 twice, negated
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(V01) : "v"(Rt01), "v"(Rt23));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(V23) : "v"(Rt23), "v"(Rt01));
        asm volatile("s_nop 1" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);                 // (hipcc otherwise moves an MFMA in between the asm statements: one wait state)
        const float Gt[4] = {Gr.x, Gm.x, Gm.y, Gr.y};
        const float V[4] = {V01.x, V01.y, V23.x, V23.y};
#endif
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[c], Gt[c], acc[c], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
      // the NEXT tile's rows are home when only the jobs issued at the top of this tile are outstanding (in-order retirement)
      if (ahead) {
        if constexpr (NJOB == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();                                     // the next tile's rows are in place; this tile's are free
    }
  }

  // ---- the two halves of the matrix steps: waves 4-7 hand their accumulators (and bias sums) to waves 0-3 ----
  float* ex = reinterpret_cast<float*>(smem);             // (the x ring is done with) [4 r][4 c][16][64] floats = 65,536 B
  if constexpr (PART == 1) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) ex[((RW * 4 + c) * 16 + r) * 64 + lane] = acc[c][r];
    if (RW == 1) ex[16384 + lane] = bsum;
  }
  __syncthreads();
  if constexpr (PART == 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] += ex[((RW * 4 + c) * 16 + r) * 64 + lane];
    if (RW == 1) bsum += ex[16384 + lane];
  }
  __syncthreads();
  // ---- (M A) in the wave, A^T across the waves, slab ----
  constexpr float sr = (RW == 1 || RW == 2) ? 0.5f : 1.f;
  if constexpr (PART == 0) {                               // [4 r][3 b][16][64] floats = 49,152 B
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float m0 = sr * acc[0][r], m1 = (0.5f * sr) * acc[1][r], m2 = (0.5f * sr) * acc[2][r], m3 = sr * acc[3][r];
    ex[((RW * 3 + 0) * 16 + r) * 64 + lane] = (m0 + m1) + m2;
    ex[((RW * 3 + 1) * 16 + r) * 64 + lane] = m1 - m2;
    ex[((RW * 3 + 2) * 16 + r) * 64 + lane] = (m1 + m2) + m3;
  }
  }
  __syncthreads();
  float* out = p.partial + (long)blockIdx.x * 9 * 1024;
  for (int o = W8; o < 9; o += 8) {
    const int a = o / 3, bb = o - 3 * a;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float t0 = ex[((0 * 3 + bb) * 16 + r) * 64 + lane], t1 = ex[((1 * 3 + bb) * 16 + r) * 64 + lane];
      const float t2 = ex[((2 * 3 + bb) * 16 + r) * 64 + lane], t3 = ex[((3 * 3 + bb) * 16 + r) * 64 + lane];
      const float v = a == 0 ? (t0 + t1) + t2 : (a == 1 ? t1 - t2 : (t1 + t2) + t3);
      const int ci = (r & 3) + 8 * (r >> 2) + 4 * h;
      out[o * 1024 + ci * 32 + li] = v;
    }
  }
  if (RW == 1 && PART == 0) {
    bsum += __shfl_xor(bsum, 32, 64);
    if (h == 0) p.partial_db[blockIdx.x * 32 + li] = bsum;
  }
}

template <int L>
__global__ __launch_bounds__(512, 1) void conv32_wino_wgrad_kernel(WgradWinoArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  switch (wave) {
    case 0: conv32_wino_wgrad_role<0, 0, L>(p, smem_dyn); break;
    case 1: conv32_wino_wgrad_role<1, 0, L>(p, smem_dyn); break;
    case 2: conv32_wino_wgrad_role<2, 0, L>(p, smem_dyn); break;
    case 3: conv32_wino_wgrad_role<3, 0, L>(p, smem_dyn); break;
    case 4: conv32_wino_wgrad_role<0, 1, L>(p, smem_dyn); break;
    case 5: conv32_wino_wgrad_role<1, 1, L>(p, smem_dyn); break;
    case 6: conv32_wino_wgrad_role<2, 1, L>(p, smem_dyn); break;
    default: conv32_wino_wgrad_role<3, 1, L>(p, smem_dyn); break;
  }
}

int conv32_wino_wgrad_slabs(void) { return WW_GRID; }

int conv32_wino_wgrad_launch(const float* x, const float* g_z, const as_pcl* g, const as_conv_shape* s, float* partial,
                             float* partial_db, void* stream) {
  static AsPerDevice attr_set[4];
  const int L = s->dil == 1 ? 0 : (s->dil == 2 ? 1 : (s->dil == 4 ? 2 : 3));
  const void* fn = L == 0 ? reinterpret_cast<const void*>(conv32_wino_wgrad_kernel<0>)
                 : L == 1 ? reinterpret_cast<const void*>(conv32_wino_wgrad_kernel<1>)
                 : L == 2 ? reinterpret_cast<const void*>(conv32_wino_wgrad_kernel<2>)
                          : reinterpret_cast<const void*>(conv32_wino_wgrad_kernel<3>);
  if (!attr_set[L].get()) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, WW_LDS_BYTES);
    if (e != hipSuccess) { as_set_error("as_conv32_wino_bwd: %s", hipGetErrorString(e)); return AS_ERR_LAUNCH; }
    attr_set[L].set();
  }
  WgradWinoArgs a;
  a.x = x; a.gz = g_z; a.partial = partial; a.partial_db = partial_db;
  a.g = as_make_dev(g);
  a.nseg = (g->W + 63) / 64;
  long pairs = 0;
  for (int r = 0; r < s->dil; ++r) pairs += ((g->H - r + s->dil - 1) / s->dil + 1) / 2;
  a.pairs = (int)pairs;
  void* kargs[] = {&a};
  hipError_t le = hipLaunchKernel(fn, dim3(WW_GRID), dim3(512), kargs, WW_LDS_BYTES, (hipStream_t)stream);
  if (le != hipSuccess) { as_set_error("as_conv32_wino_bwd: launch failed: %s", hipGetErrorString(le)); return AS_ERR_LAUNCH; }
  return AS_OK;
}
