// Whole backward of a full-resolution refinement layer (3x3, dilation 1/2/4/8, stride 1, 32->32: stereo_net.py:10-18, 33-51, 97)
// by minimal filtering in ONE launch: the data gradient F(2x2, 3x3) of conv32_wino_dgrad.hip and the weight gradient F(3x3, 2x2)
// of conv32_wino_wgrad.hip side by side in one 8-wave workgroup per CU, so that g_z — stage 3 of the layer's BatchNorm backward
// applied to (g_a, z) on the way in — never leaves the chip: the two-launch form writes it (238 MB at 4 pairs) and reads it back.
//   reads  x, g_a, z, z_next     writes  g_x, the weight / bias gradient slabs, the next BatchNorm's sums     (5 tensor passes for 7)
//
//   waves 0-3   the data gradient's four roles, exactly as in conv32_wino_dgrad_role.h (FUSED): matrix phase on the g_z ring, waves
//               1, 2 convert rows j+3, j+4 (g_a, z -> g_z in LDS, raw g_a rows for the skip connection), waves 0, 3 finish the
//               output rows (skip, store, next BatchNorm's sums)
//   waves 4-7   the weight gradient: wave 4 + r holds the sixteen-register accumulators M[r][0..3] for the WHOLE launch and per tile
//               adds 16 matrix steps (two tiles of the 32 per step: lane = channel, K = tile), operands read with ds_read_b32 from
//               the x ring (rows j-1 .. j+2, brought by LDS-DMA one tile ahead, 6 slots) and from the data gradient's g_z ring
//               (rows j, j+1, swizzled for the other waves' ds_read_b128: a voxel's 32 channels are still ONE 128-byte line, read
//               through eight per-lane bases, one per swizzle key)
//
// Schedule.  On gfx950 fp32 MFMAs and vector instructions of the two waves of a SIMD do not overlap — their times add
// (tests/tools/scratch/mfma_sustained.hip: 64 cycles per MFMA + ~5 per packed vector instruction, whichever wave issues them) —
// so a tile costs the SIMD 128 MFMAs + all vector work; what the pairing buys is that nobody WAITS: the weight-gradient waves
// run tile j's matrix steps between the data gradient's barriers B1 and B2 (its conversion / epilogue phase: HBM round trips,
// LDS traffic), and sit at the barrier during its matrix phase.
//   P1 (B2 of tile j-2 .. B1): data-gradient matrix phase on rows j-1 .. j+2;  weight-gradient waves: the first FB_K1 matrix steps
//   P2 (B1 .. B2):            conversion of rows j+3, j+4 + output rows;       the other weight-gradient matrix steps (rows j, j+1 / x rows)
// Rows j, j+1 of g_z must therefore outlive B1 while rows j+3, j+4 are written: a ring of FIVE rows (row r in slot r mod 5).
//
// Shared columns.  The last 64-pixel segment of a row is shifted back to end at the image edge (conv32_wino.hip); its neighbour
// keeps only its first W mod 64 columns.  The weight gradient must count every g_z pixel once, so in that neighbour segment the
// g_z values of columns it does not own are read as zeros (a second instantiation of the matrix steps, chosen per piece).
//
// LDS: g_z ring 5 x (64 + 2d) x 128 | coefficients 768 | T exchange 16 KB | raw g_a ring 24 KB | x ring 6 x 72 (80 for d = 8) x 128
//      = 139,264 / 140,544 / 143,104 / 154,368 bytes for d = 1 / 2 / 4 / 8 — one workgroup per CU, every dilation.
// g_x and the next BatchNorm's partial sums are those of the two-launch form bit for bit per element / equal to rounding per
// partial (256 workgroups here, 512 there); dW differs from conv32_wino_wgrad.hip's by the order of the sum over tiles only.
#include "as_common.h"
#include "conv32_wino.h"
#include "conv32_wino_dev.h"
#include "conv32_wino_dgrad_role.h"
#ifdef FB_TIMING_BUILD
#include <cstdio>
#include <cstdlib>
#endif

#ifndef FB_GRID
#define FB_GRID 256
#endif
#ifndef FB_K1
#define FB_K1 6                       // weight-gradient steps of a tile (of 16) that run beside the data gradient's matrix phase: 0 .. 16 swept,
                                      // 4-6 best by 1-3 % per launch, 0.4 % of a step (profiles/r05_s_ab_*: the pipe is the same pipe)
#endif

template <int L> struct FbGeo {
  using D = DgGeo<L, true>;
  static constexpr int d = 1 << L;
  static constexpr int NX = ((D::NV + 7) / 8) * 8;       // staged x voxels per row: whole 1-KB DMA pieces (72, 72, 72, 80)
  static constexpr int XROWB = NX * 128;
  static constexpr int XSLOTS = 6;
  static constexpr int XR_OFF = D::LDS;                  // x ring behind everything the data gradient owns
  static constexpr int LDS = XR_OFF + XSLOTS * XROWB;
  static constexpr int EX_OFF = 4096;                    // (after the tile loops) [4 r][3 b][16][64] floats, behind the sums' scratch
  __device__ static inline int xslot(int row) { return (int)((unsigned)(row + 6) % 6u); }
};

struct FusedBwdArgs {
  DgradArgs dg;            // (g_z unused)
  const float* x;          // layer input (PCL, zero halo)
  float* partial;          // [FB_GRID][9][32][32]
  float* partial_db;       // [FB_GRID][32]
};

typedef __attribute__((address_space(3))) void* fb_lds_t;
typedef float fb_f32x2 __attribute__((ext_vector_type(2)));

__device__ inline void fb_dma_1kb(const float* sbase, unsigned voff, unsigned m0) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0), "v"(voff), "s"(sbase) : "memory", "m0");
}

// staged voxel (80-voxel frame of wn_addr) -> LDS position, and swizzle key, as compile-time constants
template <int L> __host__ __device__ constexpr int fb_pos(int v) { return L == 0 ? ((v & ~3) | ((v & 1) << 1) | ((v >> 1) & 1)) : v; }
template <int L> __host__ __device__ constexpr int fb_swz(int v) {
  const int w = v + 8;
  const int key = ((w >> (L + 1)) << L) | (w & ((1 << L) - 1));
  return (key >> 1) & 7;
}

// The sixteen matrix steps of one tile (rows j, j+1 of g_z; rows j-1 .. j+2 of x) for the wave that owns row RW of the transformed
// tiles.  Step s takes tiles s (lanes 0-31) and s + 16 (lanes 32-63: 32 voxels further in both rings, the same swizzle key).
// MASKED: g_z values of columns >= keep are not this segment's (kh = keep - 32 * half).
template <int RW, int L, bool MASKED, int S0, int S1>
__device__ __forceinline__ void fb_w_steps(f32x16 (&acc)[4], float& bsum, const char* smem, int xa_base, int xb_base,
                                           const int (&g0_base)[8], const int (&g1_base)[8], int kh) {
  if constexpr (S0 >= S1) return;
  using G = DgGeo<L, true>;
  constexpr int d = 1 << L;
  float xa[2][4], xb[2][4], g0[2][2], g1[2][2];
  auto load_step = [&](int s, float (&xa)[4], float (&xb)[4], float (&g0)[2], float (&g1)[2]) {
    const int c0 = wn_c0<L>(s);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      xa[m] = *reinterpret_cast<const float*>(smem + xa_base + (c0 + m * d) * 128);
      xb[m] = *reinterpret_cast<const float*>(smem + xb_base + (c0 + m * d) * 128);
    }
#pragma unroll
    for (int jc = 0; jc < 2; ++jc) {
      const int v = 8 + c0 + jc * d;
      const int S = fb_swz<L>(v), off = (fb_pos<L>(v) - G::V0) * 128;
      if (RW != 3) g0[jc] = *reinterpret_cast<const float*>(smem + g0_base[S] + off);
      if (RW != 0) g1[jc] = *reinterpret_cast<const float*>(smem + g1_base[S] + off);
      if constexpr (MASKED) {
        const bool own = c0 + jc * d < kh;
        if (RW != 3) g0[jc] = own ? g0[jc] : 0.f;
        if (RW != 0) g1[jc] = own ? g1[jc] : 0.f;
      }
    }
  };
  load_step(S0, xa[S0 & 1], xb[S0 & 1], g0[S0 & 1], g1[S0 & 1]);
#pragma unroll
  for (int s = S0; s < S1; ++s) {
    if (s + 1 < S1) load_step(s + 1, xa[(s + 1) & 1], xb[(s + 1) & 1], g0[(s + 1) & 1], g1[(s + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
    // both transforms on packed instructions (conv32_wino_wgrad.hip: the same sequences, the same hazard padding)
    const fb_f32x2 u0 = {g0[s & 1][0], g0[s & 1][1]}, u1 = {g1[s & 1][0], g1[s & 1][1]};
    fb_f32x2 Gr, Gm, Rt01, Rt23, V01, V23;
    if constexpr (RW == 0) Gr = u0;
    else if constexpr (RW == 3) Gr = u1;
    else if constexpr (RW == 1) { asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(Gr) : "v"(u0), "v"(u1)); bsum += Gr.x; bsum += Gr.y; }
    else asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(Gr) : "v"(u0), "v"(u1));
    asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(Gm) : "v"(Gr));
    const fb_f32x2 xa01 = {xa[s & 1][0], xa[s & 1][1]}, xa23 = {xa[s & 1][2], xa[s & 1][3]};
    const fb_f32x2 xb01 = {xb[s & 1][0], xb[s & 1][1]}, xb23 = {xb[s & 1][2], xb[s & 1][3]};
    if constexpr (RW == 1) {
      asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(Rt01) : "v"(xa01), "v"(xb01));
      asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(Rt23) : "v"(xa23), "v"(xb23));
    } else {
      asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(Rt01) : "v"(xa01), "v"(xb01));
      asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(Rt23) : "v"(xa23), "v"(xb23));
    }
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(V01) : "v"(Rt01), "v"(Rt23));
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(V23) : "v"(Rt23), "v"(Rt01));
    asm volatile("s_nop 1" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const float Gt[4] = {Gr.x, Gm.x, Gm.y, Gr.y};
    const float V[4] = {V01.x, V01.y, V23.x, V23.y};
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[c], Gt[c], acc[c], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int RW, int L>
__device__ __forceinline__ void fused_w_role(const FusedBwdArgs& q, char* smem) {
  using G = DgGeo<L, true>;
  using F = FbGeo<L>;
  constexpr int d = 1 << L;
  constexpr int NPC = F::NX / 8;                            // 1-KB DMA pieces per x row
  const DgradArgs& p = q.dg;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((fb_lds_t)smem));
  const int lane = threadIdx.x & 63;
  const int h = lane >> 5, li = lane & 31;
  const unsigned lane16 = (unsigned)lane * 16u;
  const int H = p.g.H, W = p.g.W, Wp = p.g.Wp;
  // this wave's row of the transformed tiles: R = x[ra] + sg * x[rb]  (B^T rows: x0-x2, x1+x2, x2-x1, x3-x1)
  constexpr int ra = RW;
  constexpr int rb = RW == 0 ? 2 : (RW == 1 ? 2 : 1);
  const int x_lane = F::XR_OFF + li * 4 + h * 4096;
  int g_lane[8];
#pragma unroll
  for (int S = 0; S < 8; ++S) g_lane[S] = ((((li >> 2) ^ S) << 4) | ((li & 3) << 2)) + h * 4096;

  f32x16 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float bsum = 0.f;
#ifdef FB_PRIO_W                                           // (experiment: static issue priority of the weight-gradient waves)
  __builtin_amdgcn_s_setprio(FB_PRIO_W);
#endif
  constexpr bool FUSED = true;                             // (for FB_T)
#ifdef FB_TIMING_BUILD
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = clock64();
  const long long wall0 = wall_clock64();
#endif

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
    const int x0 = min(64 * seg, W - 64);
    const int keep = (seg == p.nseg - 2 && 64 * p.nseg > W) ? W - 64 * (p.nseg - 1) : 64;
    const int kh = keep - 32 * h;
    const long img = (long)b * p.g.Hp;

    // x row jj of the comb (rows outside the image: the zero halo row above it), piece c: voxels x0 - d + 8c .. + 7
    auto issue_row = [&](int jj, int c) {
      const int yy = r0 + jj * d;
      const int y = (yy >= 0 && yy < H) ? yy : -1;
      const float* src = q.x + ((img + y + p.g.ph) * Wp + x0 - d + 8 * c + p.g.pw) * 32;
      fb_dma_1kb(src, lane16, lds0 + (unsigned)(F::XR_OFF + F::xslot(jj) * F::XROWB + c * 1024));
    };
    // ---- run-in: x rows j0-1 .. j0+2 ----
    for (int i = RW; i < 4 * NPC; i += 4) issue_row(j0 - 1 + i / NPC, i % NPC);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                       // (= the data gradient's barrier behind its run-in)
    FB_T(0);

    // Everything tile j's matrix steps read is in LDS when the barrier B2 of tile j-2 opens (g_z rows j, j+1 were converted
    // during tiles j-4 and j-2; the x rows are home: see the wait below) and nothing overwrites it before B1 of tile j+2, so
    // the first FB_K1 of the sixteen steps run BESIDE the data gradient's matrix phase (between B2 and B1): the SIMD then has
    // two waves with matrix work, and one's LDS / issue stalls are filled by the other's MFMAs.  FB_K1 = 0 is the schedule
    // of the header (every step between B1 and B2).
    for (int j = j0; j < j1; j += 2) {
      if (j + 2 < j1)                                      // rows j+3, j+4 into the two slots nobody reads (last read: tile j-2's steps)
        for (int i = RW; i < 2 * NPC; i += 4) issue_row(j + 3 + i / NPC, i % NPC);
      const int xa_base = x_lane + F::xslot(j - 1 + ra) * F::XROWB;
      const int xb_base = x_lane + F::xslot(j - 1 + rb) * F::XROWB;
      int g0_base[8], g1_base[8];
      const int s0 = G::slot(j) * G::ROWB, s1 = G::slot(j + 1) * G::ROWB;
#pragma unroll
      for (int S = 0; S < 8; ++S) { g0_base[S] = g_lane[S] + s0; g1_base[S] = g_lane[S] + s1; }
      if (keep < 64) fb_w_steps<RW, L, true, 0, FB_K1>(acc, bsum, smem, xa_base, xb_base, g0_base, g1_base, kh);
      else fb_w_steps<RW, L, false, 0, FB_K1>(acc, bsum, smem, xa_base, xb_base, g0_base, g1_base, kh);
      FB_T(1);
      __syncthreads();                                     // B1
      FB_T(3);
      if (keep < 64) fb_w_steps<RW, L, true, FB_K1, 16>(acc, bsum, smem, xa_base, xb_base, g0_base, g1_base, kh);
      else fb_w_steps<RW, L, false, FB_K1, 16>(acc, bsum, smem, xa_base, xb_base, g0_base, g1_base, kh);
      FB_T(4);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the rows requested at the top of this tile are home before B2: every wave may read them after it
      FB_T(6);
      __syncthreads();                                     // B2
      FB_T(5);
    }
  }
#ifdef FB_TIMING_BUILD
  FB_T(7);
  if (p.timing && lane == 0) {
    long long* o = p.timing + ((long)blockIdx.x * 8 + 4 + RW) * 10;
    for (int i = 0; i < 8; ++i) o[i] = tacc[i];
    o[8] = wall_clock64() - wall0;
    o[9] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
  }
#endif

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // ---- (M A) in the wave, A^T across the waves, slab (conv32_wino_wgrad.hip's epilogue) ----
  float* ex = reinterpret_cast<float*>(smem + F::EX_OFF);
  constexpr float sr = (RW == 1 || RW == 2) ? 0.5f : 1.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float m0 = sr * acc[0][r], m1 = (0.5f * sr) * acc[1][r], m2 = (0.5f * sr) * acc[2][r], m3 = sr * acc[3][r];
    ex[((RW * 3 + 0) * 16 + r) * 64 + lane] = (m0 + m1) + m2;
    ex[((RW * 3 + 1) * 16 + r) * 64 + lane] = m1 - m2;
    ex[((RW * 3 + 2) * 16 + r) * 64 + lane] = (m1 + m2) + m3;
  }
  __syncthreads();
  float* out = q.partial + (long)blockIdx.x * 9 * 1024;
  for (int o = RW; o < 9; o += 4) {
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
  if (RW == 1) {
    bsum += __shfl_xor(bsum, 32, 64);
    if (h == 0) q.partial_db[blockIdx.x * 32 + li] = bsum;
  }
}

template <int L>
__global__ __launch_bounds__(512, 1) void conv32_wino_bwd_kernel(FusedBwdArgs q) {
  extern __shared__ __attribute__((aligned(16))) char smem_fb[];
  {
    float* tab = reinterpret_cast<float*>(smem_fb + DgGeo<L, true>::COEF_OFF);
    const int i = threadIdx.x;
    if (i < 96) tab[i] = q.dg.bn_coef[i];
    else if (i < 128) tab[i] = q.dg.in_scale[i - 96];
    else if (i < 160) tab[i] = q.dg.in_shift[i - 128];
    else if (i < 192) tab[i] = q.dg.bn_mean[i - 160];
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef FB_ONLY_ROLE                                       // (diagnostic: one role's register count — results are wrong)
  if constexpr (FB_ONLY_ROLE < 4) dgrad_role<FB_ONLY_ROLE & 3, L, true>(q.dg, smem_fb);
  else fused_w_role<FB_ONLY_ROLE & 3, L>(q, smem_fb);
  return;
#endif
  switch (wave) {
    case 0: dgrad_role<0, L, true>(q.dg, smem_fb); break;
    case 1: dgrad_role<1, L, true>(q.dg, smem_fb); break;
    case 2: dgrad_role<2, L, true>(q.dg, smem_fb); break;
    case 3: dgrad_role<3, L, true>(q.dg, smem_fb); break;
    case 4: fused_w_role<0, L>(q, smem_fb); break;
    case 5: fused_w_role<1, L>(q, smem_fb); break;
    case 6: fused_w_role<2, L>(q, smem_fb); break;
    default: fused_w_role<3, L>(q, smem_fb); break;
  }
}

int conv32_wino_bwd_fused_parts(void) { return FB_GRID; }

int conv32_wino_bwd_fused_launch(const float* x, const float* g_a, const float* z, const as_pcl* g, const as_conv_shape* s,
                                 const float* wino_wt, const float* scale, const float* shift, const float* mean, const float* coef,
                                 float slope, const float* next_z, const float* next_scale, const float* next_shift,
                                 const float* next_mean, float* g_x, float* partial, float* partial_db, double* next_partial,
                                 void* stream) {
  static AsPerDevice attr_set[4];
  const int L = s->dil == 1 ? 0 : (s->dil == 2 ? 1 : (s->dil == 4 ? 2 : 3));
  const void* fn = L == 0 ? reinterpret_cast<const void*>(conv32_wino_bwd_kernel<0>)
                 : L == 1 ? reinterpret_cast<const void*>(conv32_wino_bwd_kernel<1>)
                 : L == 2 ? reinterpret_cast<const void*>(conv32_wino_bwd_kernel<2>)
                          : reinterpret_cast<const void*>(conv32_wino_bwd_kernel<3>);
  const int lds = L == 0 ? FbGeo<0>::LDS : (L == 1 ? FbGeo<1>::LDS : (L == 2 ? FbGeo<2>::LDS : FbGeo<3>::LDS));
  static_assert(FbGeo<3>::LDS <= 163840 && FbGeo<0>::LDS <= 163840, "one workgroup per CU");
  if (!attr_set[L].get()) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { as_set_error("as_conv32_wino_bwd_fused: %s", hipGetErrorString(e)); return AS_ERR_LAUNCH; }
    attr_set[L].set();
  }
  FusedBwdArgs a;
  a.dg.z = z; a.dg.g_a = g_a; a.dg.g_z = nullptr; a.dg.g_x = g_x; a.dg.wq = wino_wt;
  a.dg.in_scale = scale; a.dg.in_shift = shift; a.dg.bn_mean = mean; a.dg.bn_coef = coef;
  a.dg.nz = next_z; a.dg.n_scale = next_scale; a.dg.n_shift = next_shift; a.dg.n_mean = next_mean; a.dg.n_partial = next_partial;
  a.dg.g = as_make_dev(g);
  a.dg.nseg = (g->W + 63) / 64;
  long pairs = 0;
  for (int r = 0; r < s->dil; ++r) pairs += ((g->H - r + s->dil - 1) / s->dil + 1) / 2;
  a.dg.pairs = (int)pairs; a.dg.slope = slope;
  a.x = x; a.partial = partial; a.partial_db = partial_db;
#ifdef FB_TIMING_BUILD
  static long long* timing_buf = nullptr;
  const size_t timing_bytes = (size_t)FB_GRID * 8 * 10 * 8;
  if (!timing_buf) (void)hipMalloc(&timing_buf, timing_bytes);
  (void)hipMemsetAsync(timing_buf, 0, timing_bytes, (hipStream_t)stream);
  a.dg.timing = timing_buf;
#endif
  void* kargs[] = {&a};
  hipError_t le = hipLaunchKernel(fn, dim3(FB_GRID), dim3(512), kargs, lds, (hipStream_t)stream);
  if (le != hipSuccess) { as_set_error("as_conv32_wino_bwd_fused: launch failed: %s", hipGetErrorString(le)); return AS_ERR_LAUNCH; }
#ifdef FB_TIMING_BUILD
  if (getenv("AS_FB_TIMING")) {                            // dump THIS launch (synchronises: diagnostic build only)
    (void)hipStreamSynchronize((hipStream_t)stream);
    void* hbuf = malloc(timing_bytes); (void)hipMemcpy(hbuf, timing_buf, timing_bytes, hipMemcpyDeviceToHost);
    FILE* f = fopen("gpurun_out/fused_bwd_timing.bin", "wb"); if (f) { fwrite(hbuf, 1, timing_bytes, f); fclose(f); } free(hbuf);
  }
#endif
  return AS_OK;
}
