// Backward of one full-resolution refinement layer (3x3, any dilation <= 8, stride 1, 32->32 with BatchNorm + LeakyReLU and
// skip connection: stereo_net.py:10-18, 33-51, 97) in ONE launch:
//   g_z   = stage 3 of the layer's BatchNorm backward, (g_a * lrelu'(z*scale+shift) - k1 - (z-mean)*k2) * k3
//   g_x   = dgrad(g_z) + g_a                       (data gradient + skip connection)
//   dW,db = wgrad(x, g_z)                          (weight / bias gradient, per-workgroup slabs)
//   sums  = stage 1 of the NEXT BatchNorm backward (whose output gradient g_x is), from the g_x register tile
// Before: two launches (conv32_wgrad_lds2_kernel<true>: x, g_a, z read, g_z written; conv32_lds_kernel<3,true>: g_z, g_a,
// z_next read, g_x written) = 8 tensor passes of 238 MB at 4 pairs, 364 us each at 62-63 % of the fp32 matrix peak.  Here g_z
// never leaves the chip: x, g_a, z, z_next are read once and g_x written once (5 passes).
//
// How both gradients come out of one staged copy:
//   * a workgroup walks a COMB of rows, y = r, r+d, r+2d, .. of a 64-pixel column segment: the data gradient of row y needs
//     the g_z rows y-d, y, y+d — the previous, the current and the next row of the comb — so a ring of three g_z rows in LDS
//     is all it takes for ANY dilation, and every g_a / z row is fetched and turned into g_z exactly once per comb piece
//     (+2 rows of run-in); the launch's tiles are cut into equal ranges, one per workgroup (see the role function);
//   * the weight gradient is paired the other way round,  dW[kh,kw] = sum_u x[y][u]^T g_z[y-(kh-1)d][u-(kw-1)d]  (the same
//     sum as sum_v x[v+off]^T g_z[v], re-indexed): the CENTRE x row meets the three g_z rows that are staged anyway, so x
//     needs one row in LDS, not three;
//   * g_z is kept ONCE, with a voxel pitch of 144 bytes instead of 128 (round 3; until then two copies, a swizzled and a plain
//     one): consecutive voxels start 4 banks apart, so the data gradient's ds_read_b128 (lane = voxel) is conflict-free for any
//     tap offset, the weight gradient's ds_read_b32 (32 lanes = one voxel's 32 channels) is conflict-free as it always was,
//     and every address of the tile loop is a per-lane constant + a uniform + an immediate — no swizzle arithmetic on the
//     vector ALU next to the matrix waves (fact 1 of DESIGN 4: it costs matrix-pipe time), half the LDS writes of a conversion;
//   * the 27 KB this frees hold a ring of three RAW g_a rows: the skip connection of the data gradient (g_x = dgrad + g_a)
//     reads the row the weight-gradient waves fetched for the conversion two tiles earlier instead of fetching it from HBM a
//     second time (PMC: 1.38x the algorithmic bytes before);
//   * waves 0-1 run the data gradient of the segment's two 32-pixel halves (weights resident: 144 registers), waves 2-3 the
//     weight gradient of the same halves (nine accumulators: 144 registers): 144 MFMAs per wave and tile either way;
//     77 KB of LDS, two workgroups per CU cover each other's barrier waits and element-wise passes.
// Measured at 4 pairs: 605-630 us per layer = 70-72 % of the fp32 matrix peak for both gradients (DESIGN 4 has the phase
// costs and what the first versions lost where).
#include "as_common.h"
#include "conv32_bwd.h"
#include <cstdio>
#include <cstdlib>

#define BW_W 80                         // staged voxels per row: 8 + 64 + 8
#define BW_PITCH 144                    // bytes per staged g_z voxel (128 + 16: consecutive voxels 4 banks apart)
#define BW_ROW_BYTES (BW_W * BW_PITCH)  // 11,520
#define BW_SLOT_BYTES BW_ROW_BYTES
#define BW_X_OFF (3 * BW_SLOT_BYTES)    // 34,560: two x rows (double buffer) of the segment's own 64 voxels (the transposed
#define BW_XROW_BYTES (64 * 128)        //         pairing shifts g_z, not x: no halo), plain 128-byte voxels (LDS-DMA)
#define BW_GA_OFF (BW_X_OFF + 2 * BW_XROW_BYTES)     // 50,944: ring of three raw g_a rows (the segment's own 64 voxels)
#define BW_COEF_OFF (BW_GA_OFF + 3 * BW_XROW_BYTES)  // 75,520: k1, k2, k3, scale, shift, mean [6][32]
#define BW_LDS_BYTES (BW_COEF_OFF + 6 * 128)          // 76,288
#ifndef BW_GRID
#define BW_GRID 512                     // two resident workgroups per CU (tests/tools/grid_sweep.sh rebuilds with EXTRA=-DBW_GRID=n)
#endif
static int bw_grid(void) { return BW_GRID; }

struct BwdArgs {
  const float* x;          // layer input a_{l-1} (PCL)
  const float* ga;         // gradient w.r.t. the layer output a_l
  const float* z;          // pre-activation z_l
  const float* wq;         // packed transposed weights [9][4][64][4] (as_conv32_pack_weights, transpose_flip = 1)
  const float* bn_scale;   // this layer's BatchNorm: scale, shift, mean, stage-3 coefficients [96] = k1, k2, k3
  const float* bn_shift;
  const float* bn_mean;
  const float* bn_coef;
  const float* nz;         // next BatchNorm backward (the layer below): pre-activation, scale, shift, mean
  const float* n_scale;
  const float* n_shift;
  const float* n_mean;
  float* gx;               // g_x = dgrad(g_z) + g_a
  float* partial;          // [BW_GRID][9][32][32]
  float* partial_db;       // [BW_GRID][32]
  double* n_partial;       // [BW_GRID][64]: sum g_y, sum g_y*(z-mean) of the next BatchNorm
  PclDev g;
  int dil, nseg;
  float slope;
#ifdef BW_TIMING_BUILD
  long long* timing;       // diagnostic build only: [workgroup][wave][8] cycle counts per phase
#endif
};

// Diagnostic build (make EXTRA=-DBW_TIMING_BUILD): every wave adds up the shader cycles it spends in each phase; the launch
// dumps them to gpurun_out/bwd_timing.bin (tests/tools/microbench_bwd.py prints the averages).
#ifdef BW_TIMING_BUILD
#define BW_T(slot) do { const long long now_ = clock64(); tacc[slot] += now_ - tlast; tlast = now_; } while (0)
#else
#define BW_T(slot) do { } while (0)
#endif

typedef __attribute__((address_space(3))) void* bw_lds_t;

__device__ inline void bw_dma_1kb(const float* sbase, unsigned voff, unsigned m0) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               :: "s"(m0), "v"(voff), "s"(sbase) : "memory", "m0");
}
template <int IMM> __device__ inline void bw_store_imm(float* sbase, unsigned voff, float v) {
  asm volatile("global_store_dword %0, %1, %2 offset:%3" :: "v"(voff), "v"(v), "s"(sbase), "n"(IMM) : "memory");
}
template <int IMM> __device__ inline void bw_load_imm(float& v, const float* sbase, unsigned voff) {
  asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}
#define BW_ROW_IMM(r) ((((r) & 3) + 8 * ((r) >> 2)) * 128)
#define BW_FOR_ROWS(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

__device__ inline f32x4 bw_chunk(const char* row, int v, int h, int q) {
  return *reinterpret_cast<const f32x4*>(row + v * BW_PITCH + (4 * h + q) * 16);
}

// One role per wave, each a separate instantiation of the whole unit / tile loop (written as one body with a run-time
// `if (wave < 2)` around the role sections, hipcc kept the data-gradient waves' resident weights AND the weight-gradient
// waves' accumulators alive at once: 288 registers + temporaries, 340 spills).  Both roles execute the same barriers.
//
// Tile j of a comb (g_z rows j-1, j, j+1 in the ring, x row j in x buffer j&1), two barriers:
//   top      data-gradient waves request x row j+1 (LDS-DMA into x buffer (j+1)&1); weight-gradient waves request the g_a
//            and z chunks of row j+2 (plain coalesced loads into 40 registers: the latency hides behind the matrix phase)
//   matrix   144 MFMAs per wave: data gradient of row j | weight gradient of x row j
//   B1       everybody is done with row j-1's slot
//   vector   data-gradient waves: skip connection, store g_x, next-BatchNorm sums | weight-gradient waves: stage 3 of the
//            BatchNorm backward on the prefetched chunks -> g_z row j+2 into the freed slot, both layouts
//   B2
// (First version: every row staged by LDS-DMA and converted between two extra barriers at the top of its tile — four
// barriers and an exposed DMA round trip per tile; 765 us per layer against 728 for the two launches it replaces.)
template <bool DGRAD>
__device__ __forceinline__ void conv32_bwd_role(const BwdArgs& p, char* smem) {
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((bw_lds_t)smem));
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, li = lane & 31;
  const unsigned lane16 = (unsigned)lane * 16u;
  const int H = p.g.H, W = p.g.W, Wp = p.g.Wp, d = p.dil;
  const int half = wave & 1;                           // which 32-pixel half of the segment

  // ---- role state ----  ONE block of 144 registers per wave: the data-gradient waves keep all nine taps' B fragments in
  // it (R[tap][4q+e] = chunk q, element e), the weight-gradient waves their nine accumulators
  f32x16 R[9];
  float bn_sc = 0.f, bn_sh = 0.f, bn_mu = 0.f, bn_dy = 0.f, bn_dx = 0.f;     // data gradient: next-BatchNorm sums per lane
  float bsum = 0.f;                                                           // weight gradient: bias gradient
  if constexpr (DGRAD) {
    const float* wb = p.wq + lane * 4;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(wb + tp * 1024 + q * 256);
        R[tp][4 * q + 0] = t4.x; R[tp][4 * q + 1] = t4.y; R[tp][4 * q + 2] = t4.z; R[tp][4 * q + 3] = t4.w;
      }
    bn_sc = p.n_scale[li]; bn_sh = p.n_shift[li]; bn_mu = p.n_mean[li];
  } else {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) R[t][r] = 0.f;
  }
  // this layer's BatchNorm constants: a 768-byte table in LDS (read from global memory at every row conversion, their
  // round trip was the longest thing in the vector phase: 40 % of the weight-gradient waves' time)
  if constexpr (!DGRAD) {
    float* tab = reinterpret_cast<float*>(smem + BW_COEF_OFF);
    const int i = threadIdx.x & 127;
    if (i < 96) tab[i] = p.bn_coef[i];
    if (i < 32) { tab[96 + i] = p.bn_scale[i]; tab[128 + i] = p.bn_shift[i]; tab[160 + i] = p.bn_mean[i]; }
  }
  __syncthreads();
  // row conversion (weight-gradient waves): thread t2 = tid - 128 owns chunks f = t2 + 128k, k < 5 (voxel f >> 3, channel
  // group t2 & 7) of an 80-voxel row
  const int t2 = threadIdx.x & 127;
  const int c4 = (t2 & 7) * 4;
  const int cv_gz = (t2 >> 3) * BW_PITCH + (t2 & 7) * 16;        // + 16 voxels (2,304 bytes) per k
  const int cv_ga = ((t2 >> 3) - 8) * 128 + (t2 & 7) * 16;       // raw g_a row: staged voxel v -> own voxel v - 8 (+ 2,048 per k)
  const unsigned io_off = (unsigned)(512 * h + 4 * li);

#ifdef BW_TIMING_BUILD
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = clock64();
  const long long wall0 = wall_clock64();
#endif
  // Work split: the launch's tiles in one line — (pair, segment) blocks of H rows, inside a block the combs r = 0..d-1 one
  // after the other — cut into gridDim.x equal ranges; a workgroup walks the one to three comb pieces of its range (each
  // piece pays a run-in of three row conversions).  Round-robin units of equal row count, the first plan, left 6-22 % of
  // the workgroup slots idle (4 pairs: 480 units of 63 rows on 512 slots).
  const long t_total = (long)p.g.B * p.nseg * H;
  long t_next = t_total * blockIdx.x / gridDim.x;
  const long t_end = t_total * (blockIdx.x + 1) / gridDim.x;
  while (t_next < t_end) {
    BW_T(0);
    const int blk = (int)(t_next / H);
    int j0 = (int)(t_next - (long)blk * H);
    int r0 = 0, nrow = (H + d - 1) / d;                   // rows of comb r0
    while (j0 >= nrow) { j0 -= nrow; ++r0; nrow = (H - r0 + d - 1) / d; }
    const int j1 = (int)min((long)nrow, j0 + (t_end - t_next));
    t_next += j1 - j0;
    const int seg = blk % p.nseg;
    const int b = blk / p.nseg;
    const int x_new = 64 * seg;
    const int x0 = min(x_new, W - 64);
    const long img = (long)b * p.g.Hp;                     // padded row index base of pair b
    const int px0 = x0 - 8 + p.g.pw;

    // weight-gradient waves: fetch the g_a / z chunks of comb row jj (rows outside the image: any valid row, the
    // conversion then writes zeros — no branch around a load, hipcc's wait bookkeeping stays exact)
    f32x4 pga[5], pz[5];
    auto fetch_row = [&](int jj) {
      const int y = min(max(r0 + jj * d, 0), H - 1);
      const float* garow = p.ga + ((img + y + p.g.ph) * Wp + px0) * 32;
      const float* zrow = p.z + ((img + y + p.g.ph) * Wp + px0) * 32;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        pga[k] = *reinterpret_cast<const f32x4*>(garow + (t2 + 128 * k) * 4);
        pz[k] = *reinterpret_cast<const f32x4*>(zrow + (t2 + 128 * k) * 4);
      }
    };
    auto convert_row = [&](int jj) {                       // -> g_z ring slot (jj + 1) % 3, raw g_a ring slot (jj + 3) % 3
      const int y = r0 + jj * d;
      // destinations are lane constants + the slot + a multiple of k
      char* dst_gz = smem + ((jj + 1) % 3) * BW_SLOT_BYTES + cv_gz;
      char* dst_ga = smem + BW_GA_OFF + ((jj + 3) % 3) * BW_XROW_BYTES + cv_ga;
      if (y < 0 || y >= H) {                               // (workgroup-uniform; no memory instruction but LDS writes inside)
#pragma unroll
        for (int k = 0; k < 5; ++k) *reinterpret_cast<f32x4*>(dst_gz + k * 16 * BW_PITCH) = (f32x4){0.f, 0.f, 0.f, 0.f};
        return;                                            // (no tile of such a row: its raw g_a is never read)
      }
      const float* tab = reinterpret_cast<const float*>(smem + BW_COEF_OFF) + c4;
      const f32x4 k1 = *reinterpret_cast<const f32x4*>(tab), k2 = *reinterpret_cast<const f32x4*>(tab + 32),
                  k3 = *reinterpret_cast<const f32x4*>(tab + 64);
      const f32x4 bsc = *reinterpret_cast<const f32x4*>(tab + 96), bsh = *reinterpret_cast<const f32x4*>(tab + 128),
                  bmu = *reinterpret_cast<const f32x4*>(tab + 160);
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const f32x4 ga = pga[k], zz = pz[k];
        const f32x4 yy = zz * bsc + bsh;
        const f32x4 gl = ga * p.slope;
        f32x4 gy;
        gy.x = yy.x > 0.f ? ga.x : gl.x; gy.y = yy.y > 0.f ? ga.y : gl.y;
        gy.z = yy.z > 0.f ? ga.z : gl.z; gy.w = yy.w > 0.f ? ga.w : gl.w;
        f32x4 gzv = (gy - k1 - (zz - bmu) * k2) * k3;
        if (k == 0 || k == 4) {                            // only the halo voxels (v < 8, v >= 72) can lie outside the image
          const int xx = x0 - 8 + (t2 >> 3) + 16 * k;
          gzv = (xx >= 0 && xx < W) ? gzv : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        *reinterpret_cast<f32x4*>(dst_gz + k * 16 * BW_PITCH) = gzv;
        // the raw g_a of the segment's own 64 voxels (staged voxels 8..71) for the skip connection two tiles from now
        if (k >= 1 && k <= 3) *reinterpret_cast<f32x4*>(dst_ga + k * 2048) = ga;
        else if (k == 0 ? (t2 >> 3) >= 8 : (t2 >> 3) < 8) *reinterpret_cast<f32x4*>(dst_ga + k * 2048) = ga;
      }
    };
    auto issue_x = [&](int jj) {                           // data-gradient waves: x row of comb index jj -> x buffer jj & 1
      const int y = min(r0 + jj * d, H - 1);               // (jj = j1 is never consumed: any valid row)
      const float* xrow = p.x + ((img + y + p.g.ph) * Wp + px0 + 8) * 32;
      const unsigned dst = lds0 + BW_X_OFF + (unsigned)((jj & 1) * BW_XROW_BYTES);
      for (int i = wave; i < 8; i += 2) bw_dma_1kb(xrow + i * 256, lane16, dst + (unsigned)(i * 1024));
    };

    // ---- run-in: g_z rows j0-1, j0, j0+1 and x row j0 ----
    if constexpr (DGRAD) {
      issue_x(j0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      fetch_row(j0 - 1); convert_row(j0 - 1);
      fetch_row(j0); convert_row(j0);
      fetch_row(j0 + 1); convert_row(j0 + 1);
    }
    __syncthreads();
    BW_T(1);

    for (int j = j0; j < j1; ++j) {
      const int y = r0 + j * d;
      const char* slot_prev = smem + ((j + 0) % 3) * BW_SLOT_BYTES;     // row y-d  (comb index j-1)
      const char* slot_cur = smem + ((j + 1) % 3) * BW_SLOT_BYTES;      // row y
      const char* slot_next = smem + ((j + 2) % 3) * BW_SLOT_BYTES;     // row y+d
      const int xw = x0 + 32 * half;                                      // first pixel of this wave's half
      if constexpr (DGRAD) {
        issue_x(j + 1);
        // ---- data gradient: out[y][x] = sum_{kh,kw} g_z[y+(kh-1)d][x+(kw-1)d] * Wt[kh][kw]  (+ g_a[y][x]) ----
        const long vox0 = (img + y + p.g.ph) * Wp + xw + p.g.pw;
        float* gx_base = p.gx + vox0 * 32;
        const float* nz_base = p.nz + vox0 * 32;
        const int vbase = 8 + 32 * half + li;
        float res[16], zt[16];
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        f32x4 a[4];
        const char* rows[3] = {slot_prev, slot_cur, slot_next};
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) a[cc] = bw_chunk(rows[0], vbase - d, h, cc);
#pragma unroll
        for (int cc = 0; cc < 36; ++cc) {
          if (cc + 3 < 36) {
            const int tp = (cc + 3) >> 2;
            a[(cc + 3) & 3] = bw_chunk(rows[tp / 3], vbase + (tp % 3 - 1) * d, h, (cc + 3) & 3);
          }
#ifndef BW_EXP_NOEPI
          if (cc == 8) {
#define BW_LD(r) bw_load_imm<BW_ROW_IMM(r)>(zt[r], nz_base, io_off);
            BW_FOR_ROWS(BW_LD)
#undef BW_LD
          }
#endif
          __builtin_amdgcn_sched_barrier(0);
          const f32x4 av = a[cc & 3];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, R[cc >> 2][4 * (cc & 3) + 0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, R[cc >> 2][4 * (cc & 3) + 1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, R[cc >> 2][4 * (cc & 3) + 2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, R[cc >> 2][4 * (cc & 3) + 3], acc, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        BW_T(2);
#ifndef BW_EXP_NOB1
        __syncthreads();                                   // B1
#endif
        BW_T(3);
#ifdef BW_EXP_NOEPI
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r) bn_dy += acc[r];
#else
        // the skip connection: raw g_a of row j, left in LDS by the conversion two tiles ago (ring slot (j + 3) % 3)
        {
          const char* garow = smem + BW_GA_OFF + ((j + 3) % 3) * BW_XROW_BYTES + (32 * half + 4 * h) * 128 + li * 4;
#pragma unroll
          for (int r = 0; r < 16; ++r) res[r] = *reinterpret_cast<const float*>(garow + ((r & 3) + 8 * (r >> 2)) * 128);
        }
        // the x row of the next tile and the next layer's pre-activation row are home
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(zt[0]), "+v"(zt[1]), "+v"(zt[2]), "+v"(zt[3]), "+v"(zt[4]), "+v"(zt[5]), "+v"(zt[6]),
                       "+v"(zt[7]), "+v"(zt[8]), "+v"(zt[9]), "+v"(zt[10]), "+v"(zt[11]), "+v"(zt[12]),
                       "+v"(zt[13]), "+v"(zt[14]), "+v"(zt[15]) :: "memory");
        const int dup = x_new - xw;            // pixels below `dup` of this half also belong to the neighbouring segment
#define BW_ST(r) { const float v = acc[r] + res[r];                                          \
                   bw_store_imm<BW_ROW_IMM(r)>(gx_base, io_off, v);                           \
                   const float yv = fmaf(zt[r], bn_sc, bn_sh);                                \
                   float gy = yv > 0.f ? v : v * p.slope;                                     \
                   if (dup > 0) gy = ((r & 3) + 8 * (r >> 2) + 4 * h) >= dup ? gy : 0.f;      \
                   bn_dy += gy; bn_dx = fmaf(gy, zt[r] - bn_mu, bn_dx); }
        BW_FOR_ROWS(BW_ST)
#undef BW_ST
#endif
      } else {
#ifndef BW_EXP_NOCONV
        fetch_row(j + 2);                                  // in flight during the matrix phase
#endif
        // ---- weight gradient: acc[kh*3+kw][ci][co] += sum_u x[y][u][ci] * g_z[y-(kh-1)d][u-(kw-1)d][co] ----
        const int u0 = 8 + 32 * half + h;                                   // voxel of step 0 in the staged g_z rows
        const char* xaddr = smem + BW_X_OFF + (j & 1) * BW_XROW_BYTES + (u0 - 8) * 128 + li * 4;
        const char* gaddr[9];
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
          const char* row = (tp / 3 == 0) ? slot_next : ((tp / 3 == 1) ? slot_cur : slot_prev);
          gaddr[tp] = row + (u0 - (tp % 3 - 1) * d) * BW_PITCH + li * 4;
        }
        const int dup = x_new - xw;            // pixels below `dup` were counted by the neighbouring segment
        float av[2], bv[2][9];
        av[0] = *reinterpret_cast<const float*>(xaddr);
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) bv[0][tp] = *reinterpret_cast<const float*>(gaddr[tp]);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          if (s + 1 < 16) {
            av[(s + 1) & 1] = *reinterpret_cast<const float*>(xaddr + (s + 1) * 256);
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) bv[(s + 1) & 1][tp] = *reinterpret_cast<const float*>(gaddr[tp] + (s + 1) * 2 * BW_PITCH);
          }
          __builtin_amdgcn_sched_barrier(0);
          const bool counted = 2 * s + h >= dup;
          const float xa = counted ? av[s & 1] : 0.f;
          bsum += counted ? bv[s & 1][4] : 0.f;
#pragma unroll
          for (int tp = 0; tp < 9; ++tp)
            R[tp] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa, bv[s & 1][tp], R[tp], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        BW_T(2);
#ifndef BW_EXP_NOB1
        __syncthreads();                                   // B1: nobody reads row j-1's slot any more
#endif
        BW_T(3);
#ifndef BW_EXP_NOCONV
        convert_row(j + 2);
#endif
      }
      BW_T(4);
      __syncthreads();                                     // B2: g_z row j+2 and x row j+1 are in place
      BW_T(5);
    }
  }

  // ---- slabs: weight gradient (wave 3 -> LDS, wave 2 adds and stores), bias gradient, next-BatchNorm sums ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#ifdef BW_TIMING_BUILD
  BW_T(6);
  if (p.timing && lane == 0) {
    long long* o = p.timing + ((long)blockIdx.x * 4 + wave) * 8;
    for (int i = 0; i < 7; ++i) o[i] = tacc[i];
    o[7] = wall_clock64() - wall0;
  }
#endif
  float* slab = reinterpret_cast<float*>(smem);           // [9][16][64] floats = 36,864 B
  if (!DGRAD && wave == 3) {
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
      for (int r = 0; r < 16; ++r) slab[(tp * 16 + r) * 64 + lane] = R[tp][r];
  }
  __syncthreads();
  if (!DGRAD && wave == 2) {
    float* out = p.partial + (long)blockIdx.x * 9 * 1024;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = R[tp][r] + slab[(tp * 16 + r) * 64 + lane];
        const int ci = (r & 3) + 8 * (r >> 2) + 4 * h;
        out[tp * 1024 + ci * 32 + li] = v;
      }
  }
  __syncthreads();
  float* scr = reinterpret_cast<float*>(smem);
  if constexpr (!DGRAD) {
    bsum += __shfl_xor(bsum, 32, 64);
    if (h == 0) scr[(wave - 2) * 32 + li] = bsum;
  } else {
    scr[64 + ((wave * 2 + h) * 2 + 0) * 32 + li] = bn_dy;
    scr[64 + ((wave * 2 + h) * 2 + 1) * 32 + li] = bn_dx;
  }
  __syncthreads();
  if (threadIdx.x < 32) p.partial_db[blockIdx.x * 32 + li] = scr[li] + scr[32 + li];
  if (threadIdx.x >= 64 && threadIdx.x < 128) {
    const int which = (threadIdx.x - 64) >> 5, cch = threadIdx.x & 31;
    double sum = 0.0;
    for (int q = 0; q < 4; ++q) sum += (double)scr[64 + (q * 2 + which) * 32 + cch];
    p.n_partial[(long)blockIdx.x * 64 + which * 32 + cch] = sum;
  }
}

__global__ __launch_bounds__(256, 2) void conv32_bwd_fused_kernel(BwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) < 2) conv32_bwd_role<true>(p, smem_dyn);
  else conv32_bwd_role<false>(p, smem_dyn);
}

// ---- host ------------------------------------------------------------------------------------------------------------
bool conv32_bwd_fused_applicable(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s) {
  if (s->kd != 1 || s->kh != 3 || s->kw != 3 || s->stride != 1) return false;
  if (gin->D != 1 || gout->D != 1) return false;
  if (s->dil < 1 || s->dil > 8 || s->pad_h != s->dil || s->pad_w != s->dil) return false;
  if (gin->B != gout->B || gin->H != gout->H || gin->W != gout->W) return false;
  if (gin->ph != gout->ph || gin->pw != gout->pw || gin->pd != gout->pd) return false;   // one padded geometry for x, g_a, z, g_x
  if (gin->pw < 8 || gin->ph < s->dil) return false;
  if (gout->W < 64) return false;
  // the launch must fill the chip: at least 12 tiles per resident workgroup
  return (long)gout->B * gout->H * ((gout->W + 63) / 64) >= (long)BW_GRID * 12;
}

int conv32_bwd_fused_slabs(void) { return bw_grid(); }

int conv32_bwd_fused_launch(const float* x, const float* g_a, const float* z, const as_pcl* g, const as_conv_shape* s,
                            const float* packed_wt, const float* scale, const float* shift, const float* mean,
                            const float* coef, float slope, const float* next_z, const float* next_scale,
                            const float* next_shift, const float* next_mean, float* g_x, float* partial,
                            float* partial_db, double* next_partial, void* stream) {
  static AsPerDevice attr_set;
  if (!attr_set.get()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv32_bwd_fused_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, BW_LDS_BYTES);
    if (e != hipSuccess) { as_set_error("as_conv32_bwd_fused: %s", hipGetErrorString(e)); return AS_ERR_LAUNCH; }
    attr_set.set();
  }
  BwdArgs a;
  a.x = x; a.ga = g_a; a.z = z; a.wq = packed_wt; a.bn_scale = scale; a.bn_shift = shift; a.bn_mean = mean; a.bn_coef = coef;
  a.nz = next_z; a.n_scale = next_scale; a.n_shift = next_shift; a.n_mean = next_mean;
  a.gx = g_x; a.partial = partial; a.partial_db = partial_db; a.n_partial = next_partial;
  a.g = as_make_dev(g);
  a.dil = s->dil; a.slope = slope;
  a.nseg = (g->W + 63) / 64;
#ifdef BW_TIMING_BUILD
  static long long* timing_buf = nullptr;
  const size_t timing_bytes = (size_t)BW_GRID * 4 * 8 * 8;
  if (!timing_buf) hipMalloc(&timing_buf, timing_bytes);
  hipMemsetAsync(timing_buf, 0, timing_bytes, (hipStream_t)stream);
  a.timing = timing_buf;
#endif
  hipLaunchKernelGGL(conv32_bwd_fused_kernel, dim3(bw_grid()), dim3(256), BW_LDS_BYTES, (hipStream_t)stream, a);
#ifdef BW_TIMING_BUILD
  if (getenv("AS_BW_TIMING")) {                            // dump THIS launch (synchronises: diagnostic build only)
    hipStreamSynchronize((hipStream_t)stream);
    void* hbuf = malloc(timing_bytes); hipMemcpy(hbuf, timing_buf, timing_bytes, hipMemcpyDeviceToHost);
    FILE* f = fopen("gpurun_out/bwd_timing.bin", "wb"); if (f) { fwrite(hbuf, 1, timing_bytes, f); fclose(f); } free(hbuf);
  }
#endif
  return AS_OK;
}
