// conv32, 2-D 3x3 stride-1 instance (any dilation <= 8): the refinement's and the feature trunk's
// 32->32 layers and their data gradients — 83 % of the model's FLOPs (stereo_net.py:10-18, 33-51, 97).
//
// MI355X design
//   * persistent workgroups, TWO per CU (8 waves = two per SIMD), walk 128-pixel row segments; the last segment
//     of a row is shifted left to end at W, so every tile is 128 valid voxels (no ragged-edge code);
//   * the 3 input rows a segment needs (y-d, y, y+d; x0-8 .. x0+135) are DMA'd into LDS
//     (global_load_lds_dwordx4: 1 KB per wave instruction, scalar base + constant lane offset): HBM sees each
//     input voxel once per layer (the two re-reads by neighbouring rows hit L2);
//   * one tile buffer per workgroup: while one workgroup waits for its DMA or runs its epilogue the other one owns
//     the matrix cores — the overlap comes from the second workgroup, not from double buffering;
//   * LDS image is lane-linear (a DMA cannot pad), so bank conflicts are removed by swizzling the
//     SOURCE: LDS slot s of voxel v holds channel chunk s ^ ((v>>1)&7); a ds_read_b128 of one chunk
//     from 16 consecutive voxels then hits 16 distinct 16-byte bank groups;
//   * the weights' B fragments stay in registers for the whole launch (7-9 of 9 taps, the rest re-read from L2
//     48 MFMAs ahead of use); activations come through a 4-deep ring of ds_read_b128;
//   * vector-ALU instructions of one wave do not overlap the MFMAs of the other wave on a SIMD (measured: the times
//     add), so the per-tile path carries almost none: DMA, output stores and residual loads are scalar base +
//     constant lane offset + immediate (inline asm, which also keeps hipcc from draining them with vmcnt(0)),
//     retired by counted waits; BatchNorm moments are per-lane shifted sums (no barrier, no LDS in the loop);
//   * tiles are banded per XCD (workgroup b is on XCD b%8 under round-robin dispatch — speed only):
//     the three uses of an input row happen close in time on one XCD's L2;
//   * one BatchNorm partial per workgroup (512 per layer instead of 15,000).
// LDS per workgroup: 55,296 (tile) + 1,024; two workgroups per CU use 112 KB of the CU's 160 KB.
// Measured (4 pairs, 375x1242): 116-121 TFLOP/s forward, 106-114 with the skip connection.
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
  // MODE 3: stage 1 of the NEXT BatchNorm backward, fused (its g_a is this kernel's output)
  const float* bn_z;       // PCL, output geometry: pre-activation of the layer whose output gradient this kernel writes
  const float* bn_scale;   // [32] that layer's scale / shift / mean
  const float* bn_shift;
  const float* bn_mean;
  double* bn_partial;      // [workgroups][64]: sum g_y, sum g_y*(z-mean) per channel (layout of bn_bwd_reduce_kernel)
#ifdef AS_LDS_TRACE_BUILD
  unsigned long long* trace;           // diagnostic build only: [wg][32 tiles][8] 100-MHz timestamps
#endif
};

// Diagnostic build (make TRACE=1): wave 0 of every workgroup timestamps the phases of its first 32 tiles;
// the launch selected by AS_LDS_TRACE=<n> dumps them to gpurun_out/lds_trace.bin (tests/tools/lds_trace_view.py).
#ifdef AS_LDS_TRACE_BUILD
#define AS_TRACE(slot) do { if (p.trace && threadIdx.x == 0 && it < 32) p.trace[((long)blockIdx.x * 32 + it) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define AS_TRACE(slot) do { } while (0)
#endif

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// Row segment of a tile.  The last segment of a row is shifted LEFT to end exactly at W (x0 = W-128), so every
// tile is 128 valid voxels wide: no ragged edge anywhere in the kernel.  The voxels it shares with its
// neighbour ([x0, nominal start)) are computed twice and stored twice with identical values; only the
// BatchNorm moments have to skip them (x_new = first voxel that is this tile's alone).
__device__ inline void tile_coords(const ConvLdsArgs& p, int tile, int& b, int& y, int& x0, int& x_new) {
  const int row = tile / p.tiles_per_row;
  x_new = (tile - row * p.tiles_per_row) * 128;
  x0 = min(x_new, p.gout.W - 128);
  b = row / p.gout.H;
  y = row - b * p.gout.H;
}

// One 1-KB LDS-DMA instruction: 64 lanes x 16 B from  sbase + voff  to LDS byte  m0 + 16*lane.
// Written as inline asm ON PURPOSE: with a visible global_load_lds in the kernel hipcc (ROCm 7.2) waits
// vmcnt(0) before the first ds_read of every tile and at every use of an ordinary load — vmcnt is in-order,
// so each of those also waits for the DMA of the NEXT tile and for the output stores.  Hidden, the compiler
// counts only its own loads (its waits stay correct, merely conservative) and the DMA is retired by the
// explicit counted waits in the kernel.
__device__ inline void dma_1kb(const float* sbase, unsigned voff, unsigned m0) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               :: "s"(m0), "v"(voff), "s"(sbase) : "memory", "m0");
}

// Stage the three rows of `tile`: 54 wave instructions; wave w takes columns i = w, w+4, .. of every row
// (i&1 == w&1, so the swizzled lane offset is one per-wave constant): scalar base per instruction, constant
// lane offset, no vector ALU work.  (On this chip vector-ALU instructions of one wave do NOT overlap the
// MFMAs of the other wave on the SIMD — measured: their times add — so every VALU instruction in the
// per-tile path costs matrix-core time.)
__device__ inline void issue_tile_dma(const ConvLdsArgs& p, int tile, unsigned lds0, int wave, unsigned lane_off) {
  int b, y, x0, x_new;
  tile_coords(p, tile, b, y, x0, x_new);
  const int Wp = p.gin.Wp;
  const long rowvox = ((long)b * p.gin.Hp + (y + p.gin.ph - p.dil)) * Wp + (x0 - 8 + p.gin.pw);
  const float* row0 = p.x + rowvox * 32;                                 // wave-uniform
  const long row_stride = (long)p.dil * Wp * 32;                         // floats
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float* srow = row0 + r * row_stride;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int i = wave + 4 * k;
      if (i < TL_DMA_PER_ROW)                                            // wave-uniform
        dma_1kb(srow + i * 256, lane_off, lds0 + (unsigned)((r * TL_W + 8 * i) * 128));
    }
  }
}

// Output / residual accesses of a wave tile: scalar base + one constant lane
// offset + the row as an instruction immediate — no address arithmetic.  Inline asm for the same reason as the
// DMA (the stores stay out of hipcc's vmcnt bookkeeping; the kernel's explicit waits retire them).
template <int IMM> __device__ inline void store_imm(float* sbase, unsigned voff, float v) {
  asm volatile("global_store_dword %0, %1, %2 offset:%3" :: "v"(voff), "v"(v), "s"(sbase), "n"(IMM) : "memory");
}
template <int IMM> __device__ inline void load_imm(float& v, const float* sbase, unsigned voff) {
  asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}
#define AS_ROW_IMM(r) ((((r) & 3) + 8 * ((r) >> 2)) * 128)
#define AS_FOR_ROWS(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

__device__ inline f32x4 lds_chunk(const char* buf, int r, int v, int h, int q) {
  return *reinterpret_cast<const f32x4*>(buf + (r * TL_W + v) * 128 + (((4 * h + q) ^ ((v >> 1) & 7)) << 4));
}

__device__ inline void glb_load_w(f32x4 (&r)[4], const float* p) {
  r[0] = *reinterpret_cast<const f32x4*>(p);
  r[1] = *reinterpret_cast<const f32x4*>(p + 256);
  r[2] = *reinterpret_cast<const f32x4*>(p + 512);
  r[3] = *reinterpret_cast<const f32x4*>(p + 768);
}

// Kernel flavours (compile time, so that no path carries another one's waits):
//   MODE 0: raw output + BatchNorm moments (training forward)   MODE 1: lrelu(acc*scale+shift) (eval forward)
//   MODE 2: raw output, no moments (data gradients)             RES 1: + residual in the output geometry
//   RES 2: the residual IS the layer's input (a BasicBlock's skip connection in the eval forward): it is the centre
//          row of the staged tile, read from LDS before the buffer is released — no third tensor stream at all
//   MODE 3: MODE 2 + stage 1 of the following BatchNorm backward: the output IS that layer's g_a, so the sums
//           sum g_y and sum g_y*(z-mean), g_y = g_a * lrelu'(z*scale+shift), are taken from the register tile
//           (saves bn_bwd_reduce_kernel's pass over g_a and z: 477 MB per full-resolution layer at 4 pairs)
template <int MODE, int RES>
__device__ __forceinline__ void conv32_lds_body(const ConvLdsArgs& p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tile_buf = smem;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((lds_ptr_t)tile_buf));

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, li = lane & 31;
  const float* wb = p.wq + lane * 4;
  const unsigned io_off = (unsigned)(512 * h + 4 * li);   // byte offset of (row 4h, channel li) in a wave tile

  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int t_begin = xcd * p.tiles_per_band;
  const int t_end = min(t_begin + p.tiles_per_band, p.ntiles);
  const float bias_v = p.ep.bias ? p.ep.bias[li] : 0.f;
  float sc = 1.f, sh = 0.f;
  if (MODE == 1) { sc = p.ep.ep_scale[li]; sh = p.ep.ep_shift[li]; }
  // BatchNorm moments: every lane keeps shifted sums of ITS elements over all of the workgroup's tiles
  // (3 VALU per element, no barrier, no LDS in the tile loop); shift = the lane's first element, so the sums
  // stay small and  M2 = s2 - s1^2/n  loses nothing.  One Chan merge per workgroup at the end.
  float st_n = 0.f, st_c = 0.f, st_s1 = 0.f, st_s2 = 0.f;
  float bn_sc = 0.f, bn_sh = 0.f, bn_mu = 0.f, bn_dy = 0.f, bn_dx = 0.f;
  if (MODE == 3) { bn_sc = p.bn_scale[li]; bn_sh = p.bn_shift[li]; bn_mu = p.bn_mean[li]; }
  // DMA lane constant: voxel vl of an 8-voxel group, channel chunk swizzled by the voxel (see file header)
  const unsigned dma_lane_off =
      (unsigned)(lane >> 3) * 128u + (unsigned)(((lane & 7) ^ (((wave & 1) << 2) | (lane >> 4))) << 4);

  // B fragments of taps 0..RESIDENT-1 stay in registers for the whole launch; the others do not fit next to
  // the rest of the working set and are re-read from L2 every tile, 48 MFMAs before their use.
  constexpr int RESIDENT = (RES ? 7 : (MODE == 0 ? 8 : 9)) - (MODE == 3 ? 3 : 0);              // the residual tile needs 16 registers of its own
  constexpr int W_LEAD = 12;                          // streamed taps are requested 12 chunks (48 MFMAs) ahead
  f32x4 w[9][4];
#pragma unroll
  for (int tp = 0; tp < RESIDENT; ++tp) glb_load_w(w[tp], wb + tp * 1024);

  int tile = t_begin + j;
  if (tile < t_end) issue_tile_dma(p, tile, lds0, wave, dma_lane_off);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef AS_LDS_TRACE_BUILD
  int it = -1;
#endif
  for (; tile < t_end; tile += p.wg_per_xcd) {
#ifdef AS_LDS_TRACE_BUILD
    ++it;
#endif
    int b, y, x0, x_new;
    tile_coords(p, tile, b, y, x0, x_new);
    const int xw = x0 + 32 * wave;                      // wave-uniform
    const long vox0 = p.gout.vox(b, 0, y, xw);
    const int vbase = 32 * wave + li + 8;
    float* z_base = p.ep.z + vox0 * 32;                 // wave-uniform
    const float* res_base = RES == 1 ? p.ep.residual + vox0 * 32 : nullptr;
    const float* bnz_base = MODE == 3 ? p.bn_z + vox0 * 32 : nullptr;
    float zt[16];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias_v;
    f32x4 a[4];                        // ring of A chunks, prefetch distance 3 (= 12 MFMAs = 768 cycles)
    float res[16];
    AS_TRACE(0);
    // Retire this tile's DMA (issued before the previous tile's epilogue) but NOT that epilogue's 16 output
    // stores, which are younger: vmcnt counts loads, stores and LDS-DMA together in issue order, so "at most
    // 16 outstanding" = everything older than the 16 stores has landed.  Every wave issues exactly 16 stores
    // per tile.  (First tile: everything was retired before the loop, the wait falls through.)
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    AS_TRACE(1);
    __syncthreads();
    AS_TRACE(2);
    int offs = RESIDENT * 1024;
    asm volatile("" : "+s"(offs));                      // keep hipcc from hoisting (and then spilling) the addresses
#pragma unroll
    for (int tp = RESIDENT; tp < 9; ++tp)
      if (4 * tp - W_LEAD <= 0) glb_load_w(w[tp], wb + offs + (tp - RESIDENT) * 1024);
#pragma unroll
    for (int c = 0; c < 3; ++c) a[c] = lds_chunk(tile_buf, 0, vbase - p.dil, h, c);
#pragma unroll
    for (int c = 0; c < 36; ++c) {
      if (c + 3 < 36) {
        const int tp = (c + 3) >> 2;
        a[(c + 3) & 3] = lds_chunk(tile_buf, tp / 3, vbase + (tp % 3 - 1) * p.dil, h, (c + 3) & 3);
      }
#pragma unroll
      for (int tp = RESIDENT; tp < 9; ++tp)
        if (4 * tp - W_LEAD == c && c > 0) glb_load_w(w[tp], wb + offs + (tp - RESIDENT) * 1024);
      if (RES == 1 && c == 22) {       // the residual tile, fetched while 14 chunks of MFMA work remain
#define AS_LD(r) load_imm<AS_ROW_IMM(r)>(res[r], res_base, io_off);
        AS_FOR_ROWS(AS_LD)
#undef AS_LD
      }
      if (MODE == 3 && c == 26) {      // the next layer's pre-activation tile (for its BatchNorm backward sums)
#define AS_LD(r) load_imm<AS_ROW_IMM(r)>(zt[r], bnz_base, io_off);
        AS_FOR_ROWS(AS_LD)
#undef AS_LD
      }
      __builtin_amdgcn_sched_barrier(0);
      const f32x4 av = a[c & 3], bv = w[c >> 2][c & 3];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // Everything this wave loaded into registers (weights, residual) must be home BEFORE the next DMA goes
    // out: a later wait for it would, vmcnt being in-order, also wait for that DMA.  The residual registers
    // are operands of the wait so that no use of them can be scheduled above it.
    AS_TRACE(3);
    if (RES == 2) {
      // centre row (row 1) of the staged tile, voxel 8 + 32*wave + row(r, h), channel li: chunk li>>2 sits in slot
      // (li>>2) ^ ((v>>1)&7); the 32 lanes of a half read one voxel's 128 bytes (a permutation of its 8 slots)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int v = 8 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
        res[r] = *reinterpret_cast<const float*>(tile_buf + (TL_W + v) * 128 + ((((li >> 2) ^ ((v >> 1) & 7))) << 4) + (li & 3) * 4);
      }
    }
    if (RES == 1) {
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(res[0]), "+v"(res[1]), "+v"(res[2]), "+v"(res[3]), "+v"(res[4]), "+v"(res[5]), "+v"(res[6]),
                     "+v"(res[7]), "+v"(res[8]), "+v"(res[9]), "+v"(res[10]), "+v"(res[11]), "+v"(res[12]),
                     "+v"(res[13]), "+v"(res[14]), "+v"(res[15]) :: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (MODE == 3) {                   // same wait, the z tile as its operands (all loads are home by now)
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(zt[0]), "+v"(zt[1]), "+v"(zt[2]), "+v"(zt[3]), "+v"(zt[4]), "+v"(zt[5]), "+v"(zt[6]),
                     "+v"(zt[7]), "+v"(zt[8]), "+v"(zt[9]), "+v"(zt[10]), "+v"(zt[11]), "+v"(zt[12]),
                     "+v"(zt[13]), "+v"(zt[14]), "+v"(zt[15]) :: "memory");
    }
    __syncthreads();                   // every wave has read its operands: the tile buffer is free
    AS_TRACE(4);
    const int next_tile = tile + p.wg_per_xcd;
    if (next_tile < t_end) issue_tile_dma(p, next_tile, lds0, wave, dma_lane_off);
    else if ((MODE == 0 || MODE == 2) && RES == 0) {
      // These flavours store the accumulator straight out of the last MFMA through inline asm.  A 16-pass MFMA's result may
      // be read by a vector-memory instruction 18 wait states later at the earliest; hipcc inserts those for its own stores
      // but cannot see into inline asm.  With a next tile the DMA issue above is far longer than that; without one (the
      // workgroup's last tile) nothing else stands between the barrier and the stores.
      asm volatile("s_nop 15\n\ts_nop 1" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    AS_TRACE(5);

    // ---- epilogue: exactly 16 store instructions per wave, nothing that waits on vector memory ----
    const int dup3 = MODE == 3 ? x_new - xw : 0;          // rows below dup3 also belong to the neighbouring tile
#define AS_ST(r) { float v = acc[r];                                                        \
                   if (MODE == 1) { v = v * sc + sh; v = fmaxf(v, v * p.ep.slope); }         \
                   if (RES) v += res[r];                                                      \
                   store_imm<AS_ROW_IMM(r)>(z_base, io_off, v);                               \
                   if (MODE == 3) {                                                           \
                     const float y = fmaf(zt[r], bn_sc, bn_sh);                               \
                     float gy = y > 0.f ? v : v * p.ep.slope;                                 \
                     if (dup3 > 0) gy = ((r & 3) + 8 * (r >> 2) + 4 * h) >= dup3 ? gy : 0.f; \
                     bn_dy += gy; bn_dx = fmaf(gy, zt[r] - bn_mu, bn_dx); } }
    AS_FOR_ROWS(AS_ST)
#undef AS_ST
    if (MODE == 0) {
      const int dup = x_new - xw;      // wave-uniform: rows below `dup` also belong to the neighbouring tile
      if (dup <= 0) {
        st_c = st_n == 0.f ? acc[0] : st_c;   // data-dependent on purpose: an iteration-count test makes hipcc peel the loop
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = acc[r] - st_c; st_s1 += d; st_s2 = fmaf(d, d, st_s2); }
        st_n += 16.f;
      } else if (dup < 32) {
        st_c = st_n == 0.f ? acc[15] : st_c;  // row 27 + 4h: new whenever the wave has any new row of this half
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
          const float d = row >= dup ? acc[r] - st_c : 0.f;
          st_s1 += d; st_s2 = fmaf(d, d, st_s2); st_n += row >= dup ? 1.f : 0.f;
        }
      }
    }
    AS_TRACE(6);
#ifdef AS_LDS_TRACE_BUILD
    if (p.trace && threadIdx.x == 0 && it == 0) p.trace[((long)blockIdx.x * 32) * 8 + 7] =
        ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
#endif
  }
  if (MODE == 3) {
    // lane sums -> one [64] slab per workgroup in the layout bn_bwd_finalize_kernel reads (fixed order)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* part = reinterpret_cast<float*>(smem);      // [8][2][32]
    part[((wave * 2 + h) * 2 + 0) * 32 + li] = bn_dy;
    part[((wave * 2 + h) * 2 + 1) * 32 + li] = bn_dx;
    __syncthreads();
    if (threadIdx.x < 64) {
      const int which = threadIdx.x >> 5, c = threadIdx.x & 31;
      double sum = 0.0;
      for (int q = 0; q < 8; ++q) sum += (double)part[(q * 2 + which) * 32 + c];
      p.bn_partial[(long)blockIdx.x * 64 + which * 32 + c] = sum;
    }
  }
  if (MODE == 0 && p.ep.stat_mean != nullptr) {
    // lane sums -> (n, mean, M2) -> one partial per workgroup: 8 (wave, half) partials per channel, merged in
    // fixed order (deterministic).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                   // the tile buffer is no longer in use
    float* part = reinterpret_cast<float*>(smem);      // [8][32][3]
    const float mean_l = st_n > 0.f ? st_c + st_s1 / st_n : 0.f;
    const float m2_l = st_n > 0.f ? fmaxf(st_s2 - st_s1 * st_s1 / st_n, 0.f) : 0.f;
    float* mine = part + ((wave * 2 + h) * 32 + li) * 3;
    mine[0] = st_n; mine[1] = mean_l; mine[2] = m2_l;
    __syncthreads();
    if (threadIdx.x < 32) {
      TileStats run; run.n = 0.f; run.mean = 0.f; run.m2 = 0.f;
      for (int q = 0; q < 8; ++q) {
        TileStats t;
        t.n = part[(q * 32 + li) * 3]; t.mean = part[(q * 32 + li) * 3 + 1]; t.m2 = part[(q * 32 + li) * 3 + 2];
        stats_merge(run, t);
      }
      stats_write(p.ep, blockIdx.x, run);
    }
  }
}

template <int MODE, bool RES>
__global__ __launch_bounds__(256, 2) void conv32_lds_kernel(ConvLdsArgs p) { conv32_lds_body<MODE, RES ? 1 : 0>(p); }

// eval forward of a BasicBlock: fused BatchNorm + LeakyReLU, skip connection taken from the staged input (RES 2)
__global__ __launch_bounds__(256, 2) void conv32_lds_skip_kernel(ConvLdsArgs p) { conv32_lds_body<1, 2>(p); }

// ---- host ---------------------------------------------------------------------------------------
bool conv32_lds_applicable(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s) {
  if (s->kd != 1 || s->kh != 3 || s->kw != 3 || s->stride != 1) return false;
  if (gin->D != 1 || gout->D != 1) return false;
  if (s->dil < 1 || s->dil > 8 || s->pad_h != s->dil || s->pad_w != s->dil) return false;
  if (gin->H != gout->H || gin->W != gout->W) return false;
  if (gin->pw < 8 || gin->ph < s->dil) return false;      // the staged tile always spans x0-8 .. x0+135
  if (gout->W < 128) return false;                        // every tile is a full 128-voxel row segment
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
                      const float* residual, float* stat_mean, float* stat_m2, float* stat_cnt,
                      const float* bn_z, const float* bn_scale, const float* bn_shift, const float* bn_mean,
                      double* bn_partial, void* stream) {
  ConvLdsArgs a;
  a.bn_z = bn_z; a.bn_scale = bn_scale; a.bn_shift = bn_shift; a.bn_mean = bn_mean; a.bn_partial = bn_partial;
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
  const int mode = bn_partial != nullptr ? 3 : (epilogue == 1 ? 1 : (a.ep.stat_mean != nullptr ? 0 : 2));
  const void* fn = nullptr;
#define AS_LDS_PICK(M, R) (R ? reinterpret_cast<const void*>(conv32_lds_kernel<M, true>) : reinterpret_cast<const void*>(conv32_lds_kernel<M, false>))
  const bool has_res = residual != nullptr;
  // eval forward of a BasicBlock: the residual is the input itself (same buffer, same padded geometry)
  const bool res_self = mode == 1 && residual == x && gin->ph == gout->ph && gin->pw == gout->pw && gin->pd == gout->pd;
  fn = mode == 0 ? AS_LDS_PICK(0, has_res) : mode == 1 ? AS_LDS_PICK(1, has_res) : mode == 2 ? AS_LDS_PICK(2, has_res)
                                                                                              : AS_LDS_PICK(3, has_res);
  if (res_self) fn = reinterpret_cast<const void*>(conv32_lds_skip_kernel);
#undef AS_LDS_PICK
  static AsPerDevice attr_set[9];
  const int fi = res_self ? 8 : mode * 2 + (has_res ? 1 : 0);
  if (!attr_set[fi].get()) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, TL_LDS_BYTES);
    if (e != hipSuccess) {
      as_set_error("conv32_lds: cannot reserve %d bytes of LDS: %s", TL_LDS_BYTES, hipGetErrorString(e));
      return AS_ERR_LAUNCH;
    }
    attr_set[fi].set();
  }
#ifdef AS_LDS_TRACE_BUILD
  static int trace_countdown = -2;
  static unsigned long long* trace_buf = nullptr;
  if (trace_countdown == -2) { const char* e = getenv("AS_LDS_TRACE"); trace_countdown = e ? atoi(e) : -1; }
  a.trace = nullptr;
  const size_t trace_bytes = (size_t)TL_MAX_WG * 32 * 8 * 8;
  if (trace_countdown > 0 && --trace_countdown == 0) {
    hipMalloc(&trace_buf, trace_bytes); hipMemset(trace_buf, 0, trace_bytes); a.trace = trace_buf;
  }
#endif
  const int prof_id = mode == 3 ? 6 : 2;
  as_prof_mark(prof_id, st, 1, 0.0);
  void* kargs[] = {&a};
  hipError_t le = hipLaunchKernel(fn, dim3(grid), dim3(256), kargs, TL_LDS_BYTES, st);
  if (le != hipSuccess) { as_set_error("as_conv32_fwd(lds): launch failed: %s", hipGetErrorString(le)); return AS_ERR_LAUNCH; }
#ifdef AS_LDS_TRACE_BUILD
  if (a.trace) {
    hipStreamSynchronize(st);
    void* hbuf = malloc(trace_bytes); hipMemcpy(hbuf, trace_buf, trace_bytes, hipMemcpyDeviceToHost);
    FILE* f = fopen("gpurun_out/lds_trace.bin", "wb"); if (f) { fwrite(hbuf, 1, trace_bytes, f); fclose(f); } free(hbuf);
  }
#endif
  as_prof_mark(prof_id, st, 0, 2.0 * (double)gout->B * gout->H * gout->W * 1024.0 * 9);
  AS_CHECK_LAUNCH("as_conv32_fwd(lds)");
  return AS_OK;
}

// =====================================================================================================
// Weight gradient, same instance:  dW[tap][ci][co] = sum_v X[v + off(tap)][ci] * G[v][co].
// MFMA rows i = ci, columns j = co, reduction k = voxel.  The direct-load kernel (conv32_mfma.hip) issues
// one 256-byte wave load per operand per MFMA (1.33 per MFMA with 3 taps per wave): the CU's vector-memory
// instruction rate, not bytes, limited it to 58 TFLOP/s, and each tap group re-fetched its rows (2.9x the
// algorithmic HBM bytes).  Here a persistent workgroup stages the three X rows and the G row of a
// 128-pixel segment in LDS by DMA, every wave takes 32 of the segment's voxels for ALL NINE taps
// (9 independent accumulators = 144 registers), and the accumulators live in registers across all of the
// workgroup's tiles: one slab per workgroup, summed in fixed order by wgrad_reduce_kernel (deterministic,
// no float atomics).
// The per-tile path carries almost no vector-ALU work (it would cost matrix-core time, see issue_tile_dma):
//   * LDS image = plain copy of the rows (no swizzle): an operand read is a ds_read_b32 of one voxel's 32
//     channels by 32 lanes (+ the next voxel by the other 32) = 64 consecutive banks, conflict-free as is;
//     the address of step s is  base(kx, lane) + row*18432 + s*256  — all immediates;
//   * DMA source = scalar base + lane*16; the ragged end of a row is handled by clamping the SCALAR base of
//     an 8-voxel group: X groups stay inside the padded row (what they then fetch only ever multiplies zero
//     G), G groups at or beyond W read the row's right halo, which is zero (needs pw >= 8).
#define TLG_BYTES (128 * 128)                           // G row segment
#define TLW_LDS_BYTES (TL_BUF_BYTES + TLG_BYTES)        // 71,680 B; two workgroups per CU

struct WgradLdsArgs {
  const float* x;
  const float* gz;
  float* partial;      // [wgs][9][32][32]
  float* partial_db;   // [wgs][32]
  PclDev gin, gout;
  int dil, tiles_per_row, ntiles, tiles_per_band, wg_per_xcd;
  // APPLY flavour: gz is the layer's output gradient g_a, stage 3 of its BatchNorm backward runs on the staged row
  const float* bn_z;       // PCL (output geometry): pre-activation
  const float* bn_scale;   // [32] scale, shift, mean of the forward pass
  const float* bn_shift;
  const float* bn_mean;
  const float* bn_coef;    // [96] k1, k2, k3 of bn_bwd_finalize_kernel
  float* gz_out;           // PCL (output geometry): g_z, written here for the data gradient that follows
  float slope;
};

template <int NW, bool WITH_Z = false>
__device__ inline void issue_wgrad_dma(const WgradLdsArgs& p, int tile, unsigned lds_x, unsigned lds_g, int wave,
                                       unsigned lane16, unsigned lds_z = 0) {
  const int row = tile / p.tiles_per_row;
  const int x0 = (tile - row * p.tiles_per_row) * 128;
  const int b = row / p.gout.H, y = row - b * p.gout.H;
  const int Wp = p.gin.Wp;
  const float* xrow0 = p.x + ((long)b * p.gin.Hp + (y + p.gin.ph - p.dil)) * Wp * 32;      // wave-uniform
  const long row_stride = (long)p.dil * Wp * 32;
  const int px0 = x0 - 8 + p.gin.pw;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float* srow = xrow0 + r * row_stride;
#pragma unroll
    for (int k = 0; k < (TL_DMA_PER_ROW + NW - 1) / NW; ++k) {
      const int i = wave + NW * k;
      if (i < TL_DMA_PER_ROW)
        dma_1kb(srow + (long)min(px0 + 8 * i, Wp - 8) * 32, lane16, lds_x + (unsigned)((r * TL_W + 8 * i) * 128));
    }
  }
  const float* grow = p.gz + (((long)b * p.gout.Hp + (y + p.gout.ph)) * p.gout.Wp + p.gout.pw) * 32;
#pragma unroll
  for (int k = 0; k < 16 / NW; ++k) {
    const int i = wave + NW * k;
    dma_1kb(grow + (long)min(x0 + 8 * i, p.gout.W) * 32, lane16, lds_g + (unsigned)(8 * i * 128));
  }
  if (WITH_Z) {
    const float* zrow = p.bn_z + (((long)b * p.gout.Hp + (y + p.gout.ph)) * p.gout.Wp + p.gout.pw) * 32;
#pragma unroll
    for (int k = 0; k < 16 / NW; ++k) {
      const int i = wave + NW * k;
      dma_1kb(zrow + (long)min(x0 + 8 * i, p.gout.W) * 32, lane16, lds_z + (unsigned)(8 * i * 128));
    }
  }
}

__global__ __launch_bounds__(256, 2) void conv32_wgrad_lds_kernel(WgradLdsArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xbuf = smem;
  char* gbuf = smem + TL_BUF_BYTES;
  const unsigned lds_x = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((lds_ptr_t)xbuf));
  const unsigned lds_g = lds_x + TL_BUF_BYTES;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, li = lane & 31;
  const unsigned lane16 = (unsigned)lane * 16u;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  // operand addresses of step 0 (tile-invariant): voxel 32*wave + h (+8 + (kx-1)*dil in the staged X rows)
  const char* gaddr = gbuf + (32 * wave + h) * 128 + li * 4;
  const char* xaddr[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) xaddr[kx] = xbuf + (32 * wave + h + 8 + (kx - 1) * p.dil) * 128 + li * 4;

  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int t_begin = xcd * p.tiles_per_band;
  const int t_end = min(t_begin + p.tiles_per_band, p.ntiles);
  for (int tile = t_begin + j; tile < t_end; tile += p.wg_per_xcd) {
    issue_wgrad_dma<4>(p, tile, lds_x, lds_g, wave, lane16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // this wave's 32 voxels of the segment: 16 steps of one voxel pair (lane half h picks the voxel)
    float bv[2], av[2][9];
    bv[0] = *reinterpret_cast<const float*>(gaddr);
#pragma unroll
    for (int t = 0; t < 9; ++t) av[0][t] = *reinterpret_cast<const float*>(xaddr[t % 3] + (t / 3) * TL_ROW_BYTES);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (s + 1 < 16) {
        bv[(s + 1) & 1] = *reinterpret_cast<const float*>(gaddr + (s + 1) * 256);
#pragma unroll
        for (int t = 0; t < 9; ++t)
          av[(s + 1) & 1][t] = *reinterpret_cast<const float*>(xaddr[t % 3] + (t / 3) * TL_ROW_BYTES + (s + 1) * 256);
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


// Double-buffered form for launches that fill the chip: ONE 8-wave workgroup per CU, two tile buffers (143 KB of
// LDS), the DMA of tile i+1 in flight while tile i is multiplied, one barrier per tile.  A wave takes 16 of the
// segment's voxels (8 steps of one voxel pair) for all nine taps.
// APPLY: the G row arrives as the layer's OUTPUT gradient g_a together with its pre-activation row z; stage 3 of the
// BatchNorm backward, g_z = (g_a*lrelu'(z*scale+shift) - k1 - (z-mean)*k2)*k3, is applied to the staged row in LDS
// (8 elements per thread) and the result is also written out for the data gradient that follows — the separate
// element-wise pass (read g_a, read z, write g_z: 715 MB per full-resolution layer) disappears.  The z row has a
// single buffer (159,744 B of LDS in all): it is dead once the row is transformed, before the next tile's DMA.
#define TLW2_LDS_BYTES (2 * TLW_LDS_BYTES)
#define TLW2A_LDS_BYTES (2 * TLW_LDS_BYTES + TLG_BYTES)
template <bool APPLY>
__global__ __launch_bounds__(512, 1) void conv32_wgrad_lds2_kernel(WgradLdsArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((lds_ptr_t)smem));
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, li = lane & 31;
  const unsigned lane16 = (unsigned)lane * 16u;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  const char* gaddr0 = smem + TL_BUF_BYTES + (16 * wave + h) * 128 + li * 4;
  const char* xaddr0[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) xaddr0[kx] = smem + (16 * wave + h + 8 + (kx - 1) * p.dil) * 128 + li * 4;

  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int t_begin = xcd * p.tiles_per_band;
  const int t_end = min(t_begin + p.tiles_per_band, p.ntiles);
  const unsigned lds_z = lds0 + TLW2_LDS_BYTES;
  // APPLY: this thread's two float4 of a 128-voxel row (float4 index f = tid, tid + 512: channel group tid & 7)
  f32x4 k1, k2, k3, bsc, bsh, bmu;
  if (APPLY) {
    const int c4 = (threadIdx.x & 7) * 4;
    k1 = *reinterpret_cast<const f32x4*>(p.bn_coef + c4); k2 = *reinterpret_cast<const f32x4*>(p.bn_coef + 32 + c4);
    k3 = *reinterpret_cast<const f32x4*>(p.bn_coef + 64 + c4);
    bsc = *reinterpret_cast<const f32x4*>(p.bn_scale + c4); bsh = *reinterpret_cast<const f32x4*>(p.bn_shift + c4);
    bmu = *reinterpret_cast<const f32x4*>(p.bn_mean + c4);
  }
  int tile = t_begin + j;
  if (tile < t_end) issue_wgrad_dma<8, APPLY>(p, tile, lds0, lds0 + TL_BUF_BYTES, wave, lane16, lds_z);
  int cur = 0;
  for (; tile < t_end; tile += p.wg_per_xcd, cur ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of the current tile
    __syncthreads();                                      // everybody's share landed; everybody is done with the other buffer
    if (APPLY) {
      const int row = tile / p.tiles_per_row;
      const int x0 = (tile - row * p.tiles_per_row) * 128;
      const int b = row / p.gout.H, y = row - b * p.gout.H;
      float* gsm = reinterpret_cast<float*>(smem + cur * TLW_LDS_BYTES + TL_BUF_BYTES);
      const float* zsm = reinterpret_cast<const float*>(smem + TLW2_LDS_BYTES);
      float* orow = p.gz_out + p.gout.vox(b, 0, y, x0) * 32;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int f = threadIdx.x + 512 * q;
        const bool inside = x0 + (f >> 3) < p.gout.W;
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gsm + f * 4);
        const f32x4 zz = *reinterpret_cast<const f32x4*>(zsm + f * 4);
        const f32x4 yy = zz * bsc + bsh;
        f32x4 gy;
        gy.x = yy.x > 0.f ? ga.x : ga.x * p.slope; gy.y = yy.y > 0.f ? ga.y : ga.y * p.slope;
        gy.z = yy.z > 0.f ? ga.z : ga.z * p.slope; gy.w = yy.w > 0.f ? ga.w : ga.w * p.slope;
        f32x4 gzv = (gy - k1 - (zz - bmu) * k2) * k3;
        if (!inside) gzv = (f32x4){0.f, 0.f, 0.f, 0.f};   // beyond W the staged rows are halo: G must vanish there
        *reinterpret_cast<f32x4*>(gsm + f * 4) = gzv;
        if (inside) *reinterpret_cast<f32x4*>(orow + f * 4) = gzv;
      }
      __syncthreads();                                    // the row is g_z now; the z buffer is free
    }
    const int next = tile + p.wg_per_xcd;
    if (next < t_end) {
      const unsigned nb = lds0 + (unsigned)((cur ^ 1) * TLW_LDS_BYTES);
      issue_wgrad_dma<8, APPLY>(p, next, nb, nb + TL_BUF_BYTES, wave, lane16, lds_z);
    }
    const char* gaddr = gaddr0 + cur * TLW_LDS_BYTES;
    const char* xaddr[3] = {xaddr0[0] + cur * TLW_LDS_BYTES, xaddr0[1] + cur * TLW_LDS_BYTES, xaddr0[2] + cur * TLW_LDS_BYTES};
    float bv[2], av[2][9];
    bv[0] = *reinterpret_cast<const float*>(gaddr);
#pragma unroll
    for (int t = 0; t < 9; ++t) av[0][t] = *reinterpret_cast<const float*>(xaddr[t % 3] + (t / 3) * TL_ROW_BYTES);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s + 1 < 8) {
        bv[(s + 1) & 1] = *reinterpret_cast<const float*>(gaddr + (s + 1) * 256);
#pragma unroll
        for (int t = 0; t < 9; ++t)
          av[(s + 1) & 1][t] = *reinterpret_cast<const float*>(xaddr[t % 3] + (t / 3) * TL_ROW_BYTES + (s + 1) * 256);
      }
      __builtin_amdgcn_sched_barrier(0);
      bsum += bv[s & 1];
#pragma unroll
      for (int t = 0; t < 9; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1][t], bv[s & 1], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();

  // reduce the eight waves' accumulators through LDS, three taps per round (fixed order w0 + w1 + .. + w7)
  float* slab = reinterpret_cast<float*>(smem);           // [7 waves][3 taps][16][64] floats = 86,016 B
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
#pragma unroll
          for (int w2 = 0; w2 < 7; ++w2) v += slab[((w2 * 3 + g) * 16 + r) * 64 + lane];
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
  if (threadIdx.x < 32) {
    float sdb = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < 8; ++w2) sdb += dbs[w2 * 32 + li];
    p.partial_db[blockIdx.x * 32 + li] = sdb;
  }
}

static bool wgrad_lds2(const as_pcl* gout) { return lds_ntiles(gout) >= 256 * 12; }   // the launch fills the chip

int conv32_wgrad_lds_slabs(const as_pcl* gout) {
  if (wgrad_lds2(gout)) return 256;                      // one double-buffered 8-wave workgroup per CU
  // every workgroup ends with a 36 KB slab (+ its share of the final reduce): give each at least 16 tiles
  const int nt = lds_ntiles(gout);
  int g = ((nt + 15) / 16 + 7) / 8 * 8;
  if (g > 512) g = 512;
  return g;
}

bool conv32_wgrad_bnapply_ok(const as_pcl* gout) { return wgrad_lds2(gout); }

int conv32_wgrad_lds_launch(const float* x, const as_pcl* gin, const float* gz, const as_pcl* gout,
                            const as_conv_shape* s, float* partial, float* partial_db, const WgradBnApply* bn,
                            void* stream) {
  static AsPerDevice attr_set;
  if (!attr_set.get()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv32_wgrad_lds_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, TLW_LDS_BYTES);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv32_wgrad_lds2_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, TLW2_LDS_BYTES);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv32_wgrad_lds2_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, TLW2A_LDS_BYTES);
    if (e != hipSuccess) {
      as_set_error("conv32_wgrad_lds: cannot reserve %d bytes of LDS: %s", TLW2_LDS_BYTES, hipGetErrorString(e));
      return AS_ERR_LAUNCH;
    }
    attr_set.set();
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
  a.bn_z = nullptr; a.bn_scale = a.bn_shift = a.bn_mean = a.bn_coef = nullptr; a.gz_out = nullptr; a.slope = 0.f;
  if (bn != nullptr) {
    if (!wgrad_lds2(gout)) { as_set_error("as_conv32_wgrad_bnapply: configuration not supported"); return AS_ERR_ARG; }
    a.bn_z = bn->z; a.bn_scale = bn->scale; a.bn_shift = bn->shift; a.bn_mean = bn->mean; a.bn_coef = bn->coef;
    a.gz_out = bn->gz_out; a.slope = bn->slope;
    hipLaunchKernelGGL(conv32_wgrad_lds2_kernel<true>, dim3(grid), dim3(512), TLW2A_LDS_BYTES, (hipStream_t)stream, a);
  } else if (wgrad_lds2(gout))
    hipLaunchKernelGGL(conv32_wgrad_lds2_kernel<false>, dim3(grid), dim3(512), TLW2_LDS_BYTES, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(conv32_wgrad_lds_kernel, dim3(grid), dim3(256), TLW_LDS_BYTES, (hipStream_t)stream, a);
  AS_CHECK_LAUNCH("as_conv32_wgrad(lds)");
  return AS_OK;
}
