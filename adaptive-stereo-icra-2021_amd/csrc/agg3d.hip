// Cost aggregation, second generation: the 3x3x3 stride-1 32->32 convolution of stereo_net.py:21-30,155-161,185-186 (and its
// data gradient) walking DOWN the disparity axis with a rolling window of planes in LDS, with the BatchNorm + LeakyReLU of
// the PREVIOUS layer applied to the operand while it sits in LDS, that BatchNorm's statistics merged from the previous
// launch's per-workgroup partials on the way.
//
// What conv3d_lds.hip left on the table (4 pairs, 12x24x78 per pair): every 128-position tile staged its three input planes
// itself (each plane of the volume staged by three different workgroups: 3.1x the algorithmic bytes), the launch was one
// round of 720 short workgroups whose three exposed DMA waits only overlapped through co-residency, and every layer was
// followed by a finalize launch and a full element-wise BatchNorm+LeakyReLU pass over the volume before the next
// convolution could start (15 launches for a2->a5).
//
// Here a workgroup owns a COLUMN: 128 consecutive positions of the flattened padded plane (conv3d_lds.hip's trick: position
// p = y*Wp + x turns every (kh, kw) tap into the constant offset (kh-1)*Wp + (kw-1)) times a segment of `seg_len` output
// planes.  Four plane runs live in an LDS ring: output plane d reads padded planes d, d+1, d+2 while plane d+3 lands by
// LDS-DMA (global_load_lds_dwordx4, 1 KB per wave instruction) — every plane is staged once per segment
// ((seg_len + 2) / seg_len instead of 3x), its latency hidden behind a whole plane of matrix work, one barrier per plane.
// Weights (27 x 4 KB) stream from L2 eight taps ahead through a nine-deep register ring (27 = 3 * 9: the ring position is a
// compile-time constant), so no wait of the compiler's ever lands near the DMA it cannot see.
//   IN 1: the operand is the previous layer's RAW convolution output z; a = lrelu(z * scale + shift) (that layer's
//         BatchNorm, finalized by its own launch) is applied to each plane once, in LDS, right after it has landed; halo
//         voxels (which must read as the convolution's zero padding, not as lrelu(shift)) are skipped by a per-thread bit
//         mask that is computed once (a workgroup's positions never change).  The activated planes a workgroup owns are
//         written back as a by-product (the backward pass needs them), so the separate element-wise pass disappears.
//   IN 2: as IN 1, but the producing layer's BatchNorm is still in per-workgroup partials: every workgroup merges them
//         (bn_merge.h) while its first three planes are in flight — no finalize launch, and nothing for the producer to
//         wait for.  (Tried first and dropped: the producer's LAST workgroup finalizing behind an agent-scope ticket —
//         write-through partial stores, fan-in on one counter and the merge's fabric loads are three dependent round
//         trips, +10 us on a 45 us launch, more than the finalize launch they replaced.)
//   EPI 0: raw output + per-workgroup BatchNorm moments (per-lane shifted sums, no barrier in the plane loop).
//   EPI 1: lrelu(acc * scale + shift) (eval forward: BatchNorm folded);   EPI 2: raw output, no moments (data gradient).
// Accumulation order per output value = conv3d_lds.hip's (bias, then taps kd-major, 16 fp32 MFMA steps per tap): outputs are
// bit-identical to that kernel's.
#include "as_common.h"
#include "conv_epilogue.h"
#include "agg3d.h"
#include "bn_merge.h"

#ifndef AGG_WRING
#define AGG_WRING 9        // weight register ring: 27 % AGG_WRING == 0; prefetch distance AGG_WRING - 1 taps
#endif

struct Agg3dArgs {
  const float* x;
  const float* wq;           // packed [27][4][64][4] (as_conv32_pack_weights)
  const float* in_scale;     // IN 1: the producing layer's BatchNorm affine
  const float* in_shift;
  float* a_out;              // IN 1: activated input written back (PCL, same geometry), or null
  EpilogueArgs ep;
  BnMergeDev in_bn;          // IN 2: the producing layer's BatchNorm partials
  PclDev g;
  int tiles_per_plane, npos, run, groups, slot_bytes;
  int seg_len, nseg, units, per_xcd;
  unsigned wp_magic;         // ceil(2^32 / Ws): pos / Ws == __umulhi(pos, wp_magic) for pos < 2^16
  // Column strips (planes wider than four LDS runs allow — k = 3: 156 and 120 columns): a plane is cut into `strips`
  // sub-planes of Ws = Cw + 2 columns (Ws a multiple of 8: every 8-voxel DMA group then lies inside one row), each staged
  // and walked exactly like a narrow plane of its own — "position" = y * Ws + x', taps are constant offsets — while global
  // addresses use the full row pitch.  The last strip is right-aligned (it may overlap its neighbour: recomputed, identical
  // values, left out of the moments).  strips == 1: Ws == Wp, the sub-plane IS the plane (the original, contiguous form).
  int strips, Ws, tiles_per_strip;
};

typedef __attribute__((address_space(3))) void* lds_as3_t;

__device__ inline void agg_dma_1kb(const float* sbase, unsigned voff, unsigned m0) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               :: "s"(m0), "v"(voff), "s"(sbase) : "memory", "m0");
}
template <int IMM> __device__ inline void agg_store_imm(float* sbase, unsigned voff, float v) {
  asm volatile("global_store_dword %0, %1, %2 offset:%3" :: "v"(voff), "v"(v), "s"(sbase), "n"(IMM) : "memory");
}
#define AGG_ROW_IMM(r) ((((r) & 3) + 8 * ((r) >> 2)) * 128)
#define AGG_FOR_ROWS(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

__device__ inline f32x4 agg_chunk(const char* slot, int v, int h, int q) {
  return *reinterpret_cast<const f32x4*>(slot + v * 128 + (((4 * h + q) ^ ((v >> 1) & 7)) << 4));
}
// One tap's B fragments through a buffer descriptor: 32-bit lane offset + scalar tap offset + immediates — no per-tap
// 64-bit address registers (hipcc otherwise keeps 27 address pairs alive across the plane loop).
typedef unsigned agg_u32x4 __attribute__((ext_vector_type(4)));
__device__ inline void agg_load_w(f32x4 (&r)[4], __amdgpu_buffer_rsrc_t rsrc, unsigned lane_bytes, int tap) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const agg_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_bytes + q * 1024, tap * 4096, 0);
    r[q] = __builtin_bit_cast(f32x4, v);
  }
}

__device__ inline int agg_div_wp(int pos, unsigned magic) { return (int)__umulhi((unsigned)pos, magic); }

// One plane run (padded plane q of pair b, positions [pos0 - Wp - 1, pos0 + 128 + Wp + 1)) into ring slot q & 3.
__device__ inline void agg_issue_plane(const Agg3dArgs& p, int b, int q, int pos0, long plane_vox, unsigned lds0, int wave,
                                       unsigned off_reg, unsigned off_tail, unsigned tail_v0) {
  const float* src = p.x + (((long)b * p.g.Dp + q) * plane_vox + (pos0 - p.g.Wp - 1)) * 32;
  const unsigned slot = lds0 + (unsigned)((q & 3) * p.slot_bytes);
  for (int i = wave; i < p.groups - 1; i += 4)
    agg_dma_1kb(src + i * 256, off_reg, slot + (unsigned)(i * 1024));
  if (((p.groups - 1) & 3) == wave)
    agg_dma_1kb(src + (long)tail_v0 * 32, off_tail, slot + tail_v0 * 128u);
}

// Strip form: the run [rs_al, rs_al + 8 ng) in strip coordinates, group by group (a group = 8 positions of one row).
__device__ inline void agg_issue_plane_strip(const Agg3dArgs& p, int b, int q, int rs_al, int ng, int xs, unsigned lds0, int wave,
                                             unsigned off_reg) {
  const float* plane = p.x + ((long)b * p.g.Dp + q) * ((long)p.g.Wp * p.g.Hp) * 32;
  const unsigned slot = lds0 + (unsigned)((q & 3) * p.slot_bytes);
  for (int i = wave; i < ng; i += 4) {
    const int pos = rs_al + 8 * i;
    const int y = (int)__umulhi((unsigned)pos, p.wp_magic), xq = pos - y * p.Ws;
    agg_dma_1kb(plane + ((long)y * p.g.Wp + xs + xq) * 32, off_reg, slot + (unsigned)(i * 1024));
  }
}

template <int IN, int EPI, bool STRIP>
__global__ __launch_bounds__(256, 1) void agg3d_kernel(Agg3dArgs p) {
  extern __shared__ __attribute__((aligned(16))) char ring[];
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((lds_as3_t)ring));
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, li = lane & 31;
  const int Wp = STRIP ? p.Ws : p.g.Wp;          // row pitch of the staged (sub-)plane: every tap offset is in these units

  // unit = (pair b, tile t of the plane, segment of output planes); the segments of one column sit on one XCD
  const int unit = (blockIdx.x & 7) * p.per_xcd + (blockIdx.x >> 3);
  const bool active = unit < p.units;                                   // workgroup-uniform
  const int col = active ? unit / p.nseg : 0, seg = active ? unit - col * p.nseg : 0;
  const int b = col / p.tiles_per_plane;
  int t = col - b * p.tiles_per_plane;
  const int strip = STRIP ? t / p.tiles_per_strip : 0;
  if (STRIP) t -= strip * p.tiles_per_strip;
  const int xs = STRIP ? min(strip * (p.Ws - 2), p.g.Wp - p.Ws) : 0;    // padded global column of strip column 0
  const int dupc = STRIP ? strip * (p.Ws - 2) - xs : 0;                 // interior strip columns <= dupc belong to the strip before
  const int d0 = seg * p.seg_len, d1 = active ? min(d0 + p.seg_len, p.g.D) : d0;
  const int first = p.g.ph * Wp + p.g.pw;
  const int pos_new = first + 128 * t;
  const int pos0 = min(pos_new, first + p.npos - 128);
  const long plane_vox = (long)p.g.Wp * p.g.Hp;
  // strip form: the run starts at a multiple of 8 (lead = what that adds in front) and is ng groups long
  const int lead = STRIP ? ((pos0 - Wp - 1) & 7) : 0;
  const int rs_al = pos0 - Wp - 1 - lead;
  const int ng = STRIP ? (lead + 130 + 2 * Wp + 7) >> 3 : p.groups;

  // DMA lane constants (conv3d_lds.hip): source-side swizzle, the last group placed to END at the run's end
  const unsigned vl = (unsigned)(lane >> 3), sl = (unsigned)(lane & 7);
  const unsigned off_reg = vl * 128u + ((sl ^ ((((unsigned)wave & 1u) << 2) | (vl >> 1))) << 4);
  const unsigned tail_v0 = (unsigned)(p.run - 8);
  const unsigned off_tail = vl * 128u + ((sl ^ ((((tail_v0 + vl) >> 1)) & 7u)) << 4);

  // (the prologue's DMA is in flight while the lane masks below are computed)
  auto issue = [&](int q) {
    if (STRIP) agg_issue_plane_strip(p, b, q, rs_al, ng, xs, lds0, wave, off_reg);
    else agg_issue_plane(p, b, q, pos0, plane_vox, lds0, wave, off_reg, off_tail, tail_v0);
  };
  if (active)
    for (int q = d0; q < d0 + 3; ++q) issue(q);

  // this lane's 16 output rows: position pos0 + 32*wave + row(r, h); halo columns are stored as zeros (they must stay
  // zero), duplicates of the previous tile (shifted last tile) are stored again with identical values, both are left out
  // of the moments.
  unsigned keep = 0u, fresh = 0u;                                        // bit r: interior column / first written here
  unsigned so[STRIP ? 16 : 1];                                           // strip form: byte offset of row r's voxel in its plane
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int pos = pos0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
    const int yp = agg_div_wp(pos, p.wp_magic);
    const int xp = pos - yp * Wp;
    bool in = xp >= p.g.pw && xp < p.g.pw + p.g.W;
    bool fr = in && pos >= pos_new;
    if (STRIP) {
      const int gx = xs + xp;                                            // padded global column
      in = xp >= 1 && xp <= Wp - 2 && gx < p.g.pw + p.g.W;
      fr = in && pos >= pos_new && xp > dupc;
      // rows that are not kept are stored (as zeros) onto voxel 0 of the plane — a halo voxel, zero for life
      so[r] = (in ? (unsigned)(yp * p.g.Wp + gx) * 128u : 0u) + 4u * (unsigned)li;
    }
    keep |= (in ? 1u : 0u) << r;
    fresh |= (fr ? 1u : 0u) << r;
  }
  const unsigned io_off = (unsigned)(512 * h + 4 * li);

  // IN 1: element-wise pass over a landed plane.  Thread (cg = tid & 7, voxel row = tid >> 3) owns the 16-byte chunk of
  // channels 4cg..4cg+3 of voxels (tid >> 3) + 32k; tmask bit k = that voxel is an interior voxel of its plane (and
  // exists); omask bit k = it is one of the column's own 128 positions (written back to a_out).
  f32x4 in_sc = {0.f, 0.f, 0.f, 0.f}, in_sh = {0.f, 0.f, 0.f, 0.f};
  unsigned tmask = 0u, omask = 0u;
  const int cg = threadIdx.x & 7, vrow = threadIdx.x >> 3;
  if (IN == 1) {
    in_sc = *reinterpret_cast<const f32x4*>(p.in_scale + 4 * cg);
    in_sh = *reinterpret_cast<const f32x4*>(p.in_shift + 4 * cg);
  }
  if (IN == 2) {            // while the first three planes are in flight
#ifdef AGG_EXP_NOMERGE      // diagnostic builds (tests/tools/exp_step.sh): results are wrong, only the time means something
    in_sc = (f32x4){1.f, 1.f, 1.f, 1.f};
#else
    const float* tab = bn_merge_partials<32>(p.in_bn, ring + ((d0 + 3) & 3) * p.slot_bytes, blockIdx.x == 0);   // plane d0+3's slot: no DMA target yet
    in_sc = *reinterpret_cast<const f32x4*>(tab + 4 * cg);
    in_sh = *reinterpret_cast<const f32x4*>(tab + 32 + 4 * cg);
    __syncthreads();        // every thread has its affine before plane d0+3 is requested into the scratch's slot
#endif
  }
  unsigned ao[STRIP ? 12 : 1];                       // strip form: byte offset of chunk k's voxel in its plane (by-product)
  if (IN != 0) {
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const int v = vrow + 32 * k;
      const int pos = rs_al + v;
      const int yp = agg_div_wp(max(pos, 0), p.wp_magic), xp = pos - yp * Wp;
      bool in = v < p.run && xp >= p.g.pw && xp < p.g.pw + p.g.W && yp >= p.g.ph && yp < p.g.ph + p.g.H;
      bool own = in && v >= Wp + 1 && v < Wp + 129;
      if (STRIP) {
        const int gx = xs + xp;
        // a strip's halo columns are real voxels of its neighbours: activated like any other (they feed this strip's taps)
        in = v < 8 * ng && gx >= p.g.pw && gx < p.g.pw + p.g.W && yp >= p.g.ph && yp < p.g.ph + p.g.H;
        own = in && v >= lead + Wp + 1 && v < lead + Wp + 129 && xp >= 1 && xp <= Wp - 2;
        ao[k] = (unsigned)(yp * p.g.Wp + gx) * 128u;
      }
      tmask |= (in ? 1u : 0u) << k;
      omask |= (own ? 1u : 0u) << k;
    }
  }
  // Branch-free: all 12 chunk reads of a thread are issued together (a per-k branch would expose one LDS round trip per
  // chunk); chunks that must not change (halo voxels, voxels beyond the run) are redirected to a 16-byte dump slot the
  // thread owns, so the twelve writes are unconditional too.
  char* dump = ring + 4 * p.slot_bytes + threadIdx.x * 16;
  auto activate = [&](int q) {                       // padded plane q, resident in slot q & 3
#ifdef AGG_EXP_NOACT
    return;
#endif
    if (q < p.g.pd || q >= p.g.pd + p.g.D) return;   // a halo plane: zeros stay zeros
    char* slot = ring + (q & 3) * p.slot_bytes;
    const bool own = p.a_out != nullptr && q - p.g.pd >= d0 && q - p.g.pd < d1;
    float* outp = STRIP ? p.a_out + (((long)b * p.g.Dp + q) * plane_vox) * 32 + 4 * cg
                        : p.a_out + (((long)b * p.g.Dp + q) * plane_vox + (pos0 - Wp - 1)) * 32 + 4 * cg;
    f32x4* cp[12];
    f32x4 y[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const int v = vrow + 32 * k;
      char* c = slot + v * 128 + ((cg ^ ((v >> 1) & 7)) << 4);
      cp[k] = reinterpret_cast<f32x4*>(((tmask >> k) & 1u) ? c : dump);
      y[k] = *cp[k];
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      y[k].x = fmaf(y[k].x, in_sc.x, in_sh.x); y[k].y = fmaf(y[k].y, in_sc.y, in_sh.y);
      y[k].z = fmaf(y[k].z, in_sc.z, in_sh.z); y[k].w = fmaf(y[k].w, in_sc.w, in_sh.w);
      // LeakyReLU with 0 < slope < 1 is max(y, slope*y): the same bits as the select form, one instruction less
      y[k].x = fmaxf(y[k].x, y[k].x * p.ep.slope); y[k].y = fmaxf(y[k].y, y[k].y * p.ep.slope);
      y[k].z = fmaxf(y[k].z, y[k].z * p.ep.slope); y[k].w = fmaxf(y[k].w, y[k].w * p.ep.slope);
      *cp[k] = y[k];
    }
    if (own) {
#pragma unroll
      for (int k = 0; k < 12; ++k)
        if ((omask >> k) & 1u) {
          if (STRIP) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(outp) + ao[k]) = y[k];
          else *reinterpret_cast<f32x4*>(outp + (long)(vrow + 32 * k) * 32) = y[k];
        }
    }
  };

  const unsigned wlane = (unsigned)lane * 16u;
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wq), 0, 27 * 4096, 0x00020000);
  const float bias_v = p.ep.bias ? p.ep.bias[li] : 0.f;
  float ep_sc = 1.f, ep_sh = 0.f;
  if (EPI == 1) { ep_sc = p.ep.ep_scale[li]; ep_sh = p.ep.ep_shift[li]; }
  float st_n = 0.f, st_c = 0.f, st_s1 = 0.f, st_s2 = 0.f;

  if (active) {
    // ---- prologue: the first three planes (requested above), then the fourth on its way ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (IN != 0) {
      for (int q = d0; q < d0 + 3; ++q) activate(q);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the by-product stores precede the next DMA in the queue
      __syncthreads();
    }
    if (d0 + 1 < d1) issue(d0 + 3);

    f32x4 bw[AGG_WRING][4], a[2][4];
#pragma unroll
    for (int tp = 0; tp < AGG_WRING - 1; ++tp) agg_load_w(bw[tp], wrsrc, wlane, tp);
    const int vbase = lead + 32 * wave + li;

    for (int d = d0; d < d1; ++d) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = bias_v;
      const char* s0 = ring + ((d + 0) & 3) * p.slot_bytes;
      const char* s1 = ring + ((d + 1) & 3) * p.slot_bytes;
      const char* s2 = ring + ((d + 2) & 3) * p.slot_bytes;
#pragma unroll
      for (int q = 0; q < 4; ++q) a[0][q] = agg_chunk(s0, vbase, h, q);
#pragma unroll
      for (int tap = 0; tap < 27; ++tap) {
        agg_load_w(bw[(tap + AGG_WRING - 1) % AGG_WRING], wrsrc, wlane, (tap + AGG_WRING - 1) % 27);   // AGG_WRING - 1 taps ahead (wraps into the next plane)
        if (tap + 1 < 27) {
          const int kd = (tap + 1) / 9, tp = (tap + 1) % 9;
          const char* sb = kd == 0 ? s0 : (kd == 1 ? s1 : s2);
          const int v = vbase + (tp / 3) * Wp + (tp % 3);
#pragma unroll
          for (int q = 0; q < 4; ++q) a[(tap + 1) & 1][q] = agg_chunk(sb, v, h, q);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 av = a[tap & 1][q], bv = bw[tap % AGG_WRING][q];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }

      if (d + 1 < d1) {
        // Plane d+3 was requested a whole plane of matrix work ago and every weight load since (the compiler waits for
        // each of them, in order) is younger than it: it has landed.  The guard below allows the 32 weight loads of the
        // next plane's first eight taps to stay in flight.
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (AGG_WRING - 1)) : "memory");
        __syncthreads();                               // every wave is done with plane d's slot; plane d+3 is visible
        if (IN != 0) {
          activate(d + 3);
          __syncthreads();
        }
        if (d + 2 < d1) issue(d + 4);
      }
      __builtin_amdgcn_sched_barrier(0);

      // ---- epilogue of plane d: 16 stores per lane, scalar base + lane offset + immediate ----
      float* z_base = STRIP ? p.ep.z + (((long)b * p.g.Dp + d + p.g.pd) * plane_vox) * 32
                            : p.ep.z + ((((long)b * p.g.Dp + d + p.g.pd) * plane_vox) + pos0 + 32 * wave) * 32;
#define AGG_ST(r) { float v = acc[r];                                                       \
                    if (EPI == 1) { v = fmaf(v, ep_sc, ep_sh); v = v > 0.f ? v : v * p.ep.slope; } \
                    v = ((keep >> r) & 1u) ? v : 0.f;                                        \
                    if (STRIP) agg_store_imm<0>(z_base, so[STRIP ? r : 0], v);               \
                    else agg_store_imm<AGG_ROW_IMM(r)>(z_base, io_off, v); }
      AGG_FOR_ROWS(AGG_ST)
#undef AGG_ST
      if (EPI == 0 && p.ep.stat_mean != nullptr) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool f = (fresh >> r) & 1u;
          st_c = (f && st_n == 0.f) ? acc[r] : st_c;          // shift = the lane's first counted element
          const float dlt = f ? acc[r] - st_c : 0.f;
          st_s1 += dlt; st_s2 = fmaf(dlt, dlt, st_s2); st_n += f ? 1.f : 0.f;
        }
      }
    }
  }

  if (EPI == 0 && p.ep.stat_mean != nullptr) {
    // lane sums -> (n, mean, M2) -> one partial per workgroup, 8 (wave, half) partials per channel merged in fixed order
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* part = reinterpret_cast<float*>(ring);      // [8][32][3]
    const float mean_l = st_n > 0.f ? st_c + st_s1 / st_n : 0.f;
    const float m2_l = st_n > 0.f ? fmaxf(st_s2 - st_s1 * st_s1 / st_n, 0.f) : 0.f;
    float* mine = part + ((wave * 2 + h) * 32 + li) * 3;
    mine[0] = st_n; mine[1] = mean_l; mine[2] = m2_l;
    __syncthreads();
    if (threadIdx.x < 32) {
      TileStats run; run.n = 0.f; run.mean = 0.f; run.m2 = 0.f;
      for (int q = 0; q < 8; ++q) {
        TileStats ts;
        ts.n = part[(q * 32 + li) * 3]; ts.mean = part[(q * 32 + li) * 3 + 1]; ts.m2 = part[(q * 32 + li) * 3 + 2];
        stats_merge(run, ts);
      }
      stats_write(p.ep, blockIdx.x, run);
    }
  }
}

// ---- host ------------------------------------------------------------------------------------------------------------
#define AGG_MAX_WP 83                  // widest (sub-)plane whose four runs fit in LDS: (130 + 2 * 83 + 7) / 8 = 37 groups
static int agg_strips(const as_pcl* g) { return g->W + 2 <= AGG_MAX_WP ? 1 : as_div_up(g->W, 78); }
static int agg_ws(const as_pcl* g) {       // row pitch of the staged (sub-)plane
  const int n = agg_strips(g);
  if (n == 1) return g->W + 2;
  return (as_div_up(g->W, n) + 2 + 7) / 8 * 8;
}
static int agg_run(const as_pcl* g) { return 130 + 2 * agg_ws(g) + (agg_strips(g) > 1 ? 7 : 0); }   // (strips: + alignment lead)
static int agg_tiles_per_strip(const as_pcl* g) {
  const int Ws = agg_ws(g), cw = agg_strips(g) == 1 ? g->W : Ws - 2;
  return as_div_up((int64_t)(g->H - 1) * Ws + cw, 128);
}
static int agg_tiles_per_plane(const as_pcl* g) { return agg_strips(g) * agg_tiles_per_strip(g); }

bool agg3d_applicable(const as_pcl* g) {
  if (!as_pcl_ok(g) || g->pd != 1 || g->ph != 1 || g->pw != 1) return false;
  const int Ws = agg_ws(g), cw = agg_strips(g) == 1 ? g->W : Ws - 2;
  if (Ws < 34 || Ws > AGG_MAX_WP || Ws > g->W + 2) return false;  // a wave tile (32 positions) spans at most two rows
  if ((long)(g->H - 1) * Ws + cw < 128) return false;             // at least one full tile per (sub-)plane
  if ((long)(g->H + 2) * Ws >= 65536) return false;               // positions are divided by Ws through a 16-bit magic multiply
  if (agg_run(g) > 32 * 12) return false;                         // the element-wise pass covers 12 x 32 voxels per plane
  return (long)((agg_run(g) + 7) / 8) * 1024 * 4 + 4096 <= 156 * 1024;   // four plane runs + the element-wise pass's dump slots
}

// Segment length: the launch should be as few rounds of 256 single-workgroup CUs as possible; a unit costs its planes plus
// about one plane's worth of prologue.
static int agg_seg_len(const as_pcl* g) {
  const int cols = g->B * agg_tiles_per_plane(g);
  int best = 1; double best_cost = 1e30;
  for (int L = 1; L <= g->D; ++L) {
    const int nseg = as_div_up(g->D, L);
    const long units = (long)cols * nseg;
    const long rounds = (units + 255) / 256;
    const double cost = (double)rounds * (L + 1.0);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = L; }
  }
  return best;
}

int agg3d_units(const as_pcl* g) {
  const int L = agg_seg_len(g);
  const int units = g->B * agg_tiles_per_plane(g) * as_div_up(g->D, L);
  return as_div_up(units, 8) * 8;                                  // = the grid = the number of BatchNorm partials
}

template <int IN, int EPI, bool STRIP>
static int agg_launch_s(const Agg3dArgs& a, int lds_bytes, hipStream_t st) {
  static AsPerDevice attr_set;
  if (!attr_set.get()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(agg3d_kernel<IN, EPI, STRIP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { as_set_error("as_agg3d_fwd: %s", hipGetErrorString(e)); return AS_ERR_LAUNCH; }
    attr_set.set();
  }
  hipLaunchKernelGGL((agg3d_kernel<IN, EPI, STRIP>), dim3(8 * a.per_xcd), dim3(256), lds_bytes, st, a);
  return AS_OK;
}
template <int IN, int EPI>
static int agg_launch_t(const Agg3dArgs& a, int lds_bytes, hipStream_t st) {
  return a.strips > 1 ? agg_launch_s<IN, EPI, true>(a, lds_bytes, st) : agg_launch_s<IN, EPI, false>(a, lds_bytes, st);
}

extern "C" int as_agg3d_ok(const as_pcl* g) { return agg3d_applicable(g) ? 1 : 0; }
extern "C" int as_agg3d_parts(const as_pcl* g) { return agg3d_applicable(g) ? agg3d_units(g) : AS_ERR_ARG; }

extern "C" int as_agg3d_fwd(const float* x, const as_pcl* g, const float* packed_w, const float* bias,
                            const float* in_scale, const float* in_shift, const as_bn_merge* in_bn, float* a_out,
                            float* z, int epilogue, const float* ep_scale, const float* ep_shift, float slope,
                            float* stat_mean, float* stat_m2, float* stat_cnt, void* stream) {
  AS_CHECK_ARG(agg3d_applicable(g), "as_agg3d_fwd: geometry not supported (as_agg3d_ok() == 0)");
  AS_CHECK_ARG(x && packed_w && z, "as_agg3d_fwd: null pointer");
  AS_CHECK_ARG(epilogue >= 0 && epilogue <= 2, "as_agg3d_fwd: epilogue must be 0 (raw + moments), 1 (affine + LeakyReLU) or 2 (raw)");
  AS_CHECK_ARG(epilogue != 1 || (ep_scale && ep_shift), "as_agg3d_fwd: epilogue 1 needs scale and shift");
  AS_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "as_agg3d_fwd: in_scale and in_shift must pair");
  AS_CHECK_ARG(!(in_scale && in_bn), "as_agg3d_fwd: pass the input BatchNorm either as an affine or as partials, not both");
  AS_CHECK_ARG(a_out == nullptr || in_scale != nullptr || in_bn != nullptr, "as_agg3d_fwd: a_out needs the input BatchNorm");
  AS_CHECK_ARG((in_scale == nullptr && in_bn == nullptr) || epilogue == 0, "as_agg3d_fwd: an input BatchNorm goes with epilogue 0");
  AS_CHECK_ARG((stat_mean == nullptr) == (stat_m2 == nullptr) && (stat_mean == nullptr) == (stat_cnt == nullptr),
               "as_agg3d_fwd: the three moment buffers must pair");
  AS_CHECK_ARG(a_out == nullptr || a_out != x, "as_agg3d_fwd: a_out must not alias the input");
  Agg3dArgs a;
  a.x = x; a.wq = packed_w; a.in_scale = in_scale; a.in_shift = in_shift; a.a_out = a_out;
  if (in_bn) AS_CHECK_ARG(bn_merge_fill(&a.in_bn, in_bn), "as_agg3d_fwd: incomplete as_bn_merge block");
  a.ep.bias = bias; a.ep.z = z; a.ep.ep_scale = ep_scale; a.ep.ep_shift = ep_shift; a.ep.residual = nullptr;
  const bool moments = epilogue == 0 && stat_mean != nullptr;
  a.ep.stat_mean = moments ? stat_mean : nullptr; a.ep.stat_m2 = moments ? stat_m2 : nullptr;
  a.ep.stat_cnt = moments ? stat_cnt : nullptr;
  a.ep.epilogue = epilogue; a.ep.slope = slope;
  a.g = as_make_dev(g);
  a.strips = agg_strips(g); a.Ws = agg_ws(g); a.tiles_per_strip = agg_tiles_per_strip(g);
  a.tiles_per_plane = agg_tiles_per_plane(g);
  a.npos = (g->H - 1) * a.Ws + (a.strips == 1 ? g->W : a.Ws - 2);
  a.run = agg_run(g);
  a.groups = (a.run + 7) / 8;
  a.slot_bytes = a.groups * 1024;
  a.seg_len = agg_seg_len(g);
  a.nseg = as_div_up(g->D, a.seg_len);
  a.units = g->B * a.tiles_per_plane * a.nseg;
  a.per_xcd = as_div_up(a.units, 8);
  a.wp_magic = (unsigned)((((uint64_t)1 << 32) + a.Ws - 1) / a.Ws);
  const int lds_bytes = 4 * a.slot_bytes + 4096;
  hipStream_t st = (hipStream_t)stream;
  const double flops = 2.0 * (double)g->B * g->D * g->H * g->W * 1024.0 * 27.0;
  as_prof_mark(7, st, 1, 0.0);
  int e;
  if (in_bn) e = agg_launch_t<2, 0>(a, lds_bytes, st);
  else if (in_scale) e = agg_launch_t<1, 0>(a, lds_bytes, st);
  else if (epilogue == 1) e = agg_launch_t<0, 1>(a, lds_bytes, st);
  else if (moments) e = agg_launch_t<0, 0>(a, lds_bytes, st);
  else e = agg_launch_t<0, 2>(a, lds_bytes, st);
  if (e) return e;
  as_prof_mark(7, st, 0, flops);
  AS_CHECK_LAUNCH("as_agg3d_fwd");
  return AS_OK;
}
