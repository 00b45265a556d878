// Device helpers shared by the minimal-filtering kernels of the full-resolution layers (conv32_wino.hip: forward, inference
// block, first-generation data gradient; conv32_wino_dgrad.hip: the role-specialised data gradient): hand-waited global
// loads / stores (hidden from hipcc's counter bookkeeping, retired by explicit s_waitcnt — tests/tools/check_async_loads.py
// scans the ISA for reads of in-flight destinations) and the swizzled LDS row image the tile gather reads without conflicts.
#pragma once
#include "as_common.h"

__device__ inline void wn_load4(f32x4& v, const float* sbase, unsigned voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
}
__device__ inline void wn_store4(float* sbase, unsigned voff, const f32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" :: "v"(voff), "v"(v), "s"(sbase) : "memory");
}
template <int IMM> __device__ inline void wn_load_imm(float& v, const float* sbase, unsigned voff) {
  asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}
template <int IMM> __device__ inline void wn_store_imm(float* sbase, unsigned voff, float v) {
  asm volatile("global_store_dword %0, %1, %2 offset:%3" :: "v"(voff), "v"(v), "s"(sbase), "n"(IMM) : "memory");
}

// first column (relative to the segment) of tile t at dilation 2^L: blocks of 2d columns hold d tiles
template <int L> __host__ __device__ constexpr int wn_c0(int t) { return ((t >> L) << (L + 1)) | (t & ((1 << L) - 1)); }
// LDS position and swizzle of staged voxel v (0..79)
template <int L> __device__ inline int wn_pos(int v) { return L == 0 ? ((v & ~3) | ((v & 1) << 1) | ((v >> 1) & 1)) : v; }
template <int L> __device__ inline int wn_swz(int v) {
  const int w = v + 8;
  const int key = ((w >> (L + 1)) << L) | (w & ((1 << L) - 1));
  return (key >> 1) & 7;
}
template <int L> __device__ inline int wn_addr(int v, int chunk) { return wn_pos<L>(v) * 128 + ((chunk ^ wn_swz<L>(v)) << 4); }


// ---- row-pair addressing of the conversions ----
// A conversion team of NT threads (256: all four waves of conv32_wino.hip; 128: the two inner waves of conv32_wino_dgrad.hip)
// turns a PAIR of rows, d + 64 + d voxels each, into LDS ring rows: chunk f = t + NT k of the pair's 2 * RC 16-byte chunks
// (RC = 8 NV per row), f < RC row A, else row B.  Everything a chunk's addresses depend on is linear in t, so a thread keeps
// 1 + 2 PER loop-invariant registers — its byte offset t * 16 and the LDS offsets of its first row-A / row-B chunks per swizzle
// phase (the swizzle repeats every 32 voxels) — and reaches chunk k through compile-time immediates and scalar base adds.
// (Recomputing voxel, position, swizzle and offsets per chunk and tile was ~25 integer instructions per chunk: with the
// vector ALU adding to the matrix time on a SIMD, a fifth of the forward kernel's run time.)
// a - b on two packed instructions.  hipcc (ROCm 7.2) turns a vector ADD into v_pk_add_f32 but a vector SUBTRACTION — also
// a + (-b) and fma(b, -1, a) — into one v_sub_f32 per element; the minimal-filtering transforms are signed sums, half of them
// differences, and every vector instruction of a wave costs the SIMD's matrix pipe ~5 cycles.  The neg
// modifiers make it the same IEEE operation: bit-identical results.
// HAZARDS.  hipcc pads its own instructions for the matrix pipe's data hazards; it cannot see into inline asm (the trap of
// DESIGN 4, "a second trap of the same kind").  Two rules for every use of these helpers, both enforced at the call sites:
//   * operands that are MFMA RESULTS: wn_after_mfma() first (a 16-pass MFMA's destination may be read by a vector
//     instruction 18 wait states later at the earliest) — the first version read accumulators straight away: two launches of
//     the same kernel differed (test_conv32_forward_by_minimal_filtering);
//   * results that FEED an MFMA: wn_before_mfma() behind the last of them — two wait states are required (measured:
//     tests/tools/scratch/pk_to_mfma_hazard.hip; 0 and 1 give wrong sums on every run, 2 and more never); the first build
//     (plain asm, no guard) had the last transform of a tile one instruction in front of the MFMA that read it in the MODE 0
//     instantiations: two launches of the same kernel differed (test_conv32_forward_by_minimal_filtering[4-97-700-1-False]).
// The asm statements are volatile: they keep their order relative to those two.  tests/tools/check_async_loads.py checks both
// rules on the ISA of every kernel (and its self-test builds with -DWN_TEST_NO_HAZARD_GUARD, which must be flagged).
typedef float wn_f32x2 __attribute__((ext_vector_type(2)));
#ifdef WN_TEST_NO_HAZARD_GUARD     // (only for tests/tools/check_async_loads.py's self-test: the scan must flag this build)
__device__ __forceinline__ void wn_after_mfma() {}
#else
__device__ __forceinline__ void wn_after_mfma() { asm volatile("s_nop 15\n\ts_nop 3" ::: "memory"); }
#endif
#ifdef WN_TEST_NO_HAZARD_GUARD
__device__ __forceinline__ void wn_before_mfma() {}
#else
__device__ __forceinline__ void wn_before_mfma() { asm volatile("s_nop 7" ::: "memory"); }
#endif
__device__ __forceinline__ wn_f32x2 wn_sub2(wn_f32x2 a, wn_f32x2 b) {
  wn_f32x2 r;
#ifdef WN_TEST_NO_HAZARD_GUARD     // (the first build's form, for the scanner's self-test)
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
#else
  asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
#endif
  return r;
}
__device__ __forceinline__ f32x4 wn_sub4(f32x4 a, f32x4 b) {
#ifdef WN_NO_PK_SUB
  return a - b;
#else
  const wn_f32x2 lo = wn_sub2((wn_f32x2){a.x, a.y}, (wn_f32x2){b.x, b.y}), hi = wn_sub2((wn_f32x2){a.z, a.w}, (wn_f32x2){b.z, b.w});
  return (f32x4){lo.x, lo.y, hi.x, hi.y};
#endif
}
template <bool PACKED> __device__ __forceinline__ f32x4 wn_sub4_if(f32x4 a, f32x4 b) {
  if constexpr (PACKED) return wn_sub4(a, b);
  else return a - b;
}

template <int NT, int L> struct WnPair {
  static constexpr int d = 1 << L, NV = 64 + 2 * d, V0 = 8 - d, RC = 8 * NV;
  static constexpr int K = (2 * RC + NT - 1) / NT;        // chunks per thread and pair
  static constexpr int STEP = NT / 8;                      // voxels from a thread's chunk k to its chunk k + 1
  static constexpr int PER = 32 / STEP;                    // chunks per swizzle period: 1 or 2
  static constexpr int KS = RC / NT;                       // the chunk index that straddles the two rows (if any)
  static constexpr bool STRADDLE = (RC % NT) != 0;
  static constexpr int TS = RC - NT * KS;                  // ... threads t >= TS of it are in row B
  static constexpr int CL = 2 * RC - 1 - NT * (K - 1);     // thread whose last chunk is the pair's last chunk
  __host__ __device__ static constexpr bool all_a(int k) { return NT * (k + 1) <= RC; }
  __host__ __device__ static constexpr bool all_b(int k) { return NT * k >= RC; }
  __host__ __device__ static constexpr bool full(int k) { return NT * (k + 1) <= 2 * RC; }
  __host__ __device__ static constexpr int kb(int ph) { int k = RC / NT; while (k % PER != ph) ++k; return k; }
  // can chunk k hold halo voxels (row-relative voxel vq < d or vq >= 64 + d)?  vq spans [STEP k - (B ? NV : 0), + STEP)
  __host__ __device__ static constexpr bool may_halo(int k, bool rowb) {
    const int lo = STEP * k - (rowb ? NV : 0), hi = lo + STEP - 1;
    return (lo < d && hi >= 0) || (hi >= 64 + d && lo < NV);
  }
  int t;                                                   // thread of the team
  unsigned t16;                                            // t * 16
  int la[PER], lb[PER];                                    // LDS offsets in a ring row: chunk (k = ph) of row A, chunk kb(ph) of row B
  __device__ inline void init(int t_, int lds_bias) {
    t = t_; t16 = (unsigned)t_ * 16u;
#pragma unroll
    for (int ph = 0; ph < PER; ++ph) {
      la[ph] = wn_addr<L>(V0 + (t_ >> 3) + STEP * ph, t_ & 7) - lds_bias;
      lb[ph] = wn_addr<L>(V0 + (t_ >> 3) + STEP * kb(ph) - NV, t_ & 7) - lds_bias;
    }
  }
  __device__ inline int lds_a(int k) const { return la[k % PER] + (k - k % PER) * STEP * 128; }
  __device__ inline int lds_b(int k) const { return lb[k % PER] + (k - kb(k % PER)) * STEP * 128; }
  __device__ inline bool in_b(int k) const { return all_b(k) || (!all_a(k) && t >= TS); }       // per lane only for k == KS
  __device__ inline bool active(int k) const { return full(k) || t <= CL; }
  __device__ inline int vq(int k, bool rowb) const { return (t >> 3) + STEP * k - (rowb ? NV : 0); }   // staged voxel 0..NV-1
};
