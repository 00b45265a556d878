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

