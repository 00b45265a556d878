// a3 (+ the 32->32 2-D convolutions of a1/a7): fp32 implicit-GEMM convolution on
// the CDNA4 matrix cores, forward / data-gradient / weight-gradient.
//
// Reference semantics: nn.Conv3d(32,32,3,padding=1) inside convbn_3d
// (adaptive_stereo/models/stereo_net.py:21-30, applied :185-186) and the
// nn.Conv2d(32,32,3,dilation=d,padding=d) of convbn (stereo_net.py:10-18).
//
// Why fp32 MFMA: parity with the reference's fp32 CPU path (disparity EPE <= 1e-3,
// arg-max indices equal) rules out bf16 inputs; gfx950 has no xf32.  The f32-input
// v_mfma_f32_32x32x2_f32 is an exact fp32 fma chain at the fp32 vector peak
// (157 TFLOP/s), so this kernel is MFMA-issue bound: 64 cycles per MFMA leave ample
// issue slots, and operands come straight from L1/L2 with 16-byte loads — there is
// no LDS staging because every operand float feeds exactly one MFMA lane and the
// PCL layout already makes each lane's 16 K-values one contiguous 64-byte run.
//
// GEMM view (forward):  Z[v][co] = sum_{tap} sum_{ci} X[v + off(tap)][ci] * W[tap][ci][co]
//   M = voxels (32 per wave tile, lane&31), N = 32 output channels, K = taps*32.
//   K is ordered (tap, s, h) with ci = 16*h + s, h = lane>>5: the lane-half h of an
//   MFMA holds k = h, so per tap a lane loads ci = 16h..16h+15 of its voxel: 4 x
//   dwordx4.  Weights are pre-packed to [tap][q][lane][4] so each of the B operand's
//   4 x dwordx4 per tap is one contiguous KB per wave (L1/L2 resident: 110 KB for 27
//   taps).  The zero halo of PCL supplies the padding: no predicates.
//   2-D 3x3 stride-1 layers take the LDS-staged kernel in conv32_lds.hip instead.
//
// The data gradient is the same kernel run on mirrored/transposed weights.
// The weight gradient is a second kernel: M = ci, N = co, K = voxels.
#include "as_common.h"
#include "conv_epilogue.h"
#include "conv32_lds.h"
#include <cstdlib>
#include "conv3d_lds.h"
#include "conv32_bwd.h"
#include "conv32_act.h"
#include "conv32_wino.h"
#include "conv32_s2.h"

static bool wgrad_lds_applicable(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s);

// The 5x5 stride-2 layers of the feature head on csrc/conv32_s2.hip (1, default) or on the generic direct-load kernels (0):
// same bits either way (the parity tests compare them); returns the previous setting.
static int g_conv32_s2 = 1;
extern "C" int as_conv32_s2_enable(int on) {
  const int prev = g_conv32_s2;
  if (on == 0 || on == 1) g_conv32_s2 = on;
  return prev;
}

struct ConvArgs {
  const float* x;
  const float* wp;
  EpilogueArgs ep;
  PclDev gin, gout;
  ConvMap map;
  int M;          // output voxels of this launch
  int ntaps;
  int tap_off[AS_MAX_TAPS];   // voxel offsets in the INPUT geometry
};

__device__ inline void load16(f32x4 (&r)[4], const float* p) {
  const f32x4* q = reinterpret_cast<const f32x4*>(p);
  r[0] = q[0]; r[1] = q[1]; r[2] = q[2]; r[3] = q[3];
}

__device__ inline void loadw(f32x4 (&r)[4], const float* p) {
  r[0] = *reinterpret_cast<const f32x4*>(p);
  r[1] = *reinterpret_cast<const f32x4*>(p + 256);
  r[2] = *reinterpret_cast<const f32x4*>(p + 512);
  r[3] = *reinterpret_cast<const f32x4*>(p + 768);
}

__device__ inline void mfma16(f32x16& acc, const f32x4 (&a)[4], const f32x4 (&b)[4]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, b[q].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, b[q].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, b[q].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, b[q].w, acc, 0, 0, 0);
  }
}

// One wave = one 32-voxel x 32-channel output tile; 4 waves per workgroup.
// NT = number of taps, a compile-time constant: the tap loop is fully unrolled into straight-line
// code with a static two-deep register ring (tap t+1's eight 16-byte loads are issued before tap
// t's sixteen MFMAs).  A runtime tap loop made hipcc (ROCm 7.2) (a) bounce the accumulator between
// AGPRs and VGPRs at every loop-carried branch, draining the matrix pipe, and (b) wait vmcnt(3)
// immediately after issuing the prefetch — 40 % of the fp32 MFMA peak instead of the pipelined rate.
#ifndef AS_CONV32_PREFETCH
#define AS_CONV32_PREFETCH 1
#endif
template <int NT>
__device__ __forceinline__ void conv32_fwd_body(const ConvArgs& p, const int block, float (*red)[32], float* bmean) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, li = lane & 31;
  const int tile = block * 4 + wave;
  const int v = tile * 32 + li;
  const bool valid = v < p.M;
  int in_vox, out_vox;
  conv_decode(valid ? v : p.M - 1, p.gin, p.gout, p.map, in_vox, out_vox);

  const float* xa = p.x + (long)in_vox * 32 + h * 16;
  const float* wb = p.wp + lane * 4;       // packed [tap][q][lane][4]: each wave load is one contiguous KB

  // the bias is requested first and needed (accumulator init) only after the prologue; the load is unconditional
  // so that no select forces a wait here
  const float bias_v = p.ep.bias[li];                      // never null here: launch_conv32 substitutes 32 zeros
  __builtin_amdgcn_sched_barrier(0);

  // Operands are requested PF taps ahead (a ring of PF+1 register sets): on the small maps (24x78 at 1/16 resolution)
  // every wave of a launch is resident at once and the kernel's duration is ONE wave's chain of NT dependent
  // load -> 16 MFMA rounds; with a single tap in flight a round took ~1800 cycles against 1024 of matrix work.
  constexpr int PF = AS_CONV32_PREFETCH, RING = PF + 1;
  f32x4 a[RING][4], b[RING][4];
#pragma unroll
  for (int tp = 0; tp < PF && tp < NT; ++tp) {
    load16(a[tp % RING], xa + (long)p.tap_off[tp] * 32);
    loadw(b[tp % RING], wb + tp * 1024);
    __builtin_amdgcn_sched_barrier(0);                     // in tap order: vmcnt retires in issue order
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = bias_v;
#pragma unroll
  for (int tp = 0; tp < NT; ++tp) {
    if (tp + PF < NT) {
      load16(a[(tp + PF) % RING], xa + (long)p.tap_off[tp + PF] * 32);
      loadw(b[(tp + PF) % RING], wb + (tp + PF) * 1024);
    }
    // keep the prefetch above the MFMAs: without the fences hipcc sinks every load next to its
    // use (2 loads -> vmcnt -> 4 MFMAs), trading the latency hiding for registers
    __builtin_amdgcn_sched_barrier(0);
    mfma16(acc, a[tp % RING], b[tp % RING]);
    __builtin_amdgcn_sched_barrier(0);
  }
  TileStats ts;
  conv_epilogue(acc, p.ep, out_vox, valid, min(128, p.M - block * 128), red, bmean, &ts);
  stats_write(p.ep, block, ts);
}

template <int NT>
__global__ __launch_bounds__(256) void conv32_fwd_kernel(ConvArgs p) {
  __shared__ float red[4][32];
  __shared__ float bmean[32];
  conv32_fwd_body<NT>(p, (int)blockIdx.x, red, bmean);
}

// The four parity phases of a stride-2 5x5 data gradient (9 / 6 / 6 / 4 taps, see as_conv32_dgrad_s2) in ONE launch: a
// workgroup finds its phase from its block index and runs that phase's fully unrolled body.  Four launches per layer were
// 12 launches per step on the three small levels of the feature head, each a few hundred workgroups at most.
struct DgradS2Args { ConvArgs ph[4]; int nblk[4]; };

__global__ __launch_bounds__(256) void conv32_dgrad_s2_kernel(DgradS2Args p) {
  __shared__ float red[4][32];
  __shared__ float bmean[32];
  int b = (int)blockIdx.x;
  if (b < p.nblk[0]) { conv32_fwd_body<9>(p.ph[0], b, red, bmean); return; }
  b -= p.nblk[0];
  if (b < p.nblk[1]) { conv32_fwd_body<6>(p.ph[1], b, red, bmean); return; }
  b -= p.nblk[1];
  if (b < p.nblk[2]) { conv32_fwd_body<6>(p.ph[2], b, red, bmean); return; }
  b -= p.nblk[2];
  conv32_fwd_body<4>(p.ph[3], b, red, bmean);
}

// Split-K flavour for small maps.  On a 24x78 map (1/16 resolution: 59 workgroups of 128 voxels at 4 pairs, 15 at one) the
// kernel above lasts as long as ONE wave's chain of NT x 16 dependent MFMAs plus its loads (11 us for 9 taps), on a chip that
// is mostly idle.  Here a workgroup owns ONE 32-voxel tile and its four waves split the taps (wave w takes taps w, w+4, ...):
// a quarter of the chain, four times the workgroups; waves 1-3 hand their partial accumulators over through LDS and wave 0
// adds them in fixed order and runs the ordinary epilogue (one BatchNorm partial per 32 voxels).  A different summation order
// than the one-wave chain (four partial sums), deterministic.  (Build with EXTRA=-DCONV32_SPLITK_MAX_M=0 for an A/B run.)
#ifndef CONV32_SPLITK_MAX_M
#define CONV32_SPLITK_MAX_M 32768
#endif
static inline bool conv32_splitk_applies(int ntaps, long M) {
  // 25 taps (the strided head's 5x5 convolutions, round 3): a one-wave chain is 400 dependent MFMAs = 12 us; on the two
  // smallest levels (<= 16 K output voxels: fewer wave tiles than the chip has SIMDs) the split chain of 112 wins
  return (ntaps == 9 && M <= CONV32_SPLITK_MAX_M) || (ntaps == 25 && M <= CONV32_SPLITK_MAX_M / 2);
}

template <int NT>
__global__ __launch_bounds__(256) void conv32_fwd_splitk_kernel(ConvArgs p) {
  __shared__ float red[4][32];
  __shared__ float bmean[32];
  __shared__ float part[3][16][64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, li = lane & 31;
  const int v = blockIdx.x * 32 + li;
  const bool in_range = v < p.M;
  int in_vox, out_vox;
  conv_decode(in_range ? v : p.M - 1, p.gin, p.gout, p.map, in_vox, out_vox);
  const float* xa = p.x + (long)in_vox * 32 + h * 16;
  const float* wb = p.wp + lane * 4;
  const float bias_v = wave == 0 ? p.ep.bias[li] : 0.f;
  constexpr int PER = (NT + 3) / 4;                       // taps per wave (the last ones may have one less)
  f32x4 a[2][4], b[2][4];
  if (wave < NT) {
    load16(a[0], xa + (long)p.tap_off[wave] * 32);
    loadw(b[0], wb + wave * 1024);
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = bias_v;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int tp = wave + 4 * k;                          // wave-uniform
    if (k + 1 < PER && tp + 4 < NT) {
      load16(a[(k + 1) & 1], xa + (long)p.tap_off[tp + 4] * 32);
      loadw(b[(k + 1) & 1], wb + (tp + 4) * 1024);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (tp < NT) mfma16(acc, a[k & 1], b[k & 1]);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = ((acc[r] + part[0][r][lane]) + part[1][r][lane]) + part[2][r][lane];
  }
  TileStats ts;
  conv_epilogue(acc, p.ep, out_vox, in_range && wave == 0, min(32, p.M - (int)blockIdx.x * 32), red, bmean, &ts);
  stats_write(p.ep, blockIdx.x, ts);
}

// ---------------------------------------------------------------------------------
// Weight packing: PyTorch [O][I][taps] -> MFMA B-operand order [tap][q][lane=(h,j)][e], k = 16h + 4q + e:
// lane (h, j) reads one float4 per q, and the 64 lanes of a wave read one contiguous KB.
//   forward : packed = w[o=j][i=k][t]
//   dgrad   : packed = w[o=k][i=j][T-1-t]
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ packed, int T, int flip) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= T * 1024) return;
  const int e = idx & 3, j = (idx >> 2) & 31, h = (idx >> 7) & 1, q = (idx >> 8) & 3, t = idx >> 10;
  const int k = 16 * h + 4 * q + e;
  float v;
  if (!flip) v = w[((long)j * 32 + k) * T + t];
  else       v = w[((long)k * 32 + j) * T + (T - 1 - t)];
  packed[idx] = v;
}

// ---------------------------------------------------------------------------------
// Weight gradient: dW[tap][ci][co] = sum_v X[v + off(tap)][ci] * G[v][co].
//   MFMA: i = ci (lane&31 of A), j = co (lane&31 of B), k = voxel (lane>>5 selects the
//   even/odd voxel of a pair).  Both operands are one coalesced 256-byte wave load per
//   MFMA.  A workgroup owns (tap group, chunk of image rows); its 4 waves take rows
//   round-robin, reduce through LDS and write one partial slab; a second kernel sums
//   the slabs in a fixed order (deterministic — no float atomics).
#define WG_SEG_STEPS 32            // multiple of the 8-step load groups
template <int TG>
struct WgradAcc { f32x16 a[TG]; };

struct WgradArgs {
  const float* x;
  const float* gz;
  float* partial;     // [nchunks][ntaps][32][32]
  float* partial_db;  // [nchunks][32]
  PclDev gin, gout;
  int rows;           // work units: B*D*H rows x nseg segments
  int nseg;           // segments per row
  int seg_steps;      // voxel pairs per segment (a multiple of the 8-step load groups)
  int rows_per_chunk; // units per workgroup
  int nchunks;
  int ntaps;
  int stride;
  int tap_off[AS_MAX_TAPS];
};

template <int TG>
__global__ __launch_bounds__(256) void conv32_wgrad_kernel(WgradArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [3 waves][TG*16 regs][64 lanes] + db
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, li = lane & 31;
  // XCD-aware (tap group, row chunk) assignment: the groups of one chunk read the same gz rows and
  // overlapping x rows, so they are given block ids that the round-robin dispatcher places on ONE XCD
  // (ids congruent mod 8) and that are adjacent in time — their re-reads then hit that XCD's L2 instead
  // of HBM (3 groups on 3 XCDs fetched 2.9x the algorithmic bytes).  Placement affects speed only.
  int group, chunk;
  {
    const int ngroups = p.ntaps / TG;
    const int L = blockIdx.x, xcd = L & 7, q = L >> 3;
    const int per_xcd = (p.nchunks + 7) >> 3;              // chunks handled by each XCD
    group = q % ngroups;
    chunk = xcd * per_xcd + q / ngroups;
    if (chunk >= p.nchunks || q / ngroups >= per_xcd) return;   // padding blocks (uniform per workgroup)
  }
  const int W = p.gout.W, H = p.gout.H, D = p.gout.D;
  const int r0 = chunk * p.rows_per_chunk;
  const int r1 = min(p.rows, r0 + p.rows_per_chunk);

  f32x16 acc[TG];
#pragma unroll
  for (int g = 0; g < TG; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
  float bsum = 0.f;

  int toff[TG];
#pragma unroll
  for (int g = 0; g < TG; ++g) toff[g] = p.tap_off[group * TG + g] * 32;

  // K loop over voxel pairs of the wave's rows.  Loads of WG_U steps (WG_U*(TG+1) coalesced 256-byte
  // wave loads) are issued as a group before their MFMAs so that the memory latency of a whole group
  // — not of a single step — is what the co-resident waves have to cover.
  constexpr int WG_U = 8;
  const int nsteps = (W + 1) >> 1;
  const int xs = 32 * p.stride;
  for (int unit = r0 + wave; unit < r1; unit += 4) {
    // a wave's unit of work: one segment (WG_SEG_STEPS voxel pairs) of one row — whole rows left most of the chip
    // idle on small batches (94 rows of 156 steps for the first stride-2 layer of one image)
    const int row = unit / p.nseg, seg = unit - row * p.nseg;
    int t = row;
    const int y = t % H; t /= H;
    const int d = t % D;
    const int b = t / D;
    const float* xr = p.x + p.gin.vox(b, d, y * p.stride, 0) * 32 + li;
    const float* gr = p.gz + p.gout.vox(b, d, y, 0) * 32 + li;
    const int s_end = min(nsteps, (seg + 1) * p.seg_steps);
    for (int s0 = seg * p.seg_steps; s0 < s_end; s0 += WG_U) {
      float bv[WG_U], av[WG_U][TG];
#pragma unroll
      for (int u = 0; u < WG_U; ++u) {
        const int xc = 2 * (s0 + u) + h;
        const bool ok = xc < W;
        const int xcl = ok ? xc : W - 1;
#ifdef WG_EXP_NOLOAD                           // diagnostic build (tests/tools/head_exp.sh): results are wrong, only the time counts
        const float g0 = (float)(xcl + lane);
        bv[u] = ok ? g0 : 0.f;
#pragma unroll
        for (int g = 0; g < TG; ++g) av[u][g] = (float)(xcl * xs + toff[g]);
#else
        const float g0 = gr[xcl * 32];
        bv[u] = ok ? g0 : 0.f;
#pragma unroll
        for (int g = 0; g < TG; ++g) av[u][g] = xr[xcl * xs + toff[g]];
#endif
      }
      __builtin_amdgcn_sched_barrier(0);      // all loads of the group are issued before its first MFMA
#pragma unroll
      for (int u = 0; u < WG_U; ++u) {
        bsum += bv[u];
#pragma unroll
        for (int g = 0; g < TG; ++g)
#ifdef WG_EXP_NOMFMA
          acc[g][0] += av[u][g] * bv[u];
#else
          acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][g], bv[u], acc[g], 0, 0, 0);
#endif
      }
    }
  }

  // Reduce the 4 waves' accumulators through LDS (fixed order: w0 + w1 + w2 + w3).
  float* slab = lds;                       // [3][TG*16][64]
  float* dbs = lds + 3 * TG * 16 * 64;     // [4][32]
  bsum += __shfl_xor(bsum, 32, 64);
  if (h == 0) dbs[wave * 32 + li] = bsum;
  if (wave > 0) {
#pragma unroll
    for (int g = 0; g < TG; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) slab[((wave - 1) * TG * 16 + g * 16 + r) * 64 + lane] = acc[g][r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int g = 0; g < TG; ++g) {
      const int tap = group * TG + g;
      float* out = p.partial + ((long)chunk * p.ntaps + tap) * 1024;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[g][r];
        v += slab[(0 * TG * 16 + g * 16 + r) * 64 + lane];
        v += slab[(1 * TG * 16 + g * 16 + r) * 64 + lane];
        v += slab[(2 * TG * 16 + g * 16 + r) * 64 + lane];
        const int ci = (r & 3) + 8 * (r >> 2) + 4 * h;
        out[ci * 32 + li] = v;
      }
    }
    if (group == 0 && h == 0)
      p.partial_db[chunk * 32 + li] = dbs[li] + dbs[32 + li] + dbs[64 + li] + dbs[96 + li];
  }
}

// dW[o][i][t] = sum_chunks partial[chunk][t][i][o];  db[o] = sum_chunks partial_db[chunk][o].
// 1024 threads = 64 consecutive outputs x 16 slab slices: each thread sums every 16th slab (coalesced
// 256-byte wave loads, 8 independent loads in flight), then the slices are added in fixed order through
// LDS.  A single thread per output walking all slabs was a ~500-deep chain of dependent-latency loads.
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ partial,
                                                             const float* __restrict__ partial_db, int nchunks, int T,
                                                             float* __restrict__ dW, float* __restrict__ db, int accumulate) {
  __shared__ float red[16][64];
  const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int total = T * 1024;
  const int idx = blockIdx.x * 64 + o;
  const bool is_w = idx < total, is_b = !is_w && db != nullptr && idx < total + 32;
  float s = 0.f;
  if (is_w) {
    const float* src = partial + idx;
#pragma unroll 8
    for (int c = sl; c < nchunks; c += 16) s += src[(long)c * total];
  } else if (is_b) {
    const float* src = partial_db + (idx - total);
#pragma unroll 8
    for (int c = sl; c < nchunks; c += 16) s += src[c * 32];
  }
  red[sl][o] = s;
  __syncthreads();
  if (sl == 0 && (is_w || is_b)) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][o];
    if (is_w) {
      const int oc = idx & 31, i = (idx >> 5) & 31, tp = idx >> 10;
      float* dst = dW + ((long)oc * 32 + i) * T + tp;
      *dst = accumulate ? *dst + t : t;
    } else {
      db[idx - total] = accumulate ? db[idx - total] + t : t;
    }
  }
}

// ---- deferred reductions ---------------------------------------------------------------------------------------------
// A step has ~30 of these reductions, 4.5 us each (launch-bound: a few hundred KB of slabs), one behind every weight-gradient
// kernel.  Nothing before the optimizer reads dW / db, so a caller may open a deferral region (as_wgrad_defer(1)): the weight-
// gradient entry points then only RECORD their reduction, and as_wgrad_defer_flush() runs all of them in one launch — grouped by
// destination, the jobs of one destination (a layer used twice: left and right feature tower) applied in recording order with
// the arithmetic of the separate launches (bit-identical).  The caller keeps the workspaces alive until the flush.
// The record is process-global behind a mutex: autograd runs backward nodes on its own thread.
struct ReduceJob { const float* partial; const float* partial_db; int nchunks, accumulate; };
#define RB_MAX_DEST 32
#define RB_MAX_JOBS 2
struct ReduceDest { float* dW; float* db; int T, njobs; ReduceJob job[RB_MAX_JOBS]; };
struct ReduceBatch { int ndest; ReduceDest dest[RB_MAX_DEST]; };

__global__ __launch_bounds__(1024) void wgrad_reduce_batch_kernel(ReduceBatch rb) {
  // 1024 threads = 64 groups of FOUR consecutive slab elements x 16 slab slices: a slice reads 1 KB of a slab at a time
  // (float4 per thread; with one float per thread, 256-byte pieces 36 KB apart, the batch ran at 1.7 TB/s)
  __shared__ f32x4 red[16][64];
  const ReduceDest& d = rb.dest[blockIdx.y];
  const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int total = d.T * 1024;
  const int idx = (blockIdx.x * 64 + o) * 4;                       // first of this thread's four elements
  if (blockIdx.x * 256 >= total + 32) return;                      // (workgroup-uniform)
  const bool is_w = idx < total, is_b = !is_w && d.db != nullptr && idx < total + 32;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  bool have = false;
  for (int j = 0; j < d.njobs; ++j) {
    const ReduceJob& jb = d.job[j];
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (is_w) {
      const float* src = jb.partial + idx;
#pragma unroll 8
      for (int c = sl; c < jb.nchunks; c += 16) s += *reinterpret_cast<const f32x4*>(src + (long)c * total);
    } else if (is_b) {
      const float* src = jb.partial_db + (idx - total);
#pragma unroll 8
      for (int c = sl; c < jb.nchunks; c += 16) s += *reinterpret_cast<const f32x4*>(src + c * 32);
    }
    __syncthreads();
    red[sl][o] = s;
    __syncthreads();
    if (sl == 0 && (is_w || is_b)) {
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 16; ++k) t += red[k][o];
      if (jb.accumulate) {
        if (!have) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int id = idx + e;
            v[e] = is_w ? d.dW[((long)(id & 31) * 32 + ((id >> 5) & 31)) * d.T + (id >> 10)] : d.db[id - total];
          }
          have = true;
        }
        v = v + t;
      } else { v = t; have = true; }
    }
  }
  if (sl == 0 && (is_w || is_b)) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int id = idx + e;
      if (is_w) d.dW[((long)(id & 31) * 32 + ((id >> 5) & 31)) * d.T + (id >> 10)] = v[e];
      else d.db[id - total] = v[e];
    }
  }
}

#include <mutex>
#include <vector>
struct PendingReduce { const float* partial; const float* partial_db; int nchunks, T; float* dW; float* db; int accumulate; };
static std::mutex g_defer_mutex;
static bool g_defer_on = false;
static std::vector<PendingReduce> g_defer_jobs;
static void wgrad_reduce(hipStream_t st, const float* partial, const float* partial_db, int nchunks, int T, float* dW, float* db,
                         int accumulate);
// (for the other translation units: trunk.hip)
void as_wgrad_reduce_enqueue(hipStream_t st, const float* partial, const float* partial_db, int nchunks, int T, float* dW,
                             float* db, int accumulate) {
  wgrad_reduce(st, partial, partial_db, nchunks, T, dW, db, accumulate);
}

// every weight-gradient entry point ends in this
static void wgrad_reduce(hipStream_t st, const float* partial, const float* partial_db, int nchunks, int T, float* dW, float* db,
                         int accumulate) {
  {
    std::lock_guard<std::mutex> lock(g_defer_mutex);
    // only reductions that ACCUMULATE into a gradient sink wait for the flush; a dW that is handed back to the caller
    // (accumulate == 0: autograd reads it as soon as the entry point returns) is reduced right here
    if (g_defer_on && accumulate) { g_defer_jobs.push_back({partial, partial_db, nchunks, T, dW, db, accumulate}); return; }
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(as_div_up(T * 1024 + 32, 64)), dim3(1024), 0, st, partial, partial_db, nchunks, T,
                     dW, db, accumulate);
}

extern "C" int as_wgrad_defer(int on) {
  std::lock_guard<std::mutex> lock(g_defer_mutex);
  const int prev = g_defer_on ? 1 : 0;
  g_defer_on = on != 0;
  return prev;
}

extern "C" int as_wgrad_defer_pending(void) {
  std::lock_guard<std::mutex> lock(g_defer_mutex);
  return (int)g_defer_jobs.size();
}

static void launch_reduce_batch(const ReduceBatch& rb, hipStream_t st) {
  int tmax = 0;
  for (int i = 0; i < rb.ndest; ++i) tmax = rb.dest[i].T > tmax ? rb.dest[i].T : tmax;
  hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(as_div_up(tmax * 1024 + 32, 256), rb.ndest), dim3(1024), 0, st, rb);
}

extern "C" int as_wgrad_defer_flush(void* stream) {
  std::vector<PendingReduce> jobs;
  {
    std::lock_guard<std::mutex> lock(g_defer_mutex);
    jobs.swap(g_defer_jobs);
  }
  hipStream_t st = (hipStream_t)stream;
  ReduceBatch rb;
  rb.ndest = 0;
  for (const PendingReduce& pj : jobs) {
    int di = -1;
    for (int i = 0; i < rb.ndest; ++i) if (rb.dest[i].dW == pj.dW) di = i;
    if (di >= 0 && (rb.dest[di].njobs == RB_MAX_JOBS || rb.dest[di].T != pj.T || rb.dest[di].db != pj.db)) {
      launch_reduce_batch(rb, st);          // a third job for a destination: what is gathered so far goes first (order!)
      rb.ndest = 0; di = -1;
    }
    if (di < 0) {
      if (rb.ndest == RB_MAX_DEST) { launch_reduce_batch(rb, st); rb.ndest = 0; }
      di = rb.ndest++;
      rb.dest[di].dW = pj.dW; rb.dest[di].db = pj.db; rb.dest[di].T = pj.T; rb.dest[di].njobs = 0;
    }
    ReduceJob& jb = rb.dest[di].job[rb.dest[di].njobs++];
    jb.partial = pj.partial; jb.partial_db = pj.partial_db; jb.nchunks = pj.nchunks; jb.accumulate = pj.accumulate;
  }
  if (rb.ndest > 0) launch_reduce_batch(rb, st);
  AS_CHECK_LAUNCH("as_wgrad_defer_flush");
  return AS_OK;
}

// ---------------------------------------------------------------------------------
// Host side
static int fill_taps(const as_pcl* gin, const as_conv_shape* s, int* tap_off, const char* who) {
  const int T = s->kd * s->kh * s->kw;
  AS_CHECK_ARG(T >= 1 && T <= AS_MAX_TAPS, "%s: %d taps unsupported", who, T);
  AS_CHECK_ARG(s->dil >= 1 && s->stride >= 1, "%s: bad dilation/stride", who);
  const int Hp = gin->H + 2 * gin->ph, Wp = gin->W + 2 * gin->pw;
  int n = 0;
  for (int i = 0; i < s->kd; ++i)
    for (int j = 0; j < s->kh; ++j)
      for (int l = 0; l < s->kw; ++l) {
        const int od = (s->kd > 1 ? i * s->dil : 0) - s->pad_d;
        const int oh = j * s->dil - s->pad_h;
        const int ow = l * s->dil - s->pad_w;
        tap_off[n++] = (od * Hp + oh) * Wp + ow;
      }
  return AS_OK;
}

static int check_conv(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s, const char* who) {
  AS_CHECK_ARG(as_pcl_ok(gin) && as_pcl_ok(gout) && s, "%s: bad geometry", who);
  AS_CHECK_ARG(gin->B == gout->B && gin->D == gout->D, "%s: batch/depth mismatch", who);
  // output extent of the convolution
  const int eh = (gin->H + 2 * s->pad_h - s->dil * (s->kh - 1) - 1) / s->stride + 1;
  const int ew = (gin->W + 2 * s->pad_w - s->dil * (s->kw - 1) - 1) / s->stride + 1;
  AS_CHECK_ARG(eh == gout->H && ew == gout->W, "%s: output extent %dx%d != conv result %dx%d", who,
               gout->H, gout->W, eh, ew);
  if (s->kd > 1)
    AS_CHECK_ARG(gin->D + 2 * s->pad_d - s->dil * (s->kd - 1) == gout->D, "%s: depth extent mismatch", who);
  // every tap of every output must land inside the input's halo
  const int reach_h_lo = s->pad_h, reach_w_lo = s->pad_w;
  const int reach_h_hi = (gout->H - 1) * s->stride + s->dil * (s->kh - 1) - s->pad_h - (gin->H - 1);
  const int reach_w_hi = (gout->W - 1) * s->stride + s->dil * (s->kw - 1) - s->pad_w - (gin->W - 1);
  AS_CHECK_ARG(reach_h_lo <= gin->ph && reach_w_lo <= gin->pw && reach_h_hi <= gin->ph && reach_w_hi <= gin->pw,
               "%s: input halo (%d,%d) too small for padding", who, gin->ph, gin->pw);
  if (s->kd > 1) AS_CHECK_ARG(s->pad_d <= gin->pd && s->dil * (s->kd - 1) - s->pad_d <= gin->pd,
                              "%s: input depth halo too small", who);
  return AS_OK;
}

__device__ float g_zero_bias[32];          // zero-initialised: the bias of a convolution that has none

// 32 zeros in device memory (the convolution bodies read their bias unconditionally), per HIP device
static const float* zero_bias_ptr(void) {
  static const float* zeros[32] = {};
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 32) return nullptr;
  if (zeros[d] == nullptr) {
    void* sym = nullptr;
    if (hipGetSymbolAddress(&sym, HIP_SYMBOL(g_zero_bias)) == hipSuccess) zeros[d] = static_cast<const float*>(sym);
  }
  return zeros[d];
}

static int launch_conv32(const ConvArgs& args, hipStream_t st, const char* who) {
  ConvArgs a = args;
  if (a.ep.bias == nullptr) {
    const float* zeros = zero_bias_ptr();
    if (zeros == nullptr) { as_set_error("%s: no address for the zero bias", who); return AS_ERR_LAUNCH; }
    a.ep.bias = zeros;
  }
  const dim3 grid(as_div_up(a.M, 128)), block(256);
  if (conv32_splitk_applies(a.ntaps, a.M)) {
    if (a.ntaps == 9) hipLaunchKernelGGL(conv32_fwd_splitk_kernel<9>, dim3(as_div_up(a.M, 32)), block, 0, st, a);
    else hipLaunchKernelGGL(conv32_fwd_splitk_kernel<25>, dim3(as_div_up(a.M, 32)), block, 0, st, a);
    return AS_OK;
  }
  switch (a.ntaps) {
    case 27: hipLaunchKernelGGL(conv32_fwd_kernel<27>, grid, block, 0, st, a); break;
    case 25: hipLaunchKernelGGL(conv32_fwd_kernel<25>, grid, block, 0, st, a); break;
    case 9:  hipLaunchKernelGGL(conv32_fwd_kernel<9>, grid, block, 0, st, a); break;
    case 6:  hipLaunchKernelGGL(conv32_fwd_kernel<6>, grid, block, 0, st, a); break;
    case 4:  hipLaunchKernelGGL(conv32_fwd_kernel<4>, grid, block, 0, st, a); break;
    default:
      as_set_error("%s: no kernel instance for %d taps", who, a.ntaps);
      return AS_ERR_ARG;
  }
  return AS_OK;
}

// ---- data gradient of the stride-2 5x5 convolution (downsample[1..k-1], stereo_net.py:61-69) --------
// gx[yi][xi][ci] = sum over taps (j,l) with (yi+2-j), (xi+2-l) even of gz[(yi+2-j)/2][(xi+2-l)/2][co] * W[co][ci][j][l].
// Split by the parity (py,px) of (yi,xi): each phase is an ordinary gather-convolution over gz with the
// taps j = py, py+2(, py+4), l likewise (9/6/6/4 taps), writing every second output pixel.  gz's zero
// halo (>= 1) supplies the bounds.  packed[t][q][lane=(h,j)][e] = w[o = 16h+4q+e][i = j][tap_t].
struct TapSubset { int n; int idx[25]; };

__global__ void pack_weights_subset_kernel(const float* __restrict__ w, float* __restrict__ packed, int T, TapSubset ts) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= ts.n * 1024) return;
  const int e = idx & 3, j = (idx >> 2) & 31, h = (idx >> 7) & 1, q = (idx >> 8) & 3, t = idx >> 10;
  const int k = 16 * h + 4 * q + e;
  packed[idx] = w[((long)k * 32 + j) * T + ts.idx[t]];
}

extern "C" int64_t as_conv32_dgrad_s2_workspace(void) { return 25 * 1024; }

extern "C" int as_conv32_dgrad_s2_pack(const float* w, float* packed, void* stream) {
  AS_CHECK_ARG(w && packed, "as_conv32_dgrad_s2_pack: null pointer");
  // all four parity phases' weights in one launch: the 25 taps in phase-major order (9 + 6 + 6 + 4)
  TapSubset all; all.n = 0;
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px)
      for (int j = py; j < 5; j += 2)
        for (int l = px; l < 5; l += 2) all.idx[all.n++] = j * 5 + l;
  hipLaunchKernelGGL(pack_weights_subset_kernel, dim3(as_div_up(all.n * 1024, 256)), dim3(256), 0, (hipStream_t)stream, w, packed, 25,
                     all);
  AS_CHECK_LAUNCH("as_conv32_dgrad_s2_pack");
  return AS_OK;
}

extern "C" int as_conv32_dgrad_s2(const float* gz, const as_pcl* ggz, const float* w, float* gx, const as_pcl* ggx,
                                  float* workspace, void* stream) {
  AS_CHECK_ARG(w && workspace, "as_conv32_dgrad_s2: null pointer");
  if (int e = as_conv32_dgrad_s2_pack(w, workspace, stream)) return e;
  return as_conv32_dgrad_s2_packed(gz, ggz, workspace, gx, ggx, stream);
}

extern "C" int as_conv32_dgrad_s2_packed(const float* gz, const as_pcl* ggz, const float* packed, float* gx, const as_pcl* ggx,
                                         void* stream) {
  const float* w = packed; float* workspace = const_cast<float*>(packed);
  AS_CHECK_ARG(as_pcl_ok(ggz) && as_pcl_ok(ggx) && gz && w && gx && workspace, "as_conv32_dgrad_s2: bad argument");
  AS_CHECK_ARG(ggz->D == 1 && ggx->D == 1 && ggz->B == ggx->B, "as_conv32_dgrad_s2: 2-D tensors of equal batch expected");
  AS_CHECK_ARG(ggz->H == (ggx->H - 1) / 2 + 1 && ggz->W == (ggx->W - 1) / 2 + 1,
               "as_conv32_dgrad_s2: gz extent is not that of a 5x5 stride-2 pad-2 convolution of gx's extent");
  AS_CHECK_ARG(ggz->ph >= 1 && ggz->pw >= 1, "as_conv32_dgrad_s2: gz needs a zero halo of 1");
  hipStream_t st = (hipStream_t)stream;
  if (g_conv32_s2 && conv32_s2_dgrad_applicable(ggz, ggx)) {
    as_prof_mark(AS_PROF_CONV32, st, 1, 0.0);
    if (int e = conv32_s2_dgrad_launch(gz, ggz, packed, gx, ggx, stream)) return e;
    as_prof_mark(AS_PROF_CONV32, st, 0, 2.0 * (double)ggx->B * ggz->H * ggz->W * 1024.0 * 25.0);
    AS_CHECK_LAUNCH("as_conv32_dgrad_s2(staged)");
    return AS_OK;
  }
  const int Wp = ggz->W + 2 * ggz->pw;
  const float* wp = workspace;
  DgradS2Args args;
  int total = 0, ph = 0;
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px, ++ph) {
      const int Hl = (ggx->H - py + 1) / 2, Wl = (ggx->W - px + 1) / 2;
      ConvArgs& a = args.ph[ph];
      int n = 0;
      for (int j = py; j < 5; j += 2)
        for (int l = px; l < 5; l += 2) {
          // yi = 2y'+py: yo = y' + (py + 2 - j)/2 ; same for x
          a.tap_off[n] = ((py + 2 - j) / 2) * Wp + (px + 2 - l) / 2;
          ++n;
        }
      a.x = gz; a.wp = wp;
      a.ep.bias = nullptr; a.ep.z = gx; a.ep.ep_scale = nullptr; a.ep.ep_shift = nullptr; a.ep.residual = nullptr;
      a.ep.stat_mean = nullptr; a.ep.stat_m2 = nullptr; a.ep.stat_cnt = nullptr; a.ep.epilogue = 0; a.ep.slope = 0.f;
      a.gin = as_make_dev(ggz); a.gout = as_make_dev(ggx);
      a.map.Hl = Hl > 0 ? Hl : 1; a.map.Wl = Wl > 0 ? Wl : 1; a.map.in_stride = 1; a.map.out_stride = 2; a.map.out_oy = py; a.map.out_ox = px;
      a.M = (Hl > 0 && Wl > 0) ? ggx->B * Hl * Wl : 0; a.ntaps = n;
      args.nblk[ph] = as_div_up(a.M, 128);
      total += args.nblk[ph];
      wp += n * 1024;
    }
  const float* zero_bias = zero_bias_ptr();                // (the bias of conv32_fwd_body is read unconditionally)
  if (zero_bias == nullptr) { as_set_error("as_conv32_dgrad_s2: no address for the zero bias"); return AS_ERR_LAUNCH; }
  for (int i = 0; i < 4; ++i) args.ph[i].ep.bias = zero_bias;
  as_prof_mark(AS_PROF_CONV32, st, 1, 0.0);
  if (total > 0) hipLaunchKernelGGL(conv32_dgrad_s2_kernel, dim3(total), dim3(256), 0, st, args);
  as_prof_mark(AS_PROF_CONV32, st, 0, 2.0 * (double)ggx->B * ggz->H * ggz->W * 1024.0 * 25.0);
  AS_CHECK_LAUNCH("as_conv32_dgrad_s2");
  return AS_OK;
}

extern "C" int as_conv32_pack_weights(const float* w, float* packed, const as_conv_shape* s,
                                      int transpose_flip, void* stream) {
  AS_CHECK_ARG(w && packed && s, "as_conv32_pack_weights: null pointer");
  const int T = s->kd * s->kh * s->kw;
  AS_CHECK_ARG(T >= 1 && T <= AS_MAX_TAPS, "as_conv32_pack_weights: %d taps unsupported", T);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(as_div_up(T * 1024, 256)), dim3(256), 0, (hipStream_t)stream,
                     w, packed, T, transpose_flip);
  AS_CHECK_LAUNCH("as_conv32_pack_weights");
  return AS_OK;
}

// All of a step's weight packings in ONE launch: the job table lives in device memory (the weights sit in a
// flat arena and the packed buffers are persistent, so the table is built once).
// transpose_flip doubles as the KIND of a job (round 3: every weight-derived buffer of a step comes out of this one launch):
//   0 / 1                         as_conv32_pack_weights (forward / data-gradient orientation)
//   AS_PACK_S2_DGRAD (2)          as_conv32_dgrad_s2_pack: 5x5 stride-2 data gradient, the 25 taps in parity-phase-major order
//   AS_PACK_CONV4 + Cin (16 + c)  as_conv4_pack_weights for Cin input channels, taps = T
//   AS_PACK_MIRROR_TAP / _CH      as_mirror_taps_ch0's by_tap / by_channel layouts of a [32][4][3][3] weight
//   AS_PACK_WINO / _T             as_conv32_wino_pack_weights (forward / data-gradient filter), taps = 16 transformed filters
__device__ inline int s2_phase_major_tap(int t) {          // the t-th tap of the order (py, px, j, l) of as_conv32_dgrad_s2
  int n = 0;
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px)
      for (int j = py; j < 5; j += 2)
        for (int l = px; l < 5; l += 2) { if (n == t) return j * 5 + l; ++n; }
  return 0;
}

// Transformed filters of the minimal-filtering algorithm F(2x2, 3x3): U[4r+c] = (G g G^T)[r][c] of a [32][32][3][3] weight,
// G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1], in the B-fragment layout of the standard packing: [16][4 q][64 lanes][4 e],
// element = U[ci = 16h + 4q + e][co = li].  transposed: the data gradient's filter (channels swapped, taps mirrored).
__device__ inline float wino_g_row(int r, float x0, float x1, float x2) {
  return r == 0 ? x0 : (r == 3 ? x2 : (r == 1 ? 0.5f * ((x0 + x2) + x1) : 0.5f * ((x0 + x2) - x1)));
}
__device__ inline void wino_pack_one(const float* __restrict__ w, float* __restrict__ packed, int idx, bool transposed) {
  if (idx >= 16 * 1024) return;
  const int e = idx & 3, j = (idx >> 2) & 31, h = (idx >> 7) & 1, q = (idx >> 8) & 3, kk = idx >> 10;
  const int k = 16 * h + 4 * q + e, r = kk >> 2, c = kk & 3;
  const float* g = transposed ? w + ((long)k * 32 + j) * 9 : w + ((long)j * 32 + k) * 9;
  float col[3];                                              // (g G^T)[a][c] for the three filter rows a
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float* ga = g + 3 * (transposed ? 2 - a : a);
    col[a] = transposed ? wino_g_row(c, ga[2], ga[1], ga[0]) : wino_g_row(c, ga[0], ga[1], ga[2]);
  }
  packed[idx] = wino_g_row(r, col[0], col[1], col[2]);
}
__global__ void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ packed, int transposed) {
  wino_pack_one(w, packed, blockIdx.x * blockDim.x + threadIdx.x, transposed != 0);
}

__global__ void pack_weights_batch_kernel(const as_pack_job* __restrict__ jobs) {
  const as_pack_job job = jobs[blockIdx.y];
  const int T = job.taps, kind = job.transpose_flip;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (kind >= AS_PACK_CONV4 && kind < AS_PACK_CONV4 + 8) {
    const int Cin = kind - AS_PACK_CONV4;
    if (idx >= T * 128) return;
    const int co = idx & 31, h = (idx >> 5) & 1, j = (idx >> 6) & 1, t = idx >> 7;
    const int c = 2 * j + h;
    job.packed[idx] = c < Cin ? job.w[((long)co * Cin + c) * T + t] : 0.f;
    return;
  }
  if (kind == AS_PACK_MIRROR_TAP || kind == AS_PACK_MIRROR_CH) {
    if (idx >= 288) return;
    const int c = idx / 9, t = idx - 9 * c;
    const float v = job.w[((long)c * 4) * 9 + (8 - t)];
    job.packed[kind == AS_PACK_MIRROR_TAP ? t * 32 + c : c * 9 + t] = v;
    return;
  }
  if (kind == AS_PACK_WINO || kind == AS_PACK_WINO_T) { wino_pack_one(job.w, job.packed, idx, kind == AS_PACK_WINO_T); return; }
  if (idx >= T * 1024) return;
  const int e = idx & 3, j = (idx >> 2) & 31, h = (idx >> 7) & 1, q = (idx >> 8) & 3, t = idx >> 10;
  const int k = 16 * h + 4 * q + e;
  if (kind == AS_PACK_S2_DGRAD) job.packed[idx] = job.w[((long)k * 32 + j) * T + s2_phase_major_tap(t)];
  else job.packed[idx] = kind ? job.w[((long)k * 32 + j) * T + (T - 1 - t)] : job.w[((long)j * 32 + k) * T + t];
}

extern "C" int as_conv32_pack_weights_batch(const as_pack_job* jobs, int njobs, int max_taps, void* stream) {
  AS_CHECK_ARG(jobs && njobs >= 1 && njobs <= 65535, "as_conv32_pack_weights_batch: bad job table");
  AS_CHECK_ARG(max_taps >= 1 && max_taps <= AS_MAX_TAPS, "as_conv32_pack_weights_batch: %d taps unsupported", max_taps);
  hipLaunchKernelGGL(pack_weights_batch_kernel, dim3(as_div_up(max_taps * 1024, 256), njobs), dim3(256), 0,
                     (hipStream_t)stream, jobs);
  AS_CHECK_LAUNCH("as_conv32_pack_weights_batch");
  return AS_OK;
}

extern "C" int as_conv32_wino_pack_weights(const float* w, float* packed, int transposed, void* stream) {
  AS_CHECK_ARG(w && packed, "as_conv32_wino_pack_weights: null pointer");
  hipLaunchKernelGGL(wino_pack_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, w, packed, transposed);
  AS_CHECK_LAUNCH("as_conv32_wino_pack_weights");
  return AS_OK;
}

// Weight gradient fused with stage 3 of the layer's BatchNorm backward (the gradient operand arrives as g_a).
extern "C" int as_conv32_wgrad_bnapply_ok(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s) {
  if (!as_pcl_ok(gin) || !as_pcl_ok(gout) || !s) return AS_ERR_ARG;
  return wgrad_lds_applicable(gin, gout, s) && conv32_wgrad_bnapply_ok(gout) ? 1 : 0;
}

extern "C" int as_conv32_wgrad_bnapply(const float* x, const as_pcl* gin, const float* g_a, const float* z,
                                       const as_pcl* gout, const as_conv_shape* s, const float* scale,
                                       const float* shift, const float* mean, const float* coef, float slope,
                                       float* g_z, float* dW, float* db, int accumulate, float* workspace, void* stream) {
  if (int e = check_conv(gin, gout, s, "as_conv32_wgrad_bnapply")) return e;
  AS_CHECK_ARG(x && g_a && z && scale && shift && mean && coef && g_z && dW && workspace, "as_conv32_wgrad_bnapply: null pointer");
  AS_CHECK_ARG(wgrad_lds_applicable(gin, gout, s) && conv32_wgrad_bnapply_ok(gout),
               "as_conv32_wgrad_bnapply: configuration not supported (as_conv32_wgrad_bnapply_ok() == 0)");
  const int T = 9;
  const int slabs = conv32_wgrad_lds_slabs(gout);
  float* partial_db = workspace + (int64_t)slabs * T * 1024;
  WgradBnApply bn = {z, scale, shift, mean, coef, g_z, slope};
  hipStream_t st = (hipStream_t)stream;
  as_prof_mark(3, st, 1, 0.0);
  if (int e = conv32_wgrad_lds_launch(x, gin, g_a, gout, s, workspace, partial_db, &bn, stream)) return e;
  as_prof_mark(3, st, 0, 2.0 * (double)gout->B * gout->D * gout->H * gout->W * 1024.0 * T);
  wgrad_reduce(st, workspace, partial_db, slabs, T, dW, db, accumulate);
  AS_CHECK_LAUNCH("as_conv32_wgrad_bnapply(reduce)");
  return AS_OK;
}

// Whole backward of a full-resolution layer in one launch (conv32_bwd.hip): stage 3 of its BatchNorm backward, data gradient
// + skip connection, weight / bias gradient, stage 1 of the next BatchNorm backward.
extern "C" int as_conv32_bwd_fused_ok(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s) {
  if (!as_pcl_ok(gin) || !as_pcl_ok(gout) || !s) return AS_ERR_ARG;
  return conv32_bwd_fused_applicable(gin, gout, s) ? 1 : 0;
}
extern "C" int as_conv32_bwd_fused_parts(void) { return conv32_bwd_fused_slabs(); }
extern "C" int64_t as_conv32_bwd_fused_workspace(void) { return (int64_t)conv32_bwd_fused_slabs() * (9 * 1024 + 32); }

extern "C" int as_conv32_bwd_fused(const float* x, const as_pcl* gin, const float* g_a, const float* z, const as_pcl* gout,
                                   const as_conv_shape* s, const float* packed_wt, const float* scale, const float* shift,
                                   const float* mean, const float* coef, float slope, const float* next_z,
                                   const float* next_scale, const float* next_shift, const float* next_mean, float* g_x,
                                   float* dW, float* db, int accumulate, float* next_bn_workspace, float* workspace,
                                   void* stream) {
  if (int e = check_conv(gin, gout, s, "as_conv32_bwd_fused")) return e;
  AS_CHECK_ARG(x && g_a && z && packed_wt && scale && shift && mean && coef && next_z && next_scale && next_shift && next_mean &&
               g_x && dW && next_bn_workspace && workspace, "as_conv32_bwd_fused: null pointer");
  AS_CHECK_ARG(conv32_bwd_fused_applicable(gin, gout, s), "as_conv32_bwd_fused: configuration not supported (as_conv32_bwd_fused_ok() == 0)");
  AS_CHECK_ARG(((uintptr_t)next_bn_workspace & 7) == 0, "as_conv32_bwd_fused: the BatchNorm workspace must be 8-byte aligned");
  AS_CHECK_ARG(g_x != g_a && g_x != x && g_x != z, "as_conv32_bwd_fused: g_x must not alias an input");
  const int T = 9;
  const int slabs = conv32_bwd_fused_slabs();
  float* partial_db = workspace + (int64_t)slabs * T * 1024;
  hipStream_t st = (hipStream_t)stream;
  as_prof_mark(AS_PROF_BWD_FUSED, st, 1, 0.0);
  if (int e = conv32_bwd_fused_launch(x, g_a, z, gout, s, packed_wt, scale, shift, mean, coef, slope, next_z, next_scale,
                                      next_shift, next_mean, g_x, workspace, partial_db,
                                      reinterpret_cast<double*>(next_bn_workspace), stream)) return e;
  as_prof_mark(AS_PROF_BWD_FUSED, st, 0, 2.0 * 2.0 * (double)gout->B * gout->H * gout->W * 1024.0 * T);   // dgrad + wgrad
  AS_CHECK_LAUNCH("as_conv32_bwd_fused");
  wgrad_reduce(st, workspace, partial_db, slabs, T, dW, db, accumulate);
  AS_CHECK_LAUNCH("as_conv32_bwd_fused(reduce)");
  return AS_OK;
}

// Training forward of a full-resolution layer whose operand is the PREVIOUS layer's output, formed on the way in from that
// layer's pre-activation, BatchNorm affine and skip input (csrc/conv32_act.hip).
extern "C" int as_conv32_act_ok(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s) {
  if (!as_pcl_ok(gin) || !as_pcl_ok(gout) || !s) return AS_ERR_ARG;
  return conv32_act_applicable(gin, gout, s) ? 1 : 0;
}
extern "C" int as_conv32_act_parts(void) { return conv32_act_parts(); }

extern "C" int as_conv32_act_fwd(const float* z_prev, const float* a_prevprev, const float* in_scale, const float* in_shift,
                                 float* a_out, const as_pcl* gin, const float* packed_w, const float* bias, float slope,
                                 float* z, const as_pcl* gout, const as_conv_shape* s, float* stat_mean, float* stat_m2,
                                 float* stat_cnt, void* stream) {
  if (int e = check_conv(gin, gout, s, "as_conv32_act_fwd")) return e;
  AS_CHECK_ARG(z_prev && in_scale && in_shift && a_out && packed_w && z, "as_conv32_act_fwd: null pointer");
  AS_CHECK_ARG((stat_mean != nullptr) == (stat_m2 != nullptr) && (stat_mean != nullptr) == (stat_cnt != nullptr),
               "as_conv32_act_fwd: pass all three moment arrays or none");
  AS_CHECK_ARG(conv32_act_applicable(gin, gout, s), "as_conv32_act_fwd: configuration not supported (as_conv32_act_ok() == 0)");
  AS_CHECK_ARG(slope > 0.f && slope < 1.f, "as_conv32_act_fwd: slope must lie in (0, 1)");
  AS_CHECK_ARG(a_out != z_prev && a_out != a_prevprev && z != z_prev && z != a_out && z != a_prevprev,
               "as_conv32_act_fwd: outputs must not alias inputs or each other");
  hipStream_t st = (hipStream_t)stream;
  as_prof_mark(AS_PROF_CONV_ACT, st, 1, 0.0);
  if (int e = conv32_act_launch(z_prev, a_prevprev, in_scale, in_shift, a_out, gout, s, packed_w, bias, slope, z, stat_mean,
                                stat_m2, stat_cnt, stream)) return e;
  as_prof_mark(AS_PROF_CONV_ACT, st, 0, 2.0 * (double)gout->B * gout->H * gout->W * 1024.0 * 9);
  AS_CHECK_LAUNCH("as_conv32_act_fwd");
  return AS_OK;
}

// The same layer by the minimal-filtering algorithm F(2x2, 3x3) (csrc/conv32_wino.hip): 4 matrix products per output pixel
// instead of 9.  wino_w: as_conv32_wino_pack_weights (or a batch job of kind AS_PACK_WINO).
extern "C" int as_conv32_wino_ok(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s) {
  if (!as_pcl_ok(gin) || !as_pcl_ok(gout) || !s) return AS_ERR_ARG;
  return conv32_wino_applicable(gin, gout, s) ? 1 : 0;
}
extern "C" int as_conv32_wino_parts(void) { return conv32_wino_parts(); }

extern "C" int as_conv32_wino_fwd(const float* z_prev, const float* a_prevprev, const float* in_scale, const float* in_shift,
                                  float* a_out, const as_pcl* gin, const float* wino_w, const float* bias, float slope,
                                  float* z, const as_pcl* gout, const as_conv_shape* s, float* stat_mean, float* stat_m2,
                                  float* stat_cnt, void* stream) {
  if (int e = check_conv(gin, gout, s, "as_conv32_wino_fwd")) return e;
  AS_CHECK_ARG(z_prev && in_scale && in_shift && a_out && wino_w && z, "as_conv32_wino_fwd: null pointer");
  AS_CHECK_ARG((stat_mean != nullptr) == (stat_m2 != nullptr) && (stat_mean != nullptr) == (stat_cnt != nullptr),
               "as_conv32_wino_fwd: pass all three moment arrays or none");
  AS_CHECK_ARG(conv32_wino_applicable(gin, gout, s), "as_conv32_wino_fwd: configuration not supported (as_conv32_wino_ok() == 0)");
  AS_CHECK_ARG(slope > 0.f && slope < 1.f, "as_conv32_wino_fwd: slope must lie in (0, 1)");
  AS_CHECK_ARG(a_out != z_prev && a_out != a_prevprev && z != z_prev && z != a_out && z != a_prevprev,
               "as_conv32_wino_fwd: outputs must not alias inputs or each other");
  hipStream_t st = (hipStream_t)stream;
  as_prof_mark(AS_PROF_WINO_FWD, st, 1, 0.0);
  if (int e = conv32_wino_launch(z_prev, a_prevprev, in_scale, in_shift, a_out, gout, s, wino_w, bias, slope, z, stat_mean,
                                 stat_m2, stat_cnt, stream)) return e;
  // (ALGORITHMIC flops: the direct form's 9 taps — what the layer computes, not the 4 products per pixel it executes)
  as_prof_mark(AS_PROF_WINO_FWD, st, 0, 2.0 * (double)gout->B * gout->H * gout->W * 1024.0 * 9);
  AS_CHECK_LAUNCH("as_conv32_wino_fwd");
  return AS_OK;
}

// Eval-mode BasicBlock by minimal filtering: out = lrelu((conv(x) + bias) * scale + shift) (+ x).  One read of x (the skip
// connection comes out of the staged rows), one write.
extern "C" int as_conv32_wino_eval(const float* x, const as_pcl* g, const as_conv_shape* s, const float* wino_w, const float* bias,
                                   const float* scale, const float* shift, float slope, int residual, float* out, void* stream) {
  if (int e = check_conv(g, g, s, "as_conv32_wino_eval")) return e;
  AS_CHECK_ARG(x && wino_w && scale && shift && out, "as_conv32_wino_eval: null pointer");
  AS_CHECK_ARG(conv32_wino_applicable(g, g, s), "as_conv32_wino_eval: configuration not supported (as_conv32_wino_ok() == 0)");
  AS_CHECK_ARG(slope > 0.f && slope < 1.f, "as_conv32_wino_eval: slope must lie in (0, 1)");
  AS_CHECK_ARG(out != x, "as_conv32_wino_eval: out must not alias x");
  hipStream_t st = (hipStream_t)stream;
  as_prof_mark(AS_PROF_WINO_FWD, st, 1, 0.0);
  if (int e = conv32_wino_eval_launch(x, g, s, wino_w, bias, scale, shift, slope, residual, out, stream)) return e;
  as_prof_mark(AS_PROF_WINO_FWD, st, 0, 2.0 * (double)g->B * g->H * g->W * 1024.0 * 9);
  AS_CHECK_LAUNCH("as_conv32_wino_eval");
  return AS_OK;
}

// Backward of the same layer by minimal filtering, two launches: data gradient F(2x2, 3x3) with stage 3 of the BatchNorm
// backward on the way in (g_z written once), skip connection and next-BatchNorm sums in the epilogue; then weight / bias
// gradient F(3x3, 2x2) from x and g_z.  Arguments as as_conv32_bwd_fused, plus g_z (a PCL buffer of the layer's geometry).
extern "C" int as_conv32_wino_bwd_parts(void) { return conv32_wino_dgrad_parts(); }
// Which data-gradient kernel as_conv32_wino_bwd_data launches: 2 (default) = conv32_wino_dgrad.hip where it is the faster one
// (dilation 1, 2, 4), 1 = conv32_wino.hip MODE 2 everywhere, 3 = conv32_wino_dgrad.hip everywhere (its dilation-8 instantiation
// is slower than generation 1 and only here for the parity tests).  All write the same bits; returns the previous setting.
static int g_wino_dgrad_generation = 2;
extern "C" int as_conv32_wino_bwd_generation(int generation) {
  const int prev = g_wino_dgrad_generation;
  if (generation >= 1 && generation <= 3) g_wino_dgrad_generation = generation;
  return prev;
}
extern "C" int64_t as_conv32_wino_bwd_workspace(void) { return (int64_t)conv32_wino_wgrad_slabs() * (9 * 1024 + 32); }

// The two halves separately (a caller may run the weight gradient on another stream: nothing but the step's final slab
// reduction waits for it) ...
extern "C" int as_conv32_wino_bwd_data(const float* g_a, const float* z, const as_pcl* g, const as_conv_shape* s,
                                       const float* wino_wt, const float* scale, const float* shift, const float* mean,
                                       const float* coef, float slope, const float* next_z, const float* next_scale,
                                       const float* next_shift, const float* next_mean, float* g_z, float* g_x,
                                       float* next_bn_workspace, void* stream) {
  if (int e = check_conv(g, g, s, "as_conv32_wino_bwd_data")) return e;
  AS_CHECK_ARG(g_a && z && wino_wt && scale && shift && mean && coef && next_z && next_scale && next_shift && next_mean && g_z &&
               g_x && next_bn_workspace, "as_conv32_wino_bwd_data: null pointer");
  AS_CHECK_ARG(conv32_wino_applicable(g, g, s), "as_conv32_wino_bwd_data: configuration not supported (as_conv32_wino_ok() == 0)");
  AS_CHECK_ARG(((uintptr_t)next_bn_workspace & 7) == 0, "as_conv32_wino_bwd_data: the BatchNorm workspace must be 8-byte aligned");
  AS_CHECK_ARG(slope > 0.f && slope < 1.f, "as_conv32_wino_bwd_data: slope must lie in (0, 1)");
  AS_CHECK_ARG(g_x != g_a && g_x != z && g_z != g_a && g_z != z && g_z != g_x && g_z != next_z && g_x != next_z,
               "as_conv32_wino_bwd_data: outputs must not alias inputs or each other");
  hipStream_t st = (hipStream_t)stream;
  // (ALGORITHMIC flops of the gradient in its direct form, as as_conv32_bwd_fused counts them)
  as_prof_mark(AS_PROF_WINO_DGRAD, st, 1, 0.0);
  if (conv32_wino_dgrad2_parts() != conv32_wino_dgrad_parts()) { as_set_error("as_conv32_wino_bwd_data: partial counts differ"); return AS_ERR_ARG; }
  // (generation 2 wins for dilation 1, 2, 4 — 286 / 269 / 268 us against 343 / 300 / 287 at 4 pairs — and loses at 8, where
  //  only one of its three raw-row slots fits the LDS: 335 against 288; profiles/r04_b_dgrad_generations.txt)
  const bool gen2 = g_wino_dgrad_generation == 3 || (g_wino_dgrad_generation == 2 && s->dil <= 4);
  if (int e = (gen2 ? conv32_wino_dgrad2_launch : conv32_wino_dgrad_launch)(
          g_a, z, g, s, wino_wt, scale, shift, mean, coef, slope, next_z, next_scale, next_shift, next_mean, g_z, g_x,
          reinterpret_cast<double*>(next_bn_workspace), stream)) return e;
  as_prof_mark(AS_PROF_WINO_DGRAD, st, 0, 2.0 * (double)g->B * g->H * g->W * 1024.0 * 9);
  AS_CHECK_LAUNCH("as_conv32_wino_bwd_data");
  return AS_OK;
}

extern "C" int as_conv32_wino_bwd_filter(const float* x, const float* g_z, const as_pcl* g, const as_conv_shape* s, float* dW,
                                         float* db, int accumulate, float* workspace, void* stream) {
  if (int e = check_conv(g, g, s, "as_conv32_wino_bwd_filter")) return e;
  AS_CHECK_ARG(x && g_z && dW && workspace, "as_conv32_wino_bwd_filter: null pointer");
  AS_CHECK_ARG(conv32_wino_applicable(g, g, s), "as_conv32_wino_bwd_filter: configuration not supported (as_conv32_wino_ok() == 0)");
  const int T = 9;
  const int slabs = conv32_wino_wgrad_slabs();
  float* partial_db = workspace + (int64_t)slabs * T * 1024;
  hipStream_t st = (hipStream_t)stream;
  as_prof_mark(AS_PROF_WINO_WGRAD, st, 1, 0.0);
  if (int e = conv32_wino_wgrad_launch(x, g_z, g, s, workspace, partial_db, stream)) return e;
  as_prof_mark(AS_PROF_WINO_WGRAD, st, 0, 2.0 * (double)g->B * g->H * g->W * 1024.0 * T);
  AS_CHECK_LAUNCH("as_conv32_wino_bwd_filter");
  wgrad_reduce(st, workspace, partial_db, slabs, T, dW, db, accumulate);
  AS_CHECK_LAUNCH("as_conv32_wino_bwd_filter(reduce)");
  return AS_OK;
}

// ... and both on one stream
extern "C" int as_conv32_wino_bwd(const float* x, const as_pcl* gin, const float* g_a, const float* z, const as_pcl* gout,
                                  const as_conv_shape* s, const float* wino_wt, const float* scale, const float* shift,
                                  const float* mean, const float* coef, float slope, const float* next_z,
                                  const float* next_scale, const float* next_shift, const float* next_mean, float* g_z,
                                  float* g_x, float* dW, float* db, int accumulate, float* next_bn_workspace,
                                  float* workspace, void* stream) {
  if (int e = check_conv(gin, gout, s, "as_conv32_wino_bwd")) return e;
  AS_CHECK_ARG(x && dW && workspace, "as_conv32_wino_bwd: null pointer");
  AS_CHECK_ARG(g_x != x && g_z != x, "as_conv32_wino_bwd: outputs must not alias inputs or each other");
  if (int e = as_conv32_wino_bwd_data(g_a, z, gout, s, wino_wt, scale, shift, mean, coef, slope, next_z, next_scale, next_shift,
                                      next_mean, g_z, g_x, next_bn_workspace, stream)) return e;
  return as_conv32_wino_bwd_filter(x, g_z, gout, s, dW, db, accumulate, workspace, stream);
}

// The same backward in ONE launch (csrc/conv32_wino_bwd.hip): data gradient and weight gradient side by side in an 8-wave
// workgroup per CU; g_z = stage 3 of the BatchNorm backward never leaves the chip (5 tensor passes instead of 7).  Arguments of
// as_conv32_wino_bwd without g_z; g_x bit-identical to the two-launch form.
extern "C" int as_conv32_wino_bwd_fused_parts(void) { return conv32_wino_bwd_fused_parts(); }
extern "C" int64_t as_conv32_wino_bwd_fused_workspace(void) { return (int64_t)conv32_wino_bwd_fused_parts() * (9 * 1024 + 32); }
extern "C" int as_conv32_wino_bwd_fused(const float* x, const as_pcl* gin, const float* g_a, const float* z, const as_pcl* gout,
                                        const as_conv_shape* s, const float* wino_wt, const float* scale, const float* shift,
                                        const float* mean, const float* coef, float slope, const float* next_z,
                                        const float* next_scale, const float* next_shift, const float* next_mean, float* g_x,
                                        float* dW, float* db, int accumulate, float* next_bn_workspace, float* workspace,
                                        void* stream) {
  if (int e = check_conv(gin, gout, s, "as_conv32_wino_bwd_fused")) return e;
  AS_CHECK_ARG(x && g_a && z && wino_wt && scale && shift && mean && coef && next_z && next_scale && next_shift && next_mean &&
               g_x && dW && next_bn_workspace && workspace, "as_conv32_wino_bwd_fused: null pointer");
  AS_CHECK_ARG(conv32_wino_applicable(gin, gout, s), "as_conv32_wino_bwd_fused: configuration not supported (as_conv32_wino_ok() == 0)");
  AS_CHECK_ARG(((uintptr_t)next_bn_workspace & 7) == 0, "as_conv32_wino_bwd_fused: the BatchNorm workspace must be 8-byte aligned");
  AS_CHECK_ARG(slope > 0.f && slope < 1.f, "as_conv32_wino_bwd_fused: slope must lie in (0, 1)");
  AS_CHECK_ARG(g_x != g_a && g_x != z && g_x != x && g_x != next_z, "as_conv32_wino_bwd_fused: g_x must not alias an input");
  const int T = 9;
  const int slabs = conv32_wino_bwd_fused_parts();
  float* partial_db = workspace + (int64_t)slabs * T * 1024;
  hipStream_t st = (hipStream_t)stream;
  as_prof_mark(AS_PROF_WINO_BWD, st, 1, 0.0);
  if (int e = conv32_wino_bwd_fused_launch(x, g_a, z, gout, s, wino_wt, scale, shift, mean, coef, slope, next_z, next_scale, next_shift,
                                           next_mean, g_x, workspace, partial_db, reinterpret_cast<double*>(next_bn_workspace),
                                           stream)) return e;
  as_prof_mark(AS_PROF_WINO_BWD, st, 0, 2.0 * 2.0 * (double)gout->B * gout->H * gout->W * 1024.0 * T);   // dgrad + wgrad, direct form
  AS_CHECK_LAUNCH("as_conv32_wino_bwd_fused");
  wgrad_reduce(st, workspace, partial_db, slabs, T, dW, db, accumulate);
  AS_CHECK_LAUNCH("as_conv32_wino_bwd_fused(reduce)");
  return AS_OK;
}

// Host-side view of conv3d_wgrad_lds_kernel's work assignment (csrc/conv3d_lds.h: conv3d_wgrad_assign — the very function the
// kernel calls): for workgroup `block` of a launch over `ntiles` tiles in `nchunks` chunks, its chunk, kd and extra tile (-1:
// none).  Returns 1 for a working block, 0 for a padding block.  No GPU involved: the CPU suite walks it.
extern "C" int as_conv3d_wgrad_lds_assignment(int ntiles, int nchunks, int block, int* chunk, int* kd, int* extra_tile) {
  AS_CHECK_ARG(chunk && kd && extra_tile && ntiles >= 1 && nchunks >= 1 && nchunks <= ntiles && block >= 0,
               "as_conv3d_wgrad_lds_assignment: bad arguments");
  return conv3d_wgrad_assign(block, ntiles, nchunks, chunk, kd, extra_tile) ? 1 : 0;
}

// Convolution (data gradient) fused with stage 1 of the BatchNorm backward that consumes its output.
extern "C" int as_conv32_bnbwd_parts(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s) {
  if (!as_pcl_ok(gin) || !as_pcl_ok(gout) || !s) return AS_ERR_ARG;
  return conv32_lds_applicable(gin, gout, s) ? conv32_lds_grid(gout) : 0;
}

extern "C" int as_conv32_fwd_bnbwd(const float* x, const as_pcl* gin, const float* packed_w, float* z, const as_pcl* gout,
                                   const as_conv_shape* s, const float* residual, const float* bn_z,
                                   const float* bn_scale, const float* bn_shift, const float* bn_mean, float slope,
                                   float* bn_workspace, void* stream) {
  if (int e = check_conv(gin, gout, s, "as_conv32_fwd_bnbwd")) return e;
  AS_CHECK_ARG(x && packed_w && z && bn_z && bn_scale && bn_shift && bn_mean && bn_workspace, "as_conv32_fwd_bnbwd: null pointer");
  AS_CHECK_ARG(conv32_lds_applicable(gin, gout, s), "as_conv32_fwd_bnbwd: configuration not supported (as_conv32_bnbwd_parts() == 0)");
  AS_CHECK_ARG(((uintptr_t)bn_workspace & 7) == 0, "as_conv32_fwd_bnbwd: workspace must be 8-byte aligned");
  return conv32_lds_launch(x, gin, packed_w, nullptr, z, gout, s, 0, nullptr, nullptr, slope, residual, nullptr, nullptr,
                           nullptr, bn_z, bn_scale, bn_shift, bn_mean, reinterpret_cast<double*>(bn_workspace), stream);
}

extern "C" int as_conv32_num_blocks(const as_pcl* gout) {
  if (!as_pcl_ok(gout)) return AS_ERR_ARG;
  const int64_t M = (int64_t)gout->B * gout->D * gout->H * gout->W;
  return as_div_up(M, 128);
}

// Number of BatchNorm partials as_conv32_fwd writes for this configuration.
extern "C" int as_conv32_stat_parts(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s) {
  if (!as_pcl_ok(gin) || !as_pcl_ok(gout) || !s) return AS_ERR_ARG;
  if (conv32_lds_applicable(gin, gout, s)) return conv32_lds_grid(gout);
  if (conv3d_lds_applicable(gin, gout, s)) return conv3d_lds_grid(gout);
  const int64_t M = (int64_t)gout->B * gout->D * gout->H * gout->W;
  if (conv32_splitk_applies(s->kd * s->kh * s->kw, M)) return as_div_up(M, 32);     // one partial per 32-voxel tile
  return as_conv32_num_blocks(gout);
}

extern "C" int as_conv32_fwd(const float* x, const as_pcl* gin, const float* packed_w, const float* bias,
                             float* z, const as_pcl* gout, const as_conv_shape* s,
                             int epilogue, const float* ep_scale, const float* ep_shift, float slope,
                             const float* residual, float* stat_mean, float* stat_m2, float* stat_cnt, void* stream) {
  if (int e = check_conv(gin, gout, s, "as_conv32_fwd")) return e;
  AS_CHECK_ARG(x && packed_w && z, "as_conv32_fwd: null pointer");
  AS_CHECK_ARG(epilogue_args_ok(epilogue, ep_scale, ep_shift, stat_mean, stat_m2, stat_cnt), "as_conv32_fwd: bad epilogue arguments");
  if (conv32_lds_applicable(gin, gout, s))
    return conv32_lds_launch(x, gin, packed_w, bias, z, gout, s, epilogue, ep_scale, ep_shift, slope, residual,
                             stat_mean, stat_m2, stat_cnt, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
  if (conv3d_lds_applicable(gin, gout, s))
    return conv3d_lds_launch(x, gin, packed_w, bias, z, gout, epilogue, ep_scale, ep_shift, slope, residual,
                             stat_mean, stat_m2, stat_cnt, stream);
  const int64_t M_all = (int64_t)gout->B * gout->D * gout->H * gout->W;
  if (epilogue == 0 && residual == nullptr && stat_mean == nullptr && g_conv32_s2 && conv32_s2_fwd_applicable(gin, gout, s) &&
      !conv32_splitk_applies(25, M_all)) {
    // the strided head of the feature towers on maps that fill the chip: coalesced row staging (csrc/conv32_s2.hip)
    hipStream_t st2 = (hipStream_t)stream;
    as_prof_mark(0, st2, 1, 0.0);
    if (int e = conv32_s2_fwd_launch(x, gin, packed_w, bias, z, gout, stream)) return e;
    as_prof_mark(0, st2, 0, 2.0 * (double)M_all * 1024.0 * 25.0);
    AS_CHECK_LAUNCH("as_conv32_fwd(5x5 stride 2)");
    return AS_OK;
  }
  ConvArgs a;
  a.x = x; a.wp = packed_w;
  a.ep.bias = bias; a.ep.z = z; a.ep.ep_scale = ep_scale; a.ep.ep_shift = ep_shift; a.ep.residual = residual;
  a.ep.stat_mean = epilogue == 0 ? stat_mean : nullptr; a.ep.stat_m2 = epilogue == 0 ? stat_m2 : nullptr;
  a.ep.stat_cnt = epilogue == 0 ? stat_cnt : nullptr;
  a.ep.epilogue = epilogue; a.ep.slope = slope;
  a.gin = as_make_dev(gin); a.gout = as_make_dev(gout);
  const int64_t M = (int64_t)gout->B * gout->D * gout->H * gout->W;
  a.M = (int)M; a.ntaps = s->kd * s->kh * s->kw;
  a.map.Hl = gout->H; a.map.Wl = gout->W; a.map.in_stride = s->stride; a.map.out_stride = 1; a.map.out_oy = 0; a.map.out_ox = 0;
  if (int e = fill_taps(gin, s, a.tap_off, "as_conv32_fwd")) return e;
  as_prof_mark(0, (hipStream_t)stream, 1, 0.0);
  hipStream_t st = (hipStream_t)stream;
  if (int e = launch_conv32(a, st, "as_conv32_fwd")) return e;
  as_prof_mark(0, (hipStream_t)stream, 0, 2.0 * (double)M * 1024.0 * a.ntaps);
  AS_CHECK_LAUNCH("as_conv32_fwd");
  return AS_OK;
}

// Rows are cut into segments only while whole rows would leave SIMDs idle (rows x tap groups < 1024 waves): every
// extra segment is an extra partial slab to write and reduce.
static int wgrad_segments(const as_pcl* gout, const as_conv_shape* s, int* seg_steps) {
  const int T = s->kd * s->kh * s->kw;
  const int groups = T / ((T % 3 == 0) ? 3 : (T % 5 == 0 ? 5 : 1));
  const int nsteps = (gout->W + 1) >> 1;
  const long waves = (long)gout->B * gout->D * gout->H * groups;
  int nseg = 1;
  if (waves < 1024) {
    nseg = (int)((2048 + waves - 1) / waves);
    const int most = (nsteps + WG_SEG_STEPS - 1) / WG_SEG_STEPS;         // segments of at least WG_SEG_STEPS pairs
    if (nseg > most) nseg = most;
    if (nseg < 1) nseg = 1;
  }
  int steps = (nsteps + nseg - 1) / nseg;
  steps = (steps + 7) / 8 * 8;
  *seg_steps = steps;
  return (nsteps + steps - 1) / steps;
}
static int wgrad_plan(const as_pcl* gout, const as_conv_shape* s, int* tg, int* rows_per_chunk, int* nchunks) {
  const int T = s->kd * s->kh * s->kw;
  *tg = (T % 3 == 0) ? 3 : (T % 5 == 0 ? 5 : 1);
  int seg_steps;
  const int rows = gout->B * gout->D * gout->H * wgrad_segments(gout, s, &seg_steps);       // work units (row segments)
  // aim at ~4 waves per SIMD over the chip (256 CUs x 4 SIMDs), at least 4 units (one per
  // wave) per chunk, and cap the slab count so the partial buffer stays a few MB.
  const int groups = T / *tg;
  int want_chunks = (4096 + groups * 4 - 1) / (groups * 4);
  if (want_chunks > 512) want_chunks = 512;
  int rpc = (rows + want_chunks - 1) / want_chunks;
  if (rpc < 4) rpc = 4;
  *rows_per_chunk = rpc;
  *nchunks = (rows + rpc - 1) / rpc;
  return T;
}

static bool wgrad_lds_applicable(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s) {
  return conv32_lds_applicable(gin, gout, s) && gout->pw >= 8;   // G groups beyond W read 8 zero halo voxels
}

extern "C" int64_t as_conv32_wgrad_workspace(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s) {
  if (!as_pcl_ok(gin) || !as_pcl_ok(gout) || !s) return -1;
  int tg, rpc, nchunks;
  const int T = wgrad_plan(gout, s, &tg, &rpc, &nchunks);
  if (wgrad_lds_applicable(gin, gout, s)) nchunks = conv32_wgrad_lds_slabs(gout);
  if (conv3d_wgrad_lds_applicable(gin, gout, s)) nchunks = conv3d_wgrad_lds_slabs(gout);
  return (int64_t)nchunks * T * 1024 + (int64_t)nchunks * 32;
}

template <int TG>
static void launch_wgrad(const WgradArgs& a, int groups, int nchunks, hipStream_t st) {
  const size_t lds = (size_t)(3 * TG * 16 * 64 + 4 * 32) * sizeof(float);
  const int per_xcd = (nchunks + 7) / 8;
  hipLaunchKernelGGL(conv32_wgrad_kernel<TG>, dim3(8 * per_xcd * groups), dim3(256), lds, st, a);
}

extern "C" int as_conv32_wgrad(const float* x, const as_pcl* gin, const float* gz, const as_pcl* gout,
                               const as_conv_shape* s, float* dW, float* db, int accumulate, float* workspace,
                               void* stream) {
  if (int e = check_conv(gin, gout, s, "as_conv32_wgrad")) return e;
  AS_CHECK_ARG(x && gz && dW && workspace, "as_conv32_wgrad: null pointer");
  int tg, rpc, nchunks;
  const int T = wgrad_plan(gout, s, &tg, &rpc, &nchunks);
  if (wgrad_lds_applicable(gin, gout, s)) {
    const int slabs = conv32_wgrad_lds_slabs(gout);
    float* partial_db = workspace + (int64_t)slabs * T * 1024;
    hipStream_t st = (hipStream_t)stream;
    as_prof_mark(3, st, 1, 0.0);
    if (int e = conv32_wgrad_lds_launch(x, gin, gz, gout, s, workspace, partial_db, nullptr, stream)) return e;
    as_prof_mark(3, st, 0, 2.0 * (double)gout->B * gout->D * gout->H * gout->W * 1024.0 * T);
    wgrad_reduce(st, workspace, partial_db, slabs, T, dW, db, accumulate);
    AS_CHECK_LAUNCH("as_conv32_wgrad(reduce)");
    return AS_OK;
  }
  if (conv3d_wgrad_lds_applicable(gin, gout, s)) {
    const int slabs = conv3d_wgrad_lds_slabs(gout);
    float* partial_db = workspace + (int64_t)slabs * T * 1024;
    hipStream_t st = (hipStream_t)stream;
    as_prof_mark(AS_PROF_WGRAD3D_LDS, st, 1, 0.0);
    if (int e = conv3d_wgrad_lds_launch(x, gin, gz, gout, workspace, partial_db, stream)) return e;
    as_prof_mark(AS_PROF_WGRAD3D_LDS, st, 0, 2.0 * (double)gout->B * gout->D * gout->H * gout->W * 1024.0 * T);
    wgrad_reduce(st, workspace, partial_db, slabs, T, dW, db, accumulate);
    AS_CHECK_LAUNCH("as_conv32_wgrad(reduce)");
    return AS_OK;
  }
  WgradArgs a;
  a.x = x; a.gz = gz; a.partial = workspace; a.partial_db = workspace + (int64_t)nchunks * T * 1024;
  a.gin = as_make_dev(gin); a.gout = as_make_dev(gout);
  a.nseg = wgrad_segments(gout, s, &a.seg_steps);
  a.rows = gout->B * gout->D * gout->H * a.nseg; a.rows_per_chunk = rpc; a.nchunks = nchunks; a.ntaps = T; a.stride = s->stride;
  if (int e = fill_taps(gin, s, a.tap_off, "as_conv32_wgrad")) return e;
  hipStream_t st = (hipStream_t)stream;
  as_prof_mark(1, st, 1, 0.0);
  if (tg == 3) launch_wgrad<3>(a, T / 3, nchunks, st);
  else if (tg == 5) launch_wgrad<5>(a, T / 5, nchunks, st);
  else launch_wgrad<1>(a, T, nchunks, st);
  as_prof_mark(1, st, 0, 2.0 * (double)gout->B * gout->D * gout->H * gout->W * 1024.0 * T);
  AS_CHECK_LAUNCH("as_conv32_wgrad");
  wgrad_reduce(st, a.partial, a.partial_db, nchunks, T, dW, db, accumulate);
  AS_CHECK_LAUNCH("as_conv32_wgrad(reduce)");
  return AS_OK;
}
