// a4 + a5 + a8 in ONE kernel (with the last aggregation layer's BatchNorm + LeakyReLU applied on the way in):
//   conv3d_alone = nn.Conv3d(32, 1, 3, padding=1)  -> logits            stereo_net.py:162,187
//   softmax(+logits, dim=1), sum_d d * p_d          -> coarse disparity  stereo_net.py:190-192,124-134
//   arg-max index over d, FCS = sorted[0] - mean(sorted[2:])              utils/feature_contrast.py:12-23
// It replaces bn_act_fwd (element-wise pass over the volume), conv32to1_fwd_kernel (27 taps x 128-byte lines per voxel
// through L1) and softargmax_fwd_kernel (three passes over the logits): 46 us of launches at 4 pairs for 12 MB of input.
//
// A workgroup owns 32 consecutive positions of the flattened padded plane (conv3d_lds.hip's position trick) for ALL
// disparities and walks down the disparity axis.  The kernel is latency-bound (12 MB of L2-resident input at 4 pairs), so the
// pipeline is built for overlap, not for bytes: every thread fetches ITS 16-byte chunks of plane q+2 into registers (plain
// coalesced loads the compiler tracks: two planes in flight per thread) while plane q — activated in registers on the way,
// lrelu(x * scale + shift), halo voxels left zero by a precomputed bit mask, the activated tensor stored as a by-product —
// goes to one of two LDS slots (source-side XOR swizzle: conflict-free 16-byte reads for any tap offset) and is consumed
// after ONE barrier per plane; 2-3 workgroups per CU cover each other's waits.  (A first version staged the planes by
// LDS-DMA with a three-slot ring: one plane of look-ahead left 2 us of DMA latency exposed per plane, 40 us per launch.)
// N = 1 rules out the matrix cores (31/32 of an MFMA would be wasted), so the 32->1 convolution is "project, then shift":
// every staged plane is projected ONCE onto the three kd slices of the kernel,
//   P_kd[q][p] = sum_{kh,kw,c} a[q][p + (kh-1)*Wp + (kw-1)][c] * w[c][kd][kh][kw],
// by 8 lanes per position (one 16-byte channel chunk each, the 27 x 4 weights they need in registers, packed fp32 FMAs, a
// three-step wavefront-shuffle reduction), and  logit[d] = bias + P_0[d] + P_1[d+1] + P_2[d+2]  falls out of two carried
// registers.  The D logits of a position stay in LDS; the soft-argmax, the arg-max index (first maximum, as torch.argmax)
// and the FCS are then wavefront-shuffle reductions over those 8 lanes (each holds the disparities d = j, j+8, j+16, j+24).
//   IN 1 / IN 2: x is the last layer's RAW convolution output (IN 2: its BatchNorm still in partials, merged by every
//         workgroup, bn_merge.h); IN 0: x is already activated.
#include "as_common.h"
#include "bn_merge.h"

struct TailArgs {
  const float* x;
  const float* in_scale;
  const float* in_shift;
  BnMergeDev in_bn;          // IN 2: the layer's BatchNorm still in partials
  float* a_out;
  const float* w;            // [32][27] (PyTorch conv3d_alone.weight[0])
  const float* bias;         // [1] or null
  float* logits;             // dense [B][D][H][W]
  float* pred;               // dense [B][H][W]
  int32_t* argmax;           // or null
  float* fcs;                // or null
  PclDev g;
  int tiles_per_plane, npos, run, slot_bytes;
  unsigned wp_magic;
  float slope;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
#define TAIL_MAXD 32

// NK = chunks per thread and plane (run <= 32 * NK voxels)
// OUT: the activated tensor is written back (IN != 0)
template <int IN, int NK, bool OUT>
__global__ __launch_bounds__(256, NK > 8 ? 1 : 2) void agg_tail_kernel(TailArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // layout: [2 plane slots (first: BatchNorm merge scratch)][logits TAIL_MAXD x 32 floats][weights 27 x 32 floats][256 dump slots]
  char* ring = smem;
  float* lg = reinterpret_cast<float*>(smem + 2 * p.slot_bytes);
  float* sw = lg + TAIL_MAXD * 32;
  char* dump = reinterpret_cast<char*>(sw + 27 * 32) + threadIdx.x * 16;
  const int Wp = p.g.Wp, D = p.g.D;
  const int col = blockIdx.x;
  const int b = col / p.tiles_per_plane, t = col - b * p.tiles_per_plane;
  const int first = p.g.ph * Wp + p.g.pw;
  const int pos0 = min(first + 32 * t, first + p.npos - 32);
  const long plane_vox = (long)Wp * p.g.Hp;
  const int c4 = threadIdx.x & 7, pl = threadIdx.x >> 3;

  // this thread's chunks: channel group c4 of run voxels pl + 32k.  lmask: the voxel exists (inside the run); tmask: it is
  // an interior voxel (gets the activation)
  unsigned lmask = 0u, tmask = 0u;
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int v = pl + 32 * k;
    const int ps = pos0 - Wp - 1 + v;
    const int yy = (int)__umulhi((unsigned)ps, p.wp_magic), xx = ps - yy * Wp;
    const bool in = v < p.run && xx >= p.g.pw && xx < p.g.pw + p.g.W && yy >= p.g.ph && yy < p.g.ph + p.g.H;
    lmask |= (v < p.run ? 1u : 0u) << k;
    tmask |= (in ? 1u : 0u) << k;
  }
  const long run_base = ((long)b * p.g.Dp * plane_vox + (pos0 - Wp - 1)) * 32 + 4 * c4;     // + q * plane_vox * 32
  auto fetch = [&](f32x4 (&r)[NK], int q) {             // padded plane min(q, D+1) -> registers (D+1: the zero halo plane)
    const float* src = p.x + run_base + (long)min(q, D + 1) * plane_vox * 32;
#pragma unroll
    for (int k = 0; k < NK; ++k)       // branch-free (a branch per load makes hipcc drain the whole queue at every join):
      r[k] = *reinterpret_cast<const f32x4*>(src + (long)min(pl + 32 * k, p.run - 1) * 32);   // lanes beyond the run re-read its last voxel
  };
  f32x4 ra[NK], rb[NK];
  fetch(ra, 1);
  fetch(rb, 2);

  f32x4 in_sc = {0.f, 0.f, 0.f, 0.f}, in_sh = {0.f, 0.f, 0.f, 0.f};
  if (IN == 1) {
    in_sc = *reinterpret_cast<const f32x4*>(p.in_scale + 4 * c4);
    in_sh = *reinterpret_cast<const f32x4*>(p.in_shift + 4 * c4);
  }
  if (IN == 2) {            // merged while the first two planes are in flight (before the weights occupy 108 registers);
    const float* tab = bn_merge_partials<16>(p.in_bn, ring, blockIdx.x == 0);      // scratch = the plane slots, still unused
    in_sc = *reinterpret_cast<const f32x4*>(tab + 4 * c4);
    in_sh = *reinterpret_cast<const f32x4*>(tab + 32 + 4 * c4);
  }
  // weights -> LDS (transposed to [tap][channel]) -> this lane's 27 x 4 registers
  for (int i = threadIdx.x; i < 27 * 32; i += 256) { const int tp = i >> 5, c = i & 31; sw[i] = p.w[c * 27 + tp]; }
  const int pos = pos0 + pl;
  const int yp = (int)__umulhi((unsigned)pos, p.wp_magic), xp = pos - yp * Wp;
  const bool interior = xp >= p.g.pw && xp < p.g.pw + p.g.W;          // rows are interior by construction
  const int y = yp - p.g.ph, x = xp - p.g.pw;
  __syncthreads();                                       // sw is complete
  f32x2 wr[27][2];
#pragma unroll
  for (int tp = 0; tp < 27; ++tp) {
    const f32x4 q4 = *reinterpret_cast<const f32x4*>(sw + tp * 32 + 4 * c4);
    wr[tp][0] = (f32x2){q4.x, q4.y}; wr[tp][1] = (f32x2){q4.z, q4.w};
  }
  const float bias = p.bias ? p.bias[0] : 0.f;
  float s1 = bias, s2 = 0.f;                             // s1: logit d = q (has bias + P0), s2: logit d = q-1 (+ P1)

  // The loop body is free of branches around vector-memory instructions and every plane does the same thing (hipcc's wait
  // bookkeeping turns imprecise at the join of any such branch and then drains the whole load queue before every use —
  // measured: that alone made a plane cost a full L2 round trip).  Planes beyond D are the zero halo plane: they are fetched
  // like any other, skip the activation (a scalar select on the mask) and project to zero, which completes the last logit
  // (d = D-1) with the exact bits and lets the pair loop run an even number of planes.
  auto plane = [&](f32x4 (&r)[NK], int q) {
    char* slot = ring + (q & 1) * p.slot_bytes;
    float* outp = p.a_out + run_base + (long)min(q, D + 1) * plane_vox * 32;
    const unsigned tm = q <= D ? tmask : 0u;
    // activate in registers, stage into LDS (swizzled), store the by-product
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      f32x4 yv = r[k];
      if (IN != 0) {
        f32x4 a;
        a.x = fmaf(yv.x, in_sc.x, in_sh.x); a.y = fmaf(yv.y, in_sc.y, in_sh.y);
        a.z = fmaf(yv.z, in_sc.z, in_sh.z); a.w = fmaf(yv.w, in_sc.w, in_sh.w);
        a.x = fmaxf(a.x, a.x * p.slope); a.y = fmaxf(a.y, a.y * p.slope);
        a.z = fmaxf(a.z, a.z * p.slope); a.w = fmaxf(a.w, a.w * p.slope);
        yv = ((tm >> k) & 1u) ? a : yv;                  // halo voxels and halo planes: the zero that was loaded
      }
      const int v = pl + 32 * k;
      char* dst = ((lmask >> k) & 1u) ? slot + v * 128 + ((c4 ^ ((v >> 1) & 7)) << 4) : dump;   // beyond the run: a private dump slot
      *reinterpret_cast<f32x4*>(dst) = yv;
    }
    fetch(r, q + 2);                                     // the registers are free again: two planes ahead
    __syncthreads();                                     // plane q is complete in LDS (and plane q-1's readers are done)
    if (OUT) {
      // by-product: this thread's chunk of the workgroup's own 32 positions, read back from LDS — ONE unconditional store
      // per thread and plane (a store under a per-lane branch makes hipcc drain the load queue at every join; halo columns
      // inside the tile hold the zeros that were loaded, so writing them keeps the halo zero)
      const int vo = Wp + 1 + pl;
      const f32x4 mine = *reinterpret_cast<const f32x4*>(slot + vo * 128 + ((c4 ^ ((vo >> 1) & 7)) << 4));
      *reinterpret_cast<f32x4*>(outp + (long)vo * 32) = mine;
    }
    // project plane q onto the three kd slices: packed fp32 FMAs over channel pairs
    f32x2 p0 = {0.f, 0.f}, p1 = {0.f, 0.f}, p2 = {0.f, 0.f};
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
      const int v = pl + (t9 / 3) * Wp + (t9 % 3);
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(slot + v * 128 + ((c4 ^ ((v >> 1) & 7)) << 4));
      const f32x2 lo = {a4.x, a4.y}, hi = {a4.z, a4.w};
      p0 = __builtin_elementwise_fma(lo, wr[t9][0], p0); p0 = __builtin_elementwise_fma(hi, wr[t9][1], p0);
      p1 = __builtin_elementwise_fma(lo, wr[9 + t9][0], p1); p1 = __builtin_elementwise_fma(hi, wr[9 + t9][1], p1);
      p2 = __builtin_elementwise_fma(lo, wr[18 + t9][0], p2); p2 = __builtin_elementwise_fma(hi, wr[18 + t9][1], p2);
    }
    float q0 = p0.x + p0.y, q1 = p1.x + p1.y, q2 = p2.x + p2.y;
    // wavefront-shuffle reduction over the 8 channel lanes of the position (fixed order)
    q0 += __shfl_xor(q0, 1, 64); q1 += __shfl_xor(q1, 1, 64); q2 += __shfl_xor(q2, 1, 64);
    q0 += __shfl_xor(q0, 2, 64); q1 += __shfl_xor(q1, 2, 64); q2 += __shfl_xor(q2, 2, 64);
    q0 += __shfl_xor(q0, 4, 64); q1 += __shfl_xor(q1, 4, 64); q2 += __shfl_xor(q2, 4, 64);
    // plane q is the kd = 2 plane of output d = q - 2, the kd = 1 plane of d = q - 1, the kd = 0 plane of d = q
    const float done = s2 + q2;
    s2 = s1 + q1;
    s1 = bias + q0;
    if (q >= 2 && q - 2 < D && c4 == 0) lg[(q - 2) * 32 + pl] = done;   // (to global memory at the end, all planes at once)
  };
  for (int q = 1; q <= D + 1; q += 2) {                   // planes 1 .. D+1 (+1): the last logit completes on the halo plane
    plane(ra, q);
    plane(rb, q + 1);
  }
  __syncthreads();

  // ---- soft-argmax, arg-max, FCS: lane j of a position holds d = j, j+8, j+16, j+24 ----
  float l[4];
  float m1 = -INFINITY, m2 = -INFINITY, sum = 0.f;
  int am = 0x7fffffff;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int d = c4 + 8 * k;
    l[k] = d < D ? lg[d * 32 + pl] : -INFINITY;
    if (d < D && interior) p.logits[(((long)b * D + d) * p.g.H + y) * p.g.W + x] = l[k];
    if (d < D) {
      sum += l[k];
      if (l[k] > m1) { m2 = m1; m1 = l[k]; am = d; }
      else if (l[k] > m2) { m2 = l[k]; }
    }
  }
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) {
    const float om1 = __shfl_xor(m1, o, 64), om2 = __shfl_xor(m2, o, 64), osum = __shfl_xor(sum, o, 64);
    const int oam = __shfl_xor(am, o, 64);
    sum += osum;
    // top-2 of the union (duplicates of the maximum count twice, as in a sort); first index on ties
    const float lo = fminf(m1, om1);
    m2 = fmaxf(lo, fmaxf(m2, om2));
    if (om1 > m1 || (om1 == m1 && oam < am)) { m1 = om1; am = oam; }
  }
  float e[4], se = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) { e[k] = (c4 + 8 * k) < D ? expf(l[k] - m1) : 0.f; se += e[k]; }
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) se += __shfl_xor(se, o, 64);
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) acc += (e[k] / se) * (float)(c4 + 8 * k);
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) acc += __shfl_xor(acc, o, 64);
  if (c4 == 0 && interior) {
    const long i = ((long)b * p.g.H + y) * p.g.W + x;
    p.pred[i] = acc;
    if (p.argmax) p.argmax[i] = am;
    if (p.fcs) p.fcs[i] = (D > 2) ? m1 - (sum - m1 - m2) / (float)(D - 2) : 0.f;
  }
}

// ---- second generation (round 4): no LDS staging, no barrier in the plane loop -----------------------------------------------
// What the kernel above costs at 4 pairs (40 us for 24 MB; 95 / 211 us at 16 / 32 pairs — it does not saturate, so it is not
// the launch size): a workgroup stages a RUN of 34 + 2 Wp = 194 voxels per plane for its 32 positions (6 x the bytes through
// L2: 4.13 x on the HBM counters), every plane is a barrier with ONE wave per SIMD to cover it (240 workgroups at 4 pairs), and
// the staged image is read back with 180 k bank conflicts per launch.  Here every thread fetches the nine tap chunks of ITS
// position straight from L2 / L1 — a wave instruction is 8 voxels x 128 B = one contiguous KB, the three column-shifted loads
// of a row overlap 7/8 in L1 — three planes in flight per thread (the kernel runs one workgroup per CU, i.e. one wave per SIMD:
// 512 registers), activates them in registers (nine chunks instead of eight: the same vector work), and projects with the same
// packed FMAs in the same order: bit-identical logits.  The four waves of a workgroup never wait for each other.  Workgroups
// that share an XCD (block ids congruent mod 8) take CONSECUTIVE tiles, so the rows two tiles share are fetched into one L2.
// NPOS = positions per workgroup (8 lanes each): 32, or 16 when 32 would leave CUs without a workgroup
template <int IN, bool OUT, int NPOS>
__global__ __launch_bounds__(8 * NPOS, 1) void agg_tail_direct_kernel(TailArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // layout: [BatchNorm merge scratch][logits TAIL_MAXD x 32 floats][weights 27 x 32 floats]
  float* lg = reinterpret_cast<float*>(smem + BN_MERGE_SCRATCH_BYTES);
  float* sw = lg + TAIL_MAXD * NPOS;
  const int Wp = p.g.Wp, D = p.g.D;
  int col;
  {                                                       // XCD x takes the tiles [start_x, start_x + count_x): a bijection
    const int n = (int)gridDim.x, xcd = (int)blockIdx.x & 7, q = (int)blockIdx.x >> 3;
    col = xcd * (n >> 3) + min(xcd, n & 7) + q;
  }
  const int b = col / p.tiles_per_plane, t = col - b * p.tiles_per_plane;
  const int first = p.g.ph * Wp + p.g.pw;
  const int pos0 = min(first + NPOS * t, first + p.npos - NPOS);
  const long plane_vox = (long)Wp * p.g.Hp;
  const int c4 = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int pos = pos0 + pl;
  const int yp = (int)__umulhi((unsigned)pos, p.wp_magic), xp = pos - yp * Wp;
  const bool interior = xp >= p.g.pw && xp < p.g.pw + p.g.W;          // rows are interior by construction
  const int y = yp - p.g.ph, x = xp - p.g.pw;
  // tap t9 of this position is an interior voxel (gets the activation; the others are the zero padding that was loaded)
  unsigned tmask = 0u;
#pragma unroll
  for (int t9 = 0; t9 < 9; ++t9) {
    const int ps = pos + (t9 / 3 - 1) * Wp + (t9 % 3 - 1);
    const int yy = (int)__umulhi((unsigned)ps, p.wp_magic), xx = ps - yy * Wp;
    tmask |= ((xx >= p.g.pw && xx < p.g.pw + p.g.W && yy >= p.g.ph && yy < p.g.ph + p.g.H) ? 1u : 0u) << t9;
  }
  const long base = ((long)b * p.g.Dp * plane_vox + pos) * 32 + 4 * c4;      // + q * plane_vox * 32
  auto fetch = [&](f32x4 (&r)[9], int q) {              // padded plane min(q, D+1) (D+1: the zero halo plane); branch-free
    const float* src = p.x + base + (long)min(q, D + 1) * plane_vox * 32;
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) r[t9] = *reinterpret_cast<const f32x4*>(src + ((t9 / 3 - 1) * Wp + (t9 % 3 - 1)) * 32);
  };
  f32x4 r0[9], r1[9], r2[9];
  fetch(r0, 1); fetch(r1, 2); fetch(r2, 3);

  f32x4 in_sc = {0.f, 0.f, 0.f, 0.f}, in_sh = {0.f, 0.f, 0.f, 0.f};
  if (IN == 1) {
    in_sc = *reinterpret_cast<const f32x4*>(p.in_scale + 4 * c4);
    in_sh = *reinterpret_cast<const f32x4*>(p.in_shift + 4 * c4);
  }
  if (IN == 2) {
    const float* tab = bn_merge_partials<16, NPOS / 4>(p.in_bn, smem, blockIdx.x == 0);
    in_sc = *reinterpret_cast<const f32x4*>(tab + 4 * c4);
    in_sh = *reinterpret_cast<const f32x4*>(tab + 32 + 4 * c4);
  }
  for (int i = threadIdx.x; i < 27 * 32; i += 8 * NPOS) { const int tp = i >> 5, c = i & 31; sw[i] = p.w[c * 27 + tp]; }
  __syncthreads();                                       // sw is complete
  f32x2 wr[27][2];
#pragma unroll
  for (int tp = 0; tp < 27; ++tp) {
    const f32x4 q4 = *reinterpret_cast<const f32x4*>(sw + tp * 32 + 4 * c4);
    wr[tp][0] = (f32x2){q4.x, q4.y}; wr[tp][1] = (f32x2){q4.z, q4.w};
  }
  const float bias = p.bias ? p.bias[0] : 0.f;
  float s1 = bias, s2 = 0.f;                             // s1: logit d = q (has bias + P0), s2: logit d = q-1 (+ P1)

  auto plane = [&](f32x4 (&r)[9], int q) {
    const unsigned tm = q <= D ? tmask : 0u;             // the halo plane beyond D: zeros, no activation
    f32x2 p0 = {0.f, 0.f}, p1 = {0.f, 0.f}, p2 = {0.f, 0.f};
    f32x4 centre = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
      f32x4 a4 = r[t9];
      if (IN != 0) {
        f32x4 a;
        a.x = fmaf(a4.x, in_sc.x, in_sh.x); a.y = fmaf(a4.y, in_sc.y, in_sh.y);
        a.z = fmaf(a4.z, in_sc.z, in_sh.z); a.w = fmaf(a4.w, in_sc.w, in_sh.w);
        a.x = fmaxf(a.x, a.x * p.slope); a.y = fmaxf(a.y, a.y * p.slope);
        a.z = fmaxf(a.z, a.z * p.slope); a.w = fmaxf(a.w, a.w * p.slope);
        a4 = ((tm >> t9) & 1u) ? a : a4;
      }
      if (t9 == 4) centre = a4;
      const f32x2 lo = {a4.x, a4.y}, hi = {a4.z, a4.w};
      p0 = __builtin_elementwise_fma(lo, wr[t9][0], p0); p0 = __builtin_elementwise_fma(hi, wr[t9][1], p0);
      p1 = __builtin_elementwise_fma(lo, wr[9 + t9][0], p1); p1 = __builtin_elementwise_fma(hi, wr[9 + t9][1], p1);
      p2 = __builtin_elementwise_fma(lo, wr[18 + t9][0], p2); p2 = __builtin_elementwise_fma(hi, wr[18 + t9][1], p2);
    }
    // by-product: the activated voxel of this thread's own position — ONE unconditional store per thread and plane (planes
    // beyond D write the zeros they loaded onto the zero halo plane; halo columns inside the tile likewise)
#if defined(TAIL_EXP_NT_STORE)                            // (experiment: the by-product past the caches; tests/tools/exp_step.sh)
    if (OUT) __builtin_nontemporal_store(centre, reinterpret_cast<f32x4*>(p.a_out + base + (long)min(q, D + 1) * plane_vox * 32));
#elif defined(TAIL_EXP_NO_STORE)                          // (diagnostic: no by-product at all — the backward pass is then wrong)
    if (OUT && q > 1000) *reinterpret_cast<f32x4*>(p.a_out + base + (long)min(q, D + 1) * plane_vox * 32) = centre;
#else
    if (OUT) *reinterpret_cast<f32x4*>(p.a_out + base + (long)min(q, D + 1) * plane_vox * 32) = centre;
#endif
    float q0 = p0.x + p0.y, q1 = p1.x + p1.y, q2 = p2.x + p2.y;
    q0 += __shfl_xor(q0, 1, 64); q1 += __shfl_xor(q1, 1, 64); q2 += __shfl_xor(q2, 1, 64);
    q0 += __shfl_xor(q0, 2, 64); q1 += __shfl_xor(q1, 2, 64); q2 += __shfl_xor(q2, 2, 64);
    q0 += __shfl_xor(q0, 4, 64); q1 += __shfl_xor(q1, 4, 64); q2 += __shfl_xor(q2, 4, 64);
    const float done = s2 + q2;
    s2 = s1 + q1;
    s1 = bias + q0;
    if (q >= 2 && q - 2 < D && c4 == 0) lg[(q - 2) * NPOS + pl] = done;
  };
  for (int q = 1; q <= D + 1; q += 3) {                   // planes 1 .. D+1 (and up to two more passes over the zero plane)
    plane(r0, q); fetch(r0, q + 3);
    plane(r1, q + 1); fetch(r1, q + 4);
    plane(r2, q + 2); fetch(r2, q + 5);
  }
  __syncthreads();

  // ---- soft-argmax, arg-max, FCS: lane j of a position holds d = j, j+8, j+16, j+24 (as above) ----
  float l[4];
  float m1 = -INFINITY, m2 = -INFINITY, sum = 0.f;
  int am = 0x7fffffff;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int d = c4 + 8 * k;
    l[k] = d < D ? lg[d * NPOS + pl] : -INFINITY;
    if (d < D && interior) p.logits[(((long)b * D + d) * p.g.H + y) * p.g.W + x] = l[k];
    if (d < D) {
      sum += l[k];
      if (l[k] > m1) { m2 = m1; m1 = l[k]; am = d; }
      else if (l[k] > m2) { m2 = l[k]; }
    }
  }
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) {
    const float om1 = __shfl_xor(m1, o, 64), om2 = __shfl_xor(m2, o, 64), osum = __shfl_xor(sum, o, 64);
    const int oam = __shfl_xor(am, o, 64);
    sum += osum;
    const float lo = fminf(m1, om1);
    m2 = fmaxf(lo, fmaxf(m2, om2));
    if (om1 > m1 || (om1 == m1 && oam < am)) { m1 = om1; am = oam; }
  }
  float e[4], se = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) { e[k] = (c4 + 8 * k) < D ? expf(l[k] - m1) : 0.f; se += e[k]; }
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) se += __shfl_xor(se, o, 64);
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) acc += (e[k] / se) * (float)(c4 + 8 * k);
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) acc += __shfl_xor(acc, o, 64);
  if (c4 == 0 && interior) {
    const long i = ((long)b * p.g.H + y) * p.g.W + x;
    p.pred[i] = acc;
    if (p.argmax) p.argmax[i] = am;
    if (p.fcs) p.fcs[i] = (D > 2) ? m1 - (sum - m1 - m2) / (float)(D - 2) : 0.f;
  }
}

// ---- host ------------------------------------------------------------------------------------------------------------
static int tail_run(const as_pcl* g) { return 34 + 2 * (g->W + 2 * g->pw); }
#define TAIL_TABLE_BYTES (27 * 128)

extern "C" int as_agg_tail_ok(const as_pcl* g) {
  if (!as_pcl_ok(g) || g->pd != 1 || g->ph != 1 || g->pw != 1) return 0;
  if (g->D > TAIL_MAXD) return 0;
  const int Wp = g->W + 2;
  if ((long)(g->H - 1) * Wp + g->W < 32) return 0;
  return tail_run(g) <= 12 * 32 ? 1 : 0;                            // at most twelve chunks per thread and plane
}

template <int IN, int NK, bool OUT>
static int tail_launch_t(const TailArgs& a, int grid, int lds_bytes, hipStream_t st) {
  static AsPerDevice attr_set;
  if (!attr_set.get()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(agg_tail_kernel<IN, NK, OUT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { as_set_error("as_agg_tail_fwd: %s", hipGetErrorString(e)); return AS_ERR_LAUNCH; }
    attr_set.set();
  }
  hipLaunchKernelGGL((agg_tail_kernel<IN, NK, OUT>), dim3(grid), dim3(256), lds_bytes, st, a);
  return AS_OK;
}
template <int IN, bool OUT>
static int tail_launch_direct(const TailArgs& a0, hipStream_t st) {
  // 16 positions per workgroup while 32 would give fewer than two workgroups per CU (4 pairs at 24 x 78: 240 of 32)
  TailArgs a = a0;
  const int B = a.g.B;
  const bool small = (long)B * as_div_up(a.npos, 32) < 512 && a.npos >= 16;
  const int npos = small ? 16 : 32;
  a.tiles_per_plane = as_div_up(a.npos, npos);
  const int grid = B * a.tiles_per_plane;
  const int lds = BN_MERGE_SCRATCH_BYTES + TAIL_MAXD * npos * 4 + TAIL_TABLE_BYTES;
  if (small) hipLaunchKernelGGL((agg_tail_direct_kernel<IN, OUT, 16>), dim3(grid), dim3(128), lds, st, a);
  else hipLaunchKernelGGL((agg_tail_direct_kernel<IN, OUT, 32>), dim3(grid), dim3(256), lds, st, a);
  return AS_OK;
}
template <int IN, bool OUT>
static int tail_launch_nk(const TailArgs& a, int grid, int lds_bytes, hipStream_t st) {
#ifndef TAIL_FIRST_GENERATION                             // (A/B builds: EXTRA=-DTAIL_FIRST_GENERATION)
  return tail_launch_direct<IN, OUT>(a, st);
#endif
  const int nk = (a.run + 31) / 32;
  if (nk <= 6) return tail_launch_t<IN, 6, OUT>(a, grid, lds_bytes, st);
  if (nk <= 8) return tail_launch_t<IN, 8, OUT>(a, grid, lds_bytes, st);
  return tail_launch_t<IN, 12, OUT>(a, grid, lds_bytes, st);
}

extern "C" int as_agg_tail_fwd(const float* x, const as_pcl* g, const float* in_scale, const float* in_shift,
                               const as_bn_merge* in_bn, float* a_out,
                               const float* w, const float* bias, float slope, float* logits, float* pred,
                               int32_t* argmax, float* fcs, void* stream) {
  AS_CHECK_ARG(as_agg_tail_ok(g) == 1, "as_agg_tail_fwd: geometry not supported (as_agg_tail_ok() == 0)");
  AS_CHECK_ARG(x && w && logits && pred, "as_agg_tail_fwd: null pointer");
  AS_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "as_agg_tail_fwd: in_scale and in_shift must pair");
  AS_CHECK_ARG(!(in_scale && in_bn), "as_agg_tail_fwd: pass the input BatchNorm either as an affine or as partials, not both");
  AS_CHECK_ARG(a_out == nullptr || ((in_scale != nullptr || in_bn != nullptr) && a_out != x),
               "as_agg_tail_fwd: a_out needs the input BatchNorm and must not alias x");
  TailArgs a;
  a.x = x; a.in_scale = in_scale; a.in_shift = in_shift; a.a_out = a_out; a.w = w; a.bias = bias;
  if (in_bn) AS_CHECK_ARG(bn_merge_fill(&a.in_bn, in_bn), "as_agg_tail_fwd: incomplete as_bn_merge block");
  a.logits = logits; a.pred = pred; a.argmax = argmax; a.fcs = fcs; a.slope = slope;
  a.g = as_make_dev(g);
  a.npos = (g->H - 1) * a.g.Wp + g->W;
  a.tiles_per_plane = as_div_up(a.npos, 32);
  a.run = tail_run(g);
  a.slot_bytes = (a.run + 7) / 8 * 1024;
  a.wp_magic = (unsigned)((((uint64_t)1 << 32) + a.g.Wp - 1) / a.g.Wp);
  const int lds_bytes = 2 * a.slot_bytes + TAIL_MAXD * 128 + TAIL_TABLE_BYTES + 4096;
  hipStream_t st = (hipStream_t)stream;
  const int grid = g->B * a.tiles_per_plane;
  // algorithmic bytes: the volume read once (optionally written once), logits written, three [B,H,W] maps written
  const double vol = 128.0 * (double)g->B * g->D * g->H * g->W;
  as_prof_mark(8, st, 1, 0.0);
  int e;
  if (in_bn) e = a_out ? tail_launch_nk<2, true>(a, grid, lds_bytes, st) : tail_launch_nk<2, false>(a, grid, lds_bytes, st);
  else if (in_scale) e = a_out ? tail_launch_nk<1, true>(a, grid, lds_bytes, st) : tail_launch_nk<1, false>(a, grid, lds_bytes, st);
  else e = tail_launch_nk<0, false>(a, grid, lds_bytes, st);
  if (e) return e;
  as_prof_mark(8, st, 0, vol * (a_out ? 2.0 : 1.0) + 4.0 * (double)g->B * g->H * g->W * (g->D + 3));
  AS_CHECK_LAUNCH("as_agg_tail_fwd");
  return AS_OK;
}
