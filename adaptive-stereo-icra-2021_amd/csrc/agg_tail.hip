// a4 + a5 + a8 in ONE kernel (with the last aggregation layer's BatchNorm + LeakyReLU applied on the way in):
//   conv3d_alone = nn.Conv3d(32, 1, 3, padding=1)  -> logits            stereo_net.py:162,187
//   softmax(+logits, dim=1), sum_d d * p_d          -> coarse disparity  stereo_net.py:190-192,124-134
//   arg-max index over d, FCS = sorted[0] - mean(sorted[2:])              utils/feature_contrast.py:12-23
// It replaces bn_act_fwd (element-wise pass over the volume), conv32to1_fwd_kernel (27 taps x 128-byte lines per voxel
// through L1) and softargmax_fwd_kernel (three passes over the logits): 46 us of launches at 4 pairs for 12 MB of input.
//
// A workgroup owns 32 consecutive positions of the flattened padded plane (conv3d_lds.hip's position trick) for ALL
// disparities and walks down the disparity axis: plane q (a run of 34 + 2*Wp voxels) lands by LDS-DMA into a ring of three
// slots while plane q-1 is being used.  N = 1 rules out the matrix cores (31/32 of an MFMA would be wasted), so the 32->1
// convolution is "project, then shift": every staged plane is projected ONCE onto the three kd slices of the kernel,
//   P_kd[q][p] = sum_{kh,kw,c} a[q][p + (kh-1)*Wp + (kw-1)][c] * w[c][kd][kh][kw],
// by 8 lanes per position (one 16-byte channel chunk each, the 27 x 4 weights they need in registers, a three-step
// wavefront-shuffle reduction), and  logit[d] = bias + P_0[d] + P_1[d+1] + P_2[d+2]  falls out of two carried registers.
// The D logits of a position stay in LDS; the soft-argmax, the arg-max index (first maximum, as torch.argmax) and the FCS
// are then wavefront-shuffle reductions over those 8 lanes (each holds the disparities d = j, j+8, j+16).
//   IN 1 / IN 2: x is the last layer's RAW convolution output (IN 2: its BatchNorm still in partials, merged by every
//         workgroup, bn_merge.h); lrelu(x * scale + shift) is applied to each plane once in LDS (halo
//         voxels skipped by a precomputed bit mask) and the activated tensor is written back when a_out is given.
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
  int tiles_per_plane, npos, run, groups, slot_bytes;
  unsigned wp_magic;
  float slope;
};

typedef __attribute__((address_space(3))) void* tail_lds_t;

__device__ inline void tail_dma_1kb(const float* sbase, unsigned voff, unsigned m0) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               :: "s"(m0), "v"(voff), "s"(sbase) : "memory", "m0");
}

#define TAIL_MAXD 32

template <int IN>
__global__ __launch_bounds__(256) void agg_tail_kernel(TailArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // layout: [3 plane slots][logits TAIL_MAXD x 32 floats][weights 27 x 32 floats][256 x 16-byte dump slots]
  char* ring = smem;
  float* lg = reinterpret_cast<float*>(smem + 3 * p.slot_bytes);
  float* sw = lg + TAIL_MAXD * 32;
  char* dump = reinterpret_cast<char*>(sw + 27 * 32) + threadIdx.x * 16;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((tail_lds_t)ring));
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int Wp = p.g.Wp, D = p.g.D;
  const int col = blockIdx.x;
  const int b = col / p.tiles_per_plane, t = col - b * p.tiles_per_plane;
  const int first = p.g.ph * Wp + p.g.pw;
  const int pos0 = min(first + 32 * t, first + p.npos - 32);
  const long plane_vox = (long)Wp * p.g.Hp;

  const unsigned vl = (unsigned)(lane >> 3), sl = (unsigned)(lane & 7);
  const unsigned off_reg = vl * 128u + ((sl ^ ((((unsigned)wave & 1u) << 2) | (vl >> 1))) << 4);
  const unsigned tail_v0 = (unsigned)(p.run - 8);
  const unsigned off_tail = vl * 128u + ((sl ^ ((((tail_v0 + vl) >> 1)) & 7u)) << 4);
  auto issue = [&](int q) {                              // padded plane q -> slot q % 3
    const float* src = p.x + (((long)b * p.g.Dp + q) * plane_vox + (pos0 - Wp - 1)) * 32;
    const unsigned slot = lds0 + (unsigned)((q % 3) * p.slot_bytes);
    for (int i = wave; i < p.groups - 1; i += 4) tail_dma_1kb(src + i * 256, off_reg, slot + (unsigned)(i * 1024));
    if (((p.groups - 1) & 3) == wave) tail_dma_1kb(src + (long)tail_v0 * 32, off_tail, slot + tail_v0 * 128u);
  };
  issue(1);
  if (D >= 2) issue(2);

  // weights -> LDS (transposed to [tap][channel]) -> this lane's 27 x 4 registers
  for (int i = threadIdx.x; i < 27 * 32; i += 256) { const int tp = i >> 5, c = i & 31; sw[i] = p.w[c * 27 + tp]; }

  const int c4 = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int pos = pos0 + pl;
  const int yp = (int)__umulhi((unsigned)pos, p.wp_magic), xp = pos - yp * Wp;
  const bool interior = xp >= p.g.pw && xp < p.g.pw + p.g.W;          // rows are interior by construction
  const int y = yp - p.g.ph, x = xp - p.g.pw;

  // IN 1: element-wise pass (see agg3d.hip): thread (cg = c4, row = pl) owns chunk cg of voxels pl + 32k
  f32x4 in_sc = {0.f, 0.f, 0.f, 0.f}, in_sh = {0.f, 0.f, 0.f, 0.f};
  unsigned tmask = 0u, omask = 0u;
  if (IN == 1) {
    in_sc = *reinterpret_cast<const f32x4*>(p.in_scale + 4 * c4);
    in_sh = *reinterpret_cast<const f32x4*>(p.in_shift + 4 * c4);
  }
  if (IN == 2) {            // merge the partials while planes 1 and 2 are in flight; scratch = the dump slots
    const float* tab = bn_merge_partials(p.in_bn, reinterpret_cast<char*>(sw + 27 * 32), blockIdx.x == 0);
    in_sc = *reinterpret_cast<const f32x4*>(tab + 4 * c4);
    in_sh = *reinterpret_cast<const f32x4*>(tab + 32 + 4 * c4);
  }
  if (IN != 0) {
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const int v = pl + 32 * k;
      const int ps = pos0 - Wp - 1 + v;
      const int yy = (int)__umulhi((unsigned)ps, p.wp_magic), xx = ps - yy * Wp;
      const bool in = v < p.run && xx >= p.g.pw && xx < p.g.pw + p.g.W && yy >= p.g.ph && yy < p.g.ph + p.g.H;
      tmask |= (in ? 1u : 0u) << k;
      omask |= ((in && v >= Wp + 1 && v < Wp + 33) ? 1u : 0u) << k;
    }
  }
  __syncthreads();                                       // sw is complete
  float wr[27][4];
#pragma unroll
  for (int tp = 0; tp < 27; ++tp) {
    const f32x4 q4 = *reinterpret_cast<const f32x4*>(sw + tp * 32 + 4 * c4);
    wr[tp][0] = q4.x; wr[tp][1] = q4.y; wr[tp][2] = q4.z; wr[tp][3] = q4.w;
  }
  const float bias = p.bias ? p.bias[0] : 0.f;
  float s1 = bias, s2 = 0.f;                             // s1: logit d = q (has bias + P0), s2: logit d = q-1 (+ P1)

  for (int q = 1; q <= D; ++q) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // planes q and q+1 have landed (this wave's part)
    __syncthreads();                                       // ... every wave's; plane q-1's slot is free
    char* slot = ring + (q % 3) * p.slot_bytes;
    if (IN != 0) {
      // branch-free element-wise pass (agg3d.hip): twelve reads in flight together, unchanged chunks go to a dump slot
      float* outp = p.a_out + (((long)b * p.g.Dp + q) * plane_vox + (pos0 - Wp - 1)) * 32 + 4 * c4;
      f32x4* cp[12];
      f32x4 yv[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        const int v = pl + 32 * k;
        char* c = slot + v * 128 + ((c4 ^ ((v >> 1) & 7)) << 4);
        cp[k] = reinterpret_cast<f32x4*>(((tmask >> k) & 1u) ? c : dump);
        yv[k] = *cp[k];
      }
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        yv[k].x = fmaf(yv[k].x, in_sc.x, in_sh.x); yv[k].y = fmaf(yv[k].y, in_sc.y, in_sh.y);
        yv[k].z = fmaf(yv[k].z, in_sc.z, in_sh.z); yv[k].w = fmaf(yv[k].w, in_sc.w, in_sh.w);
        yv[k].x = fmaxf(yv[k].x, yv[k].x * p.slope); yv[k].y = fmaxf(yv[k].y, yv[k].y * p.slope);
        yv[k].z = fmaxf(yv[k].z, yv[k].z * p.slope); yv[k].w = fmaxf(yv[k].w, yv[k].w * p.slope);
        *cp[k] = yv[k];
      }
      if (p.a_out != nullptr) {
#pragma unroll
        for (int k = 0; k < 12; ++k)
          if ((omask >> k) & 1u) *reinterpret_cast<f32x4*>(outp + (long)(pl + 32 * k) * 32) = yv[k];
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // by-product stores out of the queue before the next DMA
      __syncthreads();
    }
    if (q + 2 <= D) issue(q + 2);

    // project plane q onto the three kd slices
    float p0 = 0.f, p1 = 0.f, p2 = 0.f;
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
      const int v = pl + (t9 / 3) * Wp + (t9 % 3);
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(slot + v * 128 + ((c4 ^ ((v >> 1) & 7)) << 4));
      p0 = fmaf(a4.x, wr[t9][0], p0); p0 = fmaf(a4.y, wr[t9][1], p0); p0 = fmaf(a4.z, wr[t9][2], p0); p0 = fmaf(a4.w, wr[t9][3], p0);
      p1 = fmaf(a4.x, wr[9 + t9][0], p1); p1 = fmaf(a4.y, wr[9 + t9][1], p1); p1 = fmaf(a4.z, wr[9 + t9][2], p1); p1 = fmaf(a4.w, wr[9 + t9][3], p1);
      p2 = fmaf(a4.x, wr[18 + t9][0], p2); p2 = fmaf(a4.y, wr[18 + t9][1], p2); p2 = fmaf(a4.z, wr[18 + t9][2], p2); p2 = fmaf(a4.w, wr[18 + t9][3], p2);
    }
    // wavefront-shuffle reduction over the 8 channel lanes of the position (fixed order)
    p0 += __shfl_xor(p0, 1, 64); p1 += __shfl_xor(p1, 1, 64); p2 += __shfl_xor(p2, 1, 64);
    p0 += __shfl_xor(p0, 2, 64); p1 += __shfl_xor(p1, 2, 64); p2 += __shfl_xor(p2, 2, 64);
    p0 += __shfl_xor(p0, 4, 64); p1 += __shfl_xor(p1, 4, 64); p2 += __shfl_xor(p2, 4, 64);
    // plane q is the kd = 2 plane of output d = q - 2, the kd = 1 plane of d = q - 1, the kd = 0 plane of d = q
    const float done = s2 + p2;
    s2 = s1 + p1;
    s1 = bias + p0;
    if (q >= 2 && c4 == 0) {
      lg[(q - 2) * 32 + pl] = done;
      if (interior) p.logits[(((long)b * D + (q - 2)) * p.g.H + y) * p.g.W + x] = done;
    }
  }
  if (c4 == 0) {                                           // d = D-1: its kd = 2 plane is the zero halo
    lg[(D - 1) * 32 + pl] = s2;
    if (interior) p.logits[(((long)b * D + (D - 1)) * p.g.H + y) * p.g.W + x] = s2;
  }
  __syncthreads();

  // ---- soft-argmax, arg-max, FCS: lane j of a position holds d = j, j+8, j+16, (j+24) ----
  float l[4];
  float m1 = -INFINITY, m2 = -INFINITY, sum = 0.f;
  int am = 0x7fffffff;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int d = c4 + 8 * k;
    l[k] = d < D ? lg[d * 32 + pl] : -INFINITY;
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

// ---- host ------------------------------------------------------------------------------------------------------------
static int tail_run(const as_pcl* g) { return 34 + 2 * (g->W + 2 * g->pw); }

extern "C" int as_agg_tail_ok(const as_pcl* g) {
  if (!as_pcl_ok(g) || g->pd != 1 || g->ph != 1 || g->pw != 1) return 0;
  if (g->D > TAIL_MAXD) return 0;
  const int Wp = g->W + 2;
  if ((long)(g->H - 1) * Wp + g->W < 32) return 0;
  if (tail_run(g) > 12 * 32) return 0;                              // the element-wise pass covers 12 x 32 voxels per plane
  const long lds = 3L * ((tail_run(g) + 7) / 8) * 1024 + TAIL_MAXD * 128 + 27 * 128 + 4096;
  return lds <= 156 * 1024 ? 1 : 0;
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
  a.groups = (a.run + 7) / 8;
  a.slot_bytes = a.groups * 1024;
  a.wp_magic = (unsigned)((((uint64_t)1 << 32) + a.g.Wp - 1) / a.g.Wp);
  const int lds_bytes = 3 * a.slot_bytes + TAIL_MAXD * 128 + 27 * 128 + 4096;
  hipStream_t st = (hipStream_t)stream;
  const int mode = in_bn ? 2 : (in_scale ? 1 : 0);
  const void* fns[3] = {reinterpret_cast<const void*>(agg_tail_kernel<0>), reinterpret_cast<const void*>(agg_tail_kernel<1>),
                        reinterpret_cast<const void*>(agg_tail_kernel<2>)};
  static bool attr_set[3] = {false, false, false};
  if (!attr_set[mode]) {
    hipError_t e = hipFuncSetAttribute(fns[mode], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { as_set_error("as_agg_tail_fwd: %s", hipGetErrorString(e)); return AS_ERR_LAUNCH; }
    attr_set[mode] = true;
  }
  const int grid = g->B * a.tiles_per_plane;
  // algorithmic bytes: the volume read once (optionally written once), logits written, three [B,H,W] maps written
  const double vol = 128.0 * (double)g->B * g->D * g->H * g->W;
  as_prof_mark(8, st, 1, 0.0);
  if (mode == 2) hipLaunchKernelGGL(agg_tail_kernel<2>, dim3(grid), dim3(256), lds_bytes, st, a);
  else if (mode == 1) hipLaunchKernelGGL(agg_tail_kernel<1>, dim3(grid), dim3(256), lds_bytes, st, a);
  else hipLaunchKernelGGL(agg_tail_kernel<0>, dim3(grid), dim3(256), lds_bytes, st, a);
  as_prof_mark(8, st, 0, vol * (a_out ? 2.0 : 1.0) + 4.0 * (double)g->B * g->H * g->W * (g->D + 3));
  AS_CHECK_LAUNCH("as_agg_tail_fwd");
  return AS_OK;
}
