// Backward of the cost-aggregation tail in ONE launch: soft-argmax (a5) + conv3d_alone (a4), data and weight gradient.
// Reference semantics: DisparityRegression / F.softmax(dim=1) (stereo_net.py:124-134, 190-192) and
// conv3d_alone = nn.Conv3d(32, 1, 3, padding=1) (stereo_net.py:162, 187) under autograd:
//   g_logits[d] = p_d * (d - pred) * g_pred (+ g_logits_in[d])
//   g_a[v][c]   = sum_t g_logits[v - off_t] * w[c][t]
//   g_w[c][t]   = sum_u a[u][c] * g_logits[u - off_t],   g_bias = sum_v g_logits[v]
//
// The first generation ran three launches (as_softargmax_bwd: one thread per pixel = 30 workgroups at 4 pairs;
// conv32to1_dgrad_kernel and conv32to1_wgrad_kernel<27>: 64-bit index arithmetic per voxel, 27 bounds-tested 4-byte loads of
// the logits gradient per voxel each): 13 + 22 + 25 us at 4 pairs for 11.5 MB read and 11.5 MB written.  Here a workgroup owns
// (image, row y, 26 columns, all D planes): it forms the logits gradient of the (D + 2) x 3 x 28 neighbourhood in LDS (zero
// outside the volume: no bounds tests in the tap loops; the soft-max of a pixel is recomputed by the <= 9 workgroups that
// need it — the logits are 360 KB), then every activation voxel (one 128-byte line, 8 lanes x float4) is read ONCE and serves
// both gradients from the same 27 LDS scalars: 4 FMAs per tap for g_a with the weights from LDS, 4 for g_w into 27 float4
// accumulators that live across the workgroup's units.  g_a is written once, the logits gradient never reaches HBM.
// HBM-bound by design (V read + V written); the arithmetic (216 FMAs per lane and voxel) is ~3 us of the chip at 4 pairs.
#include "as_common.h"

namespace {

struct TailBwdArgs {
  const float* logits;     // [B][D][H][W]
  const float* g_pred;     // [B][H][W] or null
  const float* g_in;       // [B][D][H][W] or null: gradient that reaches the logits directly
  const float* a;          // PCL: the tail's input (layer 4's activation)
  const float* w;          // [32][27]
  float* g_a;              // PCL out (interior voxels written)
  float* partial;          // [workgroups][27 * 32 + 1]
  PclDev g;
  int nxc, units;          // column chunks per row; units = B * H * nxc
};

// columns per unit: 26 (78 = 3 x 26; with all D planes 312 voxels per unit at D = 12) when that still gives every CU more than
// one unit, 13 otherwise (a unit is a chain of seven short phases and voxel passes: at one or four pairs the launch lasts as
// long as one workgroup's chain)
constexpr int kWSmax = 28;             // staged columns of the wider flavour: x0 - 1 .. x0 + XC
constexpr int kMaxGroups = 768;        // three resident workgroups per CU (27 float4 accumulators per lane: 168 registers; 28 KB of LDS)

// LDS floats: sg [(D+2)][3][WS] | se [D][3][WS] | spix [4][3][WS] | sw [27][32]; the closing reduction reuses it from the
// start as [8][864]
__host__ __device__ inline int lds_floats(int D) {
  const int WS = kWSmax;
  const int unit_phase = (((D + 2) * 3 * WS + 3) & ~3) + ((D * 3 * WS + 3) & ~3) + 4 * 3 * WS + 27 * 32;
  const int reduce_phase = 8 * 864 + 8;
  return unit_phase > reduce_phase ? unit_phase : reduce_phase;
}

template <int XC>
__global__ __launch_bounds__(256, 3) void agg_tail_bwd_kernel(TailBwdArgs p) {
  constexpr int WS = XC + 2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int W = p.g.W, H = p.g.H, D = p.g.D;
  const long HW = (long)H * W;
  float* sg = lds;                                          // the logits gradient, zero outside the volume
  float* se = sg + (((D + 2) * 3 * WS + 3) & ~3);           // logits -> exp(l - max) -> p_d, [d][r][xs]
  float* spix = se + ((D * 3 * WS + 3) & ~3);               // per pixel: max, sum of exponentials, pred, g_pred
  float* sw = spix + 4 * 3 * WS;                            // [27][32]: the weights, tap-major (16-byte aligned)
  const int c4 = threadIdx.x & 7, vl = threadIdx.x >> 3;
  for (int i = threadIdx.x; i < 27 * 32; i += 256) {
    const int t = i >> 5, c = i & 31;
    sw[i] = p.w[c * 27 + t];
  }
  // the 27 float4 accumulators of the weight gradient live in registers across the workgroup's units (the weights would be 27
  // more: they spill at two workgroups per CU, so the data gradient reads them from LDS)
  f32x4 accw[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) accw[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float gsum = 0.f;
  const int NP = 3 * WS;                                    // staged pixels
  for (int unit = blockIdx.x; unit < p.units; unit += gridDim.x) {
    const int xc = unit % p.nxc;
    const int t0 = unit / p.nxc;
    const int y = t0 % H, b = t0 / H;
    const int x0 = xc * XC;
    const int xcnt = min(XC, W - x0);
    auto inside = [&](int r, int xs) { const int x = x0 - 1 + xs, yy = y - 1 + r; return x >= 0 && x < W && yy >= 0 && yy < H; };
    __syncthreads();                            // the previous unit's taps have been read
    // ---- soft-argmax backward on the staged pixels: the arithmetic of softargmax_bwd_kernel, term for term, with every
    // exponential and every quotient computed once ----
    // P0: logits and g_pred of rows y-1 .. y+1, columns x0-1 .. x0+XC (row-contiguous reads)
#pragma unroll 1
    for (int i = threadIdx.x; i < D * NP; i += 256) {
      const int d = i / NP, q = i - d * NP, r = q / WS, xs = q - r * WS;
      se[i] = inside(r, xs) ? p.logits[((long)b * D + d) * HW + (long)(y - 1 + r) * W + (x0 - 1 + xs)] : 0.f;
    }
#pragma unroll 1
    for (int q = threadIdx.x; q < NP; q += 256) {
      const int r = q / WS, xs = q - r * WS;
      spix[3 * NP + q] = (p.g_pred && inside(r, xs)) ? p.g_pred[(long)b * HW + (long)(y - 1 + r) * W + (x0 - 1 + xs)] : 0.f;
    }
    __syncthreads();
    // P1: per pixel, the maximum
#pragma unroll 1
    for (int q = threadIdx.x; q < NP; q += 256) {
      float m1 = -INFINITY;
#pragma unroll 1
      for (int d = 0; d < D; ++d) m1 = fmaxf(m1, se[d * NP + q]);
      spix[q] = m1;
    }
    __syncthreads();
    // P2: exp(l - max)
#pragma unroll 1
    for (int i = threadIdx.x; i < D * NP; i += 256) {
      const int d = i / NP, q = i - d * NP;
      se[i] = expf(se[i] - spix[q]);
    }
    __syncthreads();
    // P3: their sum, in d order
#pragma unroll 1
    for (int q = threadIdx.x; q < NP; q += 256) {
      float s = 0.f;
#pragma unroll 1
      for (int d = 0; d < D; ++d) s += se[d * NP + q];
      spix[NP + q] = s;
    }
    __syncthreads();
    // P4: p_d
#pragma unroll 1
    for (int i = threadIdx.x; i < D * NP; i += 256) {
      const int d = i / NP, q = i - d * NP;
      se[i] = se[i] / spix[NP + q];
    }
    __syncthreads();
    // P5: pred = sum_d p_d * d, in d order
#pragma unroll 1
    for (int q = threadIdx.x; q < NP; q += 256) {
      float pr = 0.f;
#pragma unroll 1
      for (int d = 0; d < D; ++d) pr += se[d * NP + q] * (float)d;
      spix[2 * NP + q] = pr;
    }
    __syncthreads();
    // P6: the logits gradient, planes -1 .. D (the two outer planes and everything outside the image: zero)
#pragma unroll 1
    for (int i = threadIdx.x; i < (D + 2) * NP; i += 256) {
      const int j = i / NP, q = i - j * NP, r = q / WS, xs = q - r * WS;
      const int d = j - 1;
      float gv = 0.f;
      if (d >= 0 && d < D && inside(r, xs)) {
        gv = se[d * NP + q] * ((float)d - spix[2 * NP + q]) * spix[3 * NP + q];
        if (p.g_in) gv += p.g_in[((long)b * D + d) * HW + (long)(y - 1 + r) * W + (x0 - 1 + xs)];
      }
      sg[i] = gv;
    }
    __syncthreads();
    // ---- every voxel of the unit: one line of a in, one line of g_a out, 27 taps for both gradients ----
    // (the next voxel's line is requested before this voxel's 216 FMAs: a pass never waits for its own load)
    const int nvox = D * xcnt;
    auto line_of = [&](int vi) { const int d = vi / xcnt; return p.g.vox(b, d, y, x0 + (vi - d * xcnt)) * 32 + c4 * 4; };
    f32x4 a_next = {0.f, 0.f, 0.f, 0.f};
    if (vl < nvox) a_next = *reinterpret_cast<const f32x4*>(p.a + line_of(vl));
    for (int vi = vl; vi < nvox; vi += 32) {
      const int d = vi / xcnt, xl = vi - d * xcnt;
      const long vo = line_of(vi);
      const f32x4 a4 = a_next;
      if (vi + 32 < nvox) a_next = *reinterpret_cast<const f32x4*>(p.a + line_of(vi + 32));
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const float* s0 = sg + ((d + 2) * 3 + 2) * WS + (xl + 2);       // tap (kd, kh, kw) reads s0[-(kd * 3 + kh) * WS - kw]
      // Nine rows of three taps.  A row's 3 + 3 LDS reads are issued one row ahead of its 24 FMAs, and the FMAs are tied to
      // their place (their results pass through an empty asm): left alone, the compiler issues the reads of all nine rows
      // first, into 150 registers that live until the FMAs at the end of the pass, and the 27 accumulators spill.
      float glc[3], gln[3];
      f32x4 wqc[3], wqn[3];
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        glc[kw] = s0[-kw];
        wqc[kw] = *reinterpret_cast<const f32x4*>(sw + kw * 32 + c4 * 4);
      }
#pragma unroll
      for (int row = 0; row < 9; ++row) {               // row = kd * 3 + kh
        if (row < 8) {
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            gln[kw] = s0[-(row + 1) * WS - kw];
            wqn[kw] = *reinterpret_cast<const f32x4*>(sw + ((row + 1) * 3 + kw) * 32 + c4 * 4);
          }
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int tp = row * 3 + kw;
          const float gl = glc[kw];
          const f32x4 wq = wqc[kw];
          acc.x = fmaf(gl, wq.x, acc.x); acc.y = fmaf(gl, wq.y, acc.y);
          acc.z = fmaf(gl, wq.z, acc.z); acc.w = fmaf(gl, wq.w, acc.w);
          accw[tp].x = fmaf(gl, a4.x, accw[tp].x); accw[tp].y = fmaf(gl, a4.y, accw[tp].y);
          accw[tp].z = fmaf(gl, a4.z, accw[tp].z); accw[tp].w = fmaf(gl, a4.w, accw[tp].w);
        }
        {
          f32x4& w0 = accw[row * 3], &w1 = accw[row * 3 + 1], &w2 = accw[row * 3 + 2];
          asm volatile("" : "+v"(acc.x), "+v"(acc.y), "+v"(acc.z), "+v"(acc.w), "+v"(w0.x), "+v"(w0.y), "+v"(w0.z), "+v"(w0.w),
                            "+v"(w1.x), "+v"(w1.y), "+v"(w1.z), "+v"(w1.w), "+v"(w2.x), "+v"(w2.y), "+v"(w2.z), "+v"(w2.w));
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) { glc[kw] = gln[kw]; wqc[kw] = wqn[kw]; }
      }
      *reinterpret_cast<f32x4*>(p.g_a + vo) = acc;
      if (c4 == 0) gsum += sg[((d + 1) * 3 + 1) * WS + (xl + 1)];
    }
  }
  // ---- the workgroup's slab.  Lanes i, i ^ 8, i ^ 16, i ^ 24 hold the same channels of four voxel slots: two shuffles add
  // them; the eight half-waves of the workgroup then meet in LDS (fixed order) ----
  __syncthreads();
  float* red = lds;                                         // [8][864]
  const int lane = threadIdx.x & 63, slot = (threadIdx.x >> 6) * 2 + (lane >> 5);
#pragma unroll
  for (int tp = 0; tp < 27; ++tp) {
    f32x4 v = accw[tp];
    v.x += __shfl_xor(v.x, 8, 64); v.y += __shfl_xor(v.y, 8, 64); v.z += __shfl_xor(v.z, 8, 64); v.w += __shfl_xor(v.w, 8, 64);
    v.x += __shfl_xor(v.x, 16, 64); v.y += __shfl_xor(v.y, 16, 64); v.z += __shfl_xor(v.z, 16, 64); v.w += __shfl_xor(v.w, 16, 64);
    if ((lane & 24) == 0) *reinterpret_cast<f32x4*>(red + slot * 864 + tp * 32 + c4 * 4) = v;
  }
  const float gs = wave_sum(c4 == 0 ? gsum : 0.f);
  if (lane == 0) red[8 * 864 + (threadIdx.x >> 6)] = gs;
  __syncthreads();
  float* out = p.partial + (long)blockIdx.x * (27 * 32 + 1);
  for (int o = threadIdx.x; o < 864; o += 256) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k * 864 + o];
    out[o] = s;
  }
  if (threadIdx.x == 0) out[864] = (red[8 * 864] + red[8 * 864 + 1]) + (red[8 * 864 + 2] + red[8 * 864 + 3]);
}

inline size_t lds_bytes(int D) { return (size_t)lds_floats(D) * sizeof(float); }
}  // namespace

// slab reduction of the 32->1 weight gradient (softargmax.hip)
int as_conv32to1_wgrad_reduce(const float* partial, int nblocks, int ntaps, float* g_w, float* g_bias, int accumulate, hipStream_t st);

extern "C" int as_agg_tail_bwd_ok(const as_pcl* g) {
  return as_pcl_ok(g) && g->pd >= 1 && g->ph >= 1 && g->pw >= 1 && lds_bytes(g->D) <= 64 * 1024 &&
                 (long)g->B * g->H * ((g->W + 12) / 13) < (1L << 31) ? 1 : 0;
}

extern "C" int64_t as_agg_tail_bwd_workspace(const as_pcl* g) {
  if (!as_pcl_ok(g)) return -1;
  return (int64_t)kMaxGroups * (27 * 32 + 1);
}

// logits [B][D][H][W] (saved by the forward pass); g_pred [B][H][W] and g_logits_in [B][D][H][W] may each be NULL (zero);
// a, g_a: PCL of geometry g (halo >= 1); w [32][27]; g_w [32][27] and g_bias [1] are overwritten, or added to with accumulate.
extern "C" int as_agg_tail_bwd(const float* logits, const float* g_pred, const float* g_logits_in, const float* a,
                               const as_pcl* g, const float* w, float* g_a, float* g_w, float* g_bias, int accumulate,
                               float* workspace, void* stream) {
  AS_CHECK_ARG(as_agg_tail_bwd_ok(g) == 1, "as_agg_tail_bwd: geometry not supported (halo >= 1, at most ~100 disparity planes)");
  AS_CHECK_ARG(logits && a && w && g_a && g_w && workspace, "as_agg_tail_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  TailBwdArgs p;
  p.logits = logits; p.g_pred = g_pred; p.g_in = g_logits_in; p.a = a; p.w = w; p.g_a = g_a; p.partial = workspace;
  p.g = as_make_dev(g);
  const bool wide = (long)g->B * g->H * ((g->W + 25) / 26) > kMaxGroups;
  const int XC = wide ? 26 : 13;
  p.nxc = (g->W + XC - 1) / XC;
  p.units = g->B * g->H * p.nxc;
  const int groups = p.units < kMaxGroups ? p.units : kMaxGroups;
  const double vox = (double)g->B * g->D * g->H * g->W;
  as_prof_mark(AS_PROF_OUTCONV_BWD, st, 1, 0.0);
  if (wide) hipLaunchKernelGGL(agg_tail_bwd_kernel<26>, dim3(groups), dim3(256), lds_bytes(g->D), st, p);
  else hipLaunchKernelGGL(agg_tail_bwd_kernel<13>, dim3(groups), dim3(256), lds_bytes(g->D), st, p);
  // a read, g_a written, logits and g_pred read
  as_prof_mark(AS_PROF_OUTCONV_BWD, st, 0, (2 * 128.0 + 4.0) * vox + 4.0 * (double)g->B * g->H * g->W);
  AS_CHECK_LAUNCH("as_agg_tail_bwd");
  return as_conv32to1_wgrad_reduce(workspace, groups, 27, g_w, g_bias, accumulate, st);
}
