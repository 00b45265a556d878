// a1, the 1/16-resolution trunk of FeatureExtractorNetwork: six BasicBlocks  a_l = lrelu(BN(conv3x3(a_{l-1}))) + a_{l-1}
// and conv_alone (reference: adaptive_stereo/models/stereo_net.py:33-51, 79-85), forward and backward, train mode,
// ONE launch per layer and direction for BOTH images of a pair.
//
// Why.  A 24x78 map is 59 wave tiles: every kernel of the trunk is a chain of dependent latencies on a mostly idle chip
// (profiles/r03_pmc_small_b1_baseline.json: the split-K convolution keeps 477 waves alive for 7.6 us to do 1 us of matrix
// work; BatchNorm finalize / activation / backward-reduce / -finalize / -apply are 4-7 us launches of a few waves).  Up to
// round 2 a BasicBlock cost 3 launches forward and 5 backward, per image: ~110 launches per step on two streams.  Here the
// element-wise BatchNorm work rides on the operand staging of the NEXT convolution (forward) / of the data gradient
// (backward), the BatchNorm reductions are merged by the consumer from per-workgroup partials (no finalize launch), the
// weight gradient shares the data gradient's staged tile, and the two images of a pair are two STATISTICS GROUPS of one
// launch (the reference calls feature_net twice, adapt.py:72: batch statistics stay per image): 7 + 7 launches per step.
// A kernel boundary (1.5-1.9 us in a replayed graph) is the grid-wide seam between layers; MI355X_MICROARCH.md prices an
// in-launch grid barrier at 4-5 us plus an L2 write-back, so the seams stay kernel boundaries.
//
// Tile = 32 consecutive x of one row (the last tile of a row is shifted left to end at W; its overlap is recomputed with
// identical values and left out of every reduction).  A workgroup stages the 3 x 34 voxels around its tile in LDS
// (swizzled 16-byte chunks: conflict-free ds_read_b128 for the matrix A operand AND ds_read_b32 for the weight gradient's
// B operand), its four waves split the nine taps (split-K: the chain of dependent MFMAs is 48 long instead of 144).
#include "as_common.h"
#include "conv_epilogue.h"
#include "trunk.h"
#include <cstdlib>

#define TR_COLS 34
#define TR_STAGE_VOX (3 * TR_COLS)          // 102 voxels
#define TR_OP_FLOATS (TR_STAGE_VOX * 32)    // 3264 floats
#define TR_JOBS (TR_STAGE_VOX * 8)          // 816 sixteen-byte chunks
#ifndef TR_GPER_MAX
#define TR_GPER_MAX 128                 // 2 groups x 128 workgroups = one per CU (the backward kernel holds 450 registers: no second workgroup fits a CU; 144 per group ran 50 % slower)
#endif

struct TrunkGeom {
  PclDev g;
  int ngroups, imgs_per_group;
  int tiles_per_row, tiles_per_group;
  int gper;                                  // workgroups per statistics group
};

struct TileId { int b, y, x0, dup, nvalid; };

__device__ inline TileId trunk_tile(const TrunkGeom& tg, int group, int t) {
  const int per_img = tg.g.H * tg.tiles_per_row;
  const int img = t / per_img, rem = t - img * per_img;
  const int y = rem / tg.tiles_per_row, j = rem - y * tg.tiles_per_row;
  TileId id;
  id.b = group * tg.imgs_per_group + img;
  id.y = y;
  id.x0 = min(32 * j, max(tg.g.W - 32, 0));
  id.dup = 32 * j - id.x0;                                        // lanes below this repeat the previous tile's voxels
  id.nvalid = min(tg.g.W, 32 * (j + 1)) - 32 * j;
  return id;
}

__device__ inline int op_addr(int sv, int chunk) { return sv * 32 + ((chunk ^ ((sv >> 1) & 7)) << 2); }

__device__ inline f32x4 lrelu_sel(f32x4 y, f32x4 v, float slope) {   // v where y > 0, v*slope elsewhere
  f32x4 r;
  r.x = y.x > 0.f ? v.x : v.x * slope; r.y = y.y > 0.f ? v.y : v.y * slope;
  r.z = y.z > 0.f ? v.z : v.z * slope; r.w = y.w > 0.f ? v.w : v.w * slope;
  return r;
}

__device__ inline void mfma16t(f32x16& acc, const f32x4 (&a)[4], const f32x4 (&b)[4]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, b[q].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, b[q].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, b[q].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, b[q].w, acc, 0, 0, 0);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// BatchNorm finalize by the consumer, per statistics group (bn_merge.h's one-pass fp64 merge with a pivot; see there).
// All 256 threads call.  tab[0..31] = scale, tab[32..63] = shift.  The group's first workgroup publishes
// state[group] = {mean, invstd, scale, shift, unbiased variance} (backward and the running-statistics update read it).
struct TrunkBnIn {
  const float* stat_mean;    // [ngroups][nparts][32]
  const float* stat_m2;
  const float* stat_cnt;     // [ngroups][nparts]
  const float* gamma;
  const float* beta;
  float* state;              // [ngroups][5][32]
  int nparts;
  float eps;
};

__device__ inline void trunk_bn_merge(const TrunkBnIn& m, int group, double* red /*[8][32][3]*/, float* tab, bool publish) {
  const int c = threadIdx.x & 31, slc = threadIdx.x >> 5;
  const float* sm = m.stat_mean + (long)group * m.nparts * 32;
  const float* s2 = m.stat_m2 + (long)group * m.nparts * 32;
  const float* sc = m.stat_cnt + (long)group * m.nparts;
  const int per_slice = (m.nparts + 7) >> 3;
  const double K = (double)sm[c];
  double s0 = 0.0, s1 = 0.0, sq = 0.0;
  constexpr int BATCH = 12;                              // partials per thread whose loads are in flight together (96 per group per round)
  for (int j0 = 0; j0 < per_slice; j0 += BATCH) {
    float pn[BATCH], pm[BATCH], pq[BATCH];
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const int i = slc + 8 * (j0 + j);
      const bool ok = i < m.nparts;
      const int ii = ok ? i : 0;
      pn[j] = ok ? sc[ii] : 0.f;
      pm[j] = sm[ii * 32 + c];
      pq[j] = ok ? s2[ii * 32 + c] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const double n = (double)pn[j], dm = (double)pm[j] - K;
      s0 += n; s1 += n * dm; sq += (double)pq[j] + n * dm * dm;
    }
  }
  double* mine = red + (slc * 32 + c) * 3;
  mine[0] = s0; mine[1] = s1; mine[2] = sq;
  __syncthreads();
  if (slc == 0) {
    double t0 = 0.0, t1 = 0.0, t2 = 0.0;
    for (int j = 0; j < 8; ++j) { const double* r = red + (j * 32 + c) * 3; t0 += r[0]; t1 += r[1]; t2 += r[2]; }
    const double count = t0;
    const double mean = K + t1 / count;
    const double m2 = fmax(t2 - t1 * t1 / count, 0.0);
    const double var_b = m2 / count;
    const float invstd = (float)(1.0 / sqrt(var_b + (double)m.eps));
    const float meanf = (float)mean;
    const float scl = invstd * m.gamma[c];
    const float shf = m.beta[c] - meanf * scl;
    tab[c] = scl; tab[32 + c] = shf;
    if (publish) {
      float* st = m.state + (long)group * 160;
      st[c] = meanf; st[32 + c] = invstd; st[64 + c] = scl; st[96 + c] = shf;
      st[128 + c] = (float)(count > 1.0 ? m2 / (count - 1.0) : var_b);
    }
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------------
// Forward layer.  MODE 0: the operand is `src` as it is (first block: the head's output).  MODE 1: the operand is
// a_{l-1} = lrelu(BN_{l-1}(src = z_{l-1})) + skip (= a_{l-2}), formed while staging; the tile's own voxels of a_{l-1} are
// written to a_out (the next skip connection and the backward pass read it).
// Diagnostic build (make EXTRA=-DTR_TIMING_BUILD): wave 0 of every workgroup stamps the shader clock at the phase boundaries
// of its FIRST tile; tests/tools/trunk_timing.py prints the averages.  Never for production.
#ifdef TR_TIMING_BUILD
__device__ long long* g_tr_timing = nullptr;
#define TR_T(slot) do { if (threadIdx.x == 0 && tstamp) tstamp[slot] = wall_clock64(); } while (0)
#else
#define TR_T(slot) do { } while (0)
#endif

struct TrunkFwdArgs {
  const float* src;
  const float* skip;
  float* a_out;
  TrunkBnIn bn;
  const float* wp;
  const float* bias;
  float* z;
  float* stat_mean;          // [ngroups * gper][32] (+ m2, cnt) or null
  float* stat_m2;
  float* stat_cnt;
  TrunkGeom tg;
  float slope;
#ifdef TR_TIMING_BUILD
  long long* timing;
#endif
};

template <int MODE>
__global__ __launch_bounds__(256) void trunk_fwd_kernel(TrunkFwdArgs p) {
  __shared__ __attribute__((aligned(16))) float op[TR_OP_FLOATS];
  __shared__ float part[3][16][64];
  __shared__ float red[4][32];
  __shared__ float bmean[32];
  __shared__ double mred[8 * 32 * 3];
  __shared__ float tab[64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, li = lane & 31;
  const int group = blockIdx.x / p.tg.gper, wi = blockIdx.x - group * p.tg.gper;
  const PclDev g = p.tg.g;
#ifdef TR_TIMING_BUILD
  long long* tstamp = p.timing ? p.timing + (long)blockIdx.x * 8 : nullptr;
#endif
  TR_T(0);

  // this wave's taps (wave, wave + 4, wave + 8): B fragments resident for the whole launch; requested first — they depend on
  // nothing, the merge and the staging loads below overlap their latency
  f32x4 bw[3][4];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int tp = wave + 4 * k;
    if (tp < 9) {
      const float* wb = p.wp + tp * 1024 + lane * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) bw[k][q] = *reinterpret_cast<const f32x4*>(wb + q * 256);
    }
  }
  const float bias_v = wave == 0 ? p.bias[li] : 0.f;

  // The staging loads of a tile (16-byte chunks; 816 jobs over 256 threads) are requested one tile ahead: those of the first
  // tile before the BatchNorm merge (they depend on nothing — the merge's round trips and theirs overlap), those of tile
  // t + gper behind the barrier that publishes tile t's operand, in flight during its matrix phase and epilogue.
  f32x4 q[4], r[4];
  long off[4];
  auto request = [&](int t) {
    const TileId id = trunk_tile(p.tg, group, t);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int job = threadIdx.x + 256 * k;
      const int sv = min(job, TR_JOBS - 1) >> 3, c = job & 7;
      const int row = sv / TR_COLS, col = sv - row * TR_COLS;
      const int yc = min(max(id.y + row - 1, 0), g.H - 1), xc = min(max(id.x0 + col - 1, 0), g.W - 1);
      off[k] = g.vox(id.b, 0, yc, xc) * 32 + c * 4;
      q[k] = *reinterpret_cast<const f32x4*>(p.src + off[k]);
      if (MODE == 1) r[k] = *reinterpret_cast<const f32x4*>(p.skip + off[k]);
    }
  };
  request(wi);
  TR_T(1);
  if (MODE == 1) trunk_bn_merge(p.bn, group, mred, tab, wi == 0);
  TR_T(2);

  TileStats run; run.n = 0.f; run.mean = 0.f; run.m2 = 0.f;
  for (int t = wi; t < p.tg.tiles_per_group; t += p.tg.gper) {
    const TileId id = trunk_tile(p.tg, group, t);
    // ---- stage 3 x 34 voxels ------------------------------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int job = threadIdx.x + 256 * k;
      const int sv = job >> 3, c = job & 7;
      const int svc = min(job, TR_JOBS - 1) >> 3;
      const int row = svc / TR_COLS, col = svc - row * TR_COLS;
      const int yy = id.y + row - 1, xx = id.x0 + col - 1;
      const bool inside = job < TR_JOBS && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
      const bool own = inside && row == 1 && col >= 1 && col <= 32;
      f32x4 v = q[k];
      if (MODE == 1) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(tab + c * 4);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(tab + 32 + c * 4);
        f32x4 y = q[k] * sc + sh;
        y = lrelu_sel(y, y, p.slope);
        v = y + r[k];
        if (own) *reinterpret_cast<f32x4*>(p.a_out + off[k]) = v;
      }
      if (!inside) v = f32x4{0.f, 0.f, 0.f, 0.f};              // zero padding (a BatchNorm'd halo would be lrelu(shift))
      if (job < TR_JOBS) *reinterpret_cast<f32x4*>(op + op_addr(sv, c)) = v;
    }
    __syncthreads();
    if (t == wi) TR_T(3);
    if (t + p.tg.gper < p.tg.tiles_per_group) request(t + p.tg.gper);

    // ---- split-K matrix phase: wave w multiplies taps w, w+4, w+8 ------------------------------------------------------
    f32x16 acc;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) acc[rr] = bias_v;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int tp = wave + 4 * k;                              // wave-uniform
      if (tp < 9) {
        const int kh = tp / 3, kw = tp - 3 * kh;
        const int sv = kh * TR_COLS + li + kw;
        f32x4 a[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) a[qq] = *reinterpret_cast<const f32x4*>(op + op_addr(sv, h * 4 + qq));
        mfma16t(acc, a, bw[k]);
      }
    }
    if (t == wi) TR_T(4);
    if (wave > 0) {
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) part[wave - 1][rr][lane] = acc[rr];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) acc[rr] = ((acc[rr] + part[0][rr][lane]) + part[1][rr][lane]) + part[2][rr][lane];
    }
    if (t == wi) TR_T(5);
    const int x = id.x0 + li;
    const bool valid = wave == 0 && li >= id.dup && x < g.W;
    const int out_vox = (int)g.vox(id.b, 0, id.y, min(x, g.W - 1));
    EpilogueArgs ep;
    ep.bias = nullptr; ep.z = p.z; ep.ep_scale = nullptr; ep.ep_shift = nullptr; ep.residual = nullptr;
    ep.stat_mean = p.stat_mean; ep.stat_m2 = p.stat_m2; ep.stat_cnt = p.stat_cnt; ep.epilogue = 0; ep.slope = p.slope;
    TileStats ts; ts.n = 0.f; ts.mean = 0.f; ts.m2 = 0.f;
    conv_epilogue(acc, ep, out_vox, valid, id.nvalid, red, bmean, &ts);
    if (p.stat_mean != nullptr && threadIdx.x < 32) stats_merge(run, ts);
    __syncthreads();                                            // op / part / red are rewritten by the next tile
    if (t == wi) TR_T(6);
  }
  TR_T(7);
  if (p.stat_mean != nullptr && threadIdx.x < 32) {
    const int idx = group * p.tg.gper + wi;
    p.stat_mean[idx * 32 + threadIdx.x] = run.mean;
    p.stat_m2[idx * 32 + threadIdx.x] = run.m2;
    if (threadIdx.x == 0) p.stat_cnt[idx] = run.n;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward layer.  MODE 1 (a BasicBlock):  g_z = BN-backward(lrelu'(y) g_a) formed while staging (stage-1 sums merged from
// the previous launch's per-workgroup partials), g_x = g_a + dgrad(g_z), dW/db slabs, and — for the layer below — the
// per-workgroup stage-1 sums of ITS BatchNorm backward taken from the g_x tile in registers.  MODE 0 (conv_alone): the
// staged gradient is g_a itself, no skip connection.
struct TrunkBwdArgs {
  const float* g_a;
  const float* z;
  const float* state;        // [ngroups][5][32] of this layer's BatchNorm
  const double* sums;        // [ngroups][nparts][64]: stage-1 partial sums of this layer's BatchNorm backward
  const float* gamma;
  float* bn_grads;           // [ngroups][2][32]: this launch's (g_gamma, g_beta) per group
  const float* x;            // a_{l-1}: the convolution's input
  const float* wp_t;
  float* g_x;
  const float* z_next;       // null: no stage-1 sums for the layer below
  const float* state_next;
  double* sums_next;         // [ngroups * gper][64]
  float* slab;               // [ngroups * gper][9][32][32]
  float* slab_db;            // [ngroups * gper][32]
  TrunkGeom tg;
  int nparts;
  float slope;
  float count;               // voxels per statistics group
};

template <int MODE>
__global__ __launch_bounds__(256) void trunk_bwd_kernel(TrunkBwdArgs p) {
  __shared__ __attribute__((aligned(16))) float op[TR_OP_FLOATS];
  __shared__ float part[3][16][64];
  __shared__ double sred[4][64];
  __shared__ float tab[6 * 32];          // k1, k2, k3, scale, shift, mean of this layer's BatchNorm
  __shared__ float ntab[3 * 32];         // scale, shift, mean of the layer below
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, li = lane & 31;
  const int group = blockIdx.x / p.tg.gper, wi = blockIdx.x - group * p.tg.gper;
  const PclDev g = p.tg.g;

  f32x4 bw[3][4];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int tp = wave + 4 * k;
    if (tp < 9) {
      const float* wb = p.wp_t + tp * 1024 + lane * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) bw[k][q] = *reinterpret_cast<const f32x4*>(wb + q * 256);
    }
  }

  // loads one tile ahead, as in the forward kernel: the first tile's before the merge of the stage-1 sums (their round
  // trips overlap), those of tile t + gper in flight during tile t's matrix phase and epilogue
  float xr[16];
  f32x4 q[4], r[4];
  auto request = [&](int t) {
    const TileId id = trunk_tile(p.tg, group, t);
    // the weight gradient's A operand: the tile's own voxels of x, 16 voxel pairs (lane half h = voxel parity), channel li
    const float* xrow = p.x + g.vox(id.b, 0, id.y, 0) * 32 + li;
#pragma unroll
    for (int s = 0; s < 16; ++s) xr[s] = xrow[(long)min(id.x0 + 2 * s + h, g.W - 1) * 32];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int job = threadIdx.x + 256 * k;
      const int sv = min(job, TR_JOBS - 1) >> 3, c = job & 7;
      const int row = sv / TR_COLS, col = sv - row * TR_COLS;
      const int yc = min(max(id.y + row - 1, 0), g.H - 1), xc = min(max(id.x0 + col - 1, 0), g.W - 1);
      const long off = g.vox(id.b, 0, yc, xc) * 32 + c * 4;
      q[k] = *reinterpret_cast<const f32x4*>(p.g_a + off);
      if (MODE == 1) r[k] = *reinterpret_cast<const f32x4*>(p.z + off);
    }
  };
  request(wi);

  if (MODE == 1) {
    // stage 2 of the BatchNorm backward by the consumer: fixed-order fp64 sum of the group's partials
    const int j = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const double* src = p.sums + (long)group * p.nparts * 64 + j;
    double s = 0.0;
    for (int i0 = sl; i0 < p.nparts; i0 += 32) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int i = i0 + 4 * u; v[u] = i < p.nparts ? src[(long)i * 64] : 0.0; }
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    sred[sl][j] = s;
    __syncthreads();
    if (threadIdx.x < 32) {
      const int c = threadIdx.x;
      const double sdy = ((sred[0][c] + sred[1][c]) + sred[2][c]) + sred[3][c];
      const double sdx = ((sred[0][32 + c] + sred[1][32 + c]) + sred[2][32 + c]) + sred[3][32 + c];
      const float* st = p.state + (long)group * 160;
      const double is = (double)st[32 + c];
      tab[c] = (float)(sdy / (double)p.count);
      tab[32 + c] = (float)(sdx * is * is / (double)p.count);
      tab[64 + c] = st[32 + c] * p.gamma[c];
      tab[96 + c] = st[64 + c]; tab[128 + c] = st[96 + c]; tab[160 + c] = st[c];
      if (wi == 0) {
        p.bn_grads[(group * 2 + 0) * 32 + c] = (float)(sdx * is);
        p.bn_grads[(group * 2 + 1) * 32 + c] = (float)sdy;
      }
    }
  }
  if (p.z_next != nullptr && threadIdx.x < 32) {
    const float* st = p.state_next + (long)group * 160;
    ntab[threadIdx.x] = st[64 + threadIdx.x]; ntab[32 + threadIdx.x] = st[96 + threadIdx.x]; ntab[64 + threadIdx.x] = st[threadIdx.x];
  }
  __syncthreads();

  f32x16 wacc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) wacc[k][rr] = 0.f;
  float db_run = 0.f;
  double run_dy = 0.0, run_dx = 0.0;


  for (int t = wi; t < p.tg.tiles_per_group; t += p.tg.gper) {
    const TileId id = trunk_tile(p.tg, group, t);
    float xa[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int u = 2 * s + h;
      xa[s] = (u >= id.dup && id.x0 + u < g.W) ? xr[s] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int job = threadIdx.x + 256 * k;
      const int sv = job >> 3, c = job & 7;
      const int svc = min(job, TR_JOBS - 1) >> 3;
      const int row = svc / TR_COLS, col = svc - row * TR_COLS;
      const int yy = id.y + row - 1, xx = id.x0 + col - 1;
      const bool inside = job < TR_JOBS && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
      f32x4 v = q[k];
      if (MODE == 1) {
        const f32x4 k1 = *reinterpret_cast<const f32x4*>(tab + c * 4);
        const f32x4 k2 = *reinterpret_cast<const f32x4*>(tab + 32 + c * 4);
        const f32x4 k3 = *reinterpret_cast<const f32x4*>(tab + 64 + c * 4);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(tab + 96 + c * 4);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(tab + 128 + c * 4);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(tab + 160 + c * 4);
        const f32x4 gg = lrelu_sel(r[k] * sc + sh, q[k], p.slope);
        const f32x4 dx = (r[k] - mu) * k2;
        v = (gg - k1 - dx) * k3;
      }
      if (!inside) v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (job < TR_JOBS) *reinterpret_cast<f32x4*>(op + op_addr(sv, c)) = v;
    }
    __syncthreads();
    if (t + p.tg.gper < p.tg.tiles_per_group) request(t + p.tg.gper);

    // ---- data gradient (split-K over the taps of the mirrored kernel) and weight gradient ------------------------------
    f32x16 acc;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) acc[rr] = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int tp = wave + 4 * k;
      if (tp < 9) {
        const int kh = tp / 3, kw = tp - 3 * kh;
        const int sv = kh * TR_COLS + li + kw;
        f32x4 a[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) a[qq] = *reinterpret_cast<const f32x4*>(op + op_addr(sv, h * 4 + qq));
        mfma16t(acc, a, bw[k]);
      }
    }
    // dW[kh][kw][ci][co] += sum_u x[u][ci] * g_z[u - (kh-1, kw-1)][co]: the tile's own x voxels meet the staged g_z rows
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int tp = wave + 4 * k;
      if (tp < 9) {
        const int kh = tp / 3, kw = tp - 3 * kh;
        const int base = (2 - kh) * TR_COLS + 2 - kw + h;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const int sv = base + 2 * s;
          const float bv = op[op_addr(sv, li >> 2) + (li & 3)];
          wacc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[s], bv, wacc[k], 0, 0, 0);
        }
      }
    }
    if (wave == 3) {
      // db += sum over the tile's own voxels of g_z (centre row, staged columns 1..32)
      float sdb = 0.f;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int u = 2 * s + h;
        const float v = op[op_addr(TR_COLS + 1 + u, li >> 2) + (li & 3)];
        sdb += (u >= id.dup && id.x0 + u < g.W) ? v : 0.f;
      }
      sdb += __shfl_xor(sdb, 32, 64);
      db_run += sdb;
    }
    if (wave > 0) {
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) part[wave - 1][rr][lane] = acc[rr];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) acc[rr] = ((acc[rr] + part[0][rr][lane]) + part[1][rr][lane]) + part[2][rr][lane];
      const int x = id.x0 + li;
      const bool valid = li >= id.dup && x < g.W;
      const int out_vox = (int)g.vox(id.b, 0, id.y, min(x, g.W - 1));
      int ov[16], rv[16];
      float res[16], zn[16];
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int row = (rr & 3) + 8 * (rr >> 2) + 4 * h;
        ov[rr] = __shfl(out_vox, row, 64);
        rv[rr] = __shfl((int)valid, row, 64);
      }
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        res[rr] = MODE == 1 ? p.g_a[(long)ov[rr] * 32 + li] : 0.f;
        zn[rr] = p.z_next != nullptr ? p.z_next[(long)ov[rr] * 32 + li] : 0.f;
      }
      float s_dy = 0.f, s_dx = 0.f;
      const float scn = ntab[li], shn = ntab[32 + li], mun = ntab[64 + li];
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const float gx = acc[rr] + res[rr];
        if (rv[rr]) {
          p.g_x[(long)ov[rr] * 32 + li] = gx;
          const float yn = zn[rr] * scn + shn;
          const float gg = yn > 0.f ? gx : gx * p.slope;
          s_dy += gg;
          s_dx += gg * (zn[rr] - mun);
        }
      }
      s_dy += __shfl_xor(s_dy, 32, 64);
      s_dx += __shfl_xor(s_dx, 32, 64);
      run_dy += (double)s_dy; run_dx += (double)s_dx;
    }
    __syncthreads();
  }

  // ---- this workgroup's slabs: taps are disjoint across the waves, no reduction needed -----------------------------------
  const int idx = group * p.tg.gper + wi;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int tp = wave + 4 * k;
    if (tp < 9) {
      float* out = p.slab + ((long)idx * 9 + tp) * 1024;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int ci = (rr & 3) + 8 * (rr >> 2) + 4 * h;
        out[ci * 32 + li] = wacc[k][rr];
      }
    }
  }
  if (wave == 3 && h == 0) p.slab_db[idx * 32 + li] = db_run;
  if (wave == 0 && h == 0 && p.sums_next != nullptr) {
    p.sums_next[(long)idx * 64 + li] = run_dy;
    p.sums_next[(long)idx * 64 + 32 + li] = run_dx;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// End of the forward pass: PCL -> NCHW (the public layout of the features) and the running statistics of every BatchNorm of
// the trunk, group by group in call order (the reference calls feature_net(left) first, then feature_net(right):
// running = (1 - m) * running + m * batch, twice).
struct TrunkFinishArgs {
  const float* feats_pcl;
  float* feats_nchw;
  PclDev g;
  const float* states;       // [nlayers][ngroups][5][32]
  float* rm[TRUNK_MAX_LAYERS];
  float* rv[TRUNK_MAX_LAYERS];
  int nlayers, ngroups;
  float momentum;
};

__global__ __launch_bounds__(256) void trunk_finish_fwd_kernel(TrunkFinishArgs p) {
  __shared__ float tile[32][33];
  const PclDev g = p.g;
  const int rows = g.B * g.H, segs = (g.W + 31) / 32;
  if ((int)blockIdx.x == rows * segs) {                        // one extra workgroup: running statistics
    for (int i = threadIdx.x; i < p.nlayers * 32; i += 256) {
      const int l = i >> 5, c = i & 31;
      double m = (double)p.rm[l][c], v = (double)p.rv[l][c];
      const double mo = (double)p.momentum;
      for (int gi = 0; gi < p.ngroups; ++gi) {
        const float* st = p.states + ((long)l * p.ngroups + gi) * 160;
        m = (double)(float)(mo * (double)st[c] + (1.0 - mo) * m);
        v = (double)(float)(mo * (double)st[128 + c] + (1.0 - mo) * v);
      }
      p.rm[l][c] = (float)m; p.rv[l][c] = (float)v;
    }
    return;
  }
  const int row = blockIdx.x / segs, seg = blockIdx.x - row * segs;
  const int b = row / g.H, y = row - b * g.H, x0 = seg * 32;
  const int c = threadIdx.x & 31, v0 = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int u = v0 + 8 * k, x = x0 + u;
    tile[u][c] = x < g.W ? p.feats_pcl[g.vox(b, 0, y, x) * 32 + c] : 0.f;
  }
  __syncthreads();
  const int u = threadIdx.x & 31, c0 = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int cc = c0 + 8 * k, x = x0 + u;
    if (x < g.W) p.feats_nchw[(((long)b * 32 + cc) * g.H + y) * g.W + x] = tile[u][cc];
  }
}

// Start of the backward pass: the gradient(s) of the features, NCHW as autograd hands them over — one tensor, or the left
// and the right images' separately (the two outputs of a pair pass) — into the interior of one PCL buffer.
struct TrunkBeginArgs {
  const float* ga;           // NCHW [nA][32][H][W]
  const float* gb;           // NCHW [B - nA][32][H][W] or null
  int nA;
  float* out;                // PCL, geometry g (batch B)
  PclDev g;
};

__global__ __launch_bounds__(256) void trunk_begin_bwd_kernel(TrunkBeginArgs p) {
  __shared__ float tile[32][33];
  const PclDev g = p.g;
  const int segs = (g.W + 31) / 32;
  const int row = blockIdx.x / segs, seg = blockIdx.x - row * segs;
  const int b = row / g.H, y = row - b * g.H, x0 = seg * 32;
  const float* src = b < p.nA ? p.ga + (long)b * 32 * g.H * g.W : p.gb + (long)(b - p.nA) * 32 * g.H * g.W;
  const int u = threadIdx.x & 31, c0 = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int cc = c0 + 8 * k, x = x0 + u;
    tile[u][cc] = x < g.W ? src[((long)cc * g.H + y) * g.W + x] : 0.f;
  }
  __syncthreads();
  const int c = threadIdx.x & 31, v0 = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int uu = v0 + 8 * k, x = x0 + uu;
    if (x < g.W) p.out[g.vox(b, 0, y, x) * 32 + c] = tile[uu][c];
  }
}

// End of the backward pass: the per-group BatchNorm parameter gradients of every layer into their destinations, groups in
// call order.
struct TrunkBnGradArgs {
  const float* bn_grads;     // [nlayers][ngroups][2][32]
  float* g_gamma[TRUNK_MAX_LAYERS];
  float* g_beta[TRUNK_MAX_LAYERS];
  int nlayers, ngroups, accumulate;
};

__global__ void trunk_finish_bwd_kernel(TrunkBnGradArgs p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.nlayers * 64) return;
  const int l = i >> 6, which = (i >> 5) & 1, c = i & 31;
  float* dst = (which ? p.g_beta[l] : p.g_gamma[l]) + c;
  float v = p.accumulate ? *dst : 0.f;
  for (int gi = 0; gi < p.ngroups; ++gi) v += p.bn_grads[(((long)l * p.ngroups + gi) * 2 + which) * 32 + c];
  *dst = v;
}

// ---------------------------------------------------------------------------------------------------------------------
// Host side
static int trunk_geom(TrunkGeom* tg, const as_pcl* g, int ngroups, const char* who) {
  AS_CHECK_ARG(as_pcl_ok(g) && g->D == 1 && g->pd == 0 && g->ph >= 1 && g->pw >= 1, "%s: needs a 2-D PCL geometry with a halo", who);
  AS_CHECK_ARG(ngroups >= 1 && ngroups <= TRUNK_MAX_GROUPS && g->B % ngroups == 0, "%s: %d images do not split into %d groups", who,
               g->B, ngroups);
  tg->g = as_make_dev(g);
  tg->ngroups = ngroups;
  tg->imgs_per_group = g->B / ngroups;
  tg->tiles_per_row = (g->W + 31) / 32;
  const long tiles = (long)tg->imgs_per_group * g->H * tg->tiles_per_row;
  AS_CHECK_ARG(tiles < (1L << 24), "%s: map too large for the small-map kernels", who);
  tg->tiles_per_group = (int)tiles;
  const int rounds = as_div_up(tiles, TR_GPER_MAX);
  tg->gper = as_div_up(tiles, rounds);
  return AS_OK;
}

extern "C" int as_trunk_parts(const as_pcl* g, int ngroups) {
  TrunkGeom tg;
  if (trunk_geom(&tg, g, ngroups, "as_trunk_parts") != AS_OK) return -1;
  return tg.gper;
}

extern "C" int as_trunk_fwd(const float* src, const float* skip, const as_trunk_bn* bn_prev, float* a_out, const as_pcl* g,
                            int ngroups, const float* packed_w, const float* bias, float slope, float* z, float* stat_mean,
                            float* stat_m2, float* stat_cnt, void* stream) {
  TrunkFwdArgs a;
  if (int e = trunk_geom(&a.tg, g, ngroups, "as_trunk_fwd")) return e;
  AS_CHECK_ARG(src && packed_w && bias && z, "as_trunk_fwd: null argument");
  AS_CHECK_ARG((stat_mean == nullptr) == (stat_m2 == nullptr) && (stat_mean == nullptr) == (stat_cnt == nullptr),
               "as_trunk_fwd: stat_mean / stat_m2 / stat_cnt go together");
  AS_CHECK_ARG(z != src && z != skip && (a_out == nullptr || (a_out != src && a_out != skip && a_out != z)),
               "as_trunk_fwd: outputs must not alias inputs (tiles read their neighbours' voxels)");
  a.src = src; a.skip = skip; a.a_out = a_out; a.wp = packed_w; a.bias = bias; a.z = z;
  a.stat_mean = stat_mean; a.stat_m2 = stat_m2; a.stat_cnt = stat_cnt; a.slope = slope;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(a.tg.gper * ngroups);
#ifdef TR_TIMING_BUILD
  static long long* tbuf = nullptr;
  if (!tbuf) hipMalloc(&tbuf, 1024 * 8 * 8);
  a.timing = tbuf;
#endif
  if (bn_prev != nullptr) {
    AS_CHECK_ARG(skip && a_out && bn_prev->stat_mean && bn_prev->stat_m2 && bn_prev->stat_cnt && bn_prev->gamma && bn_prev->beta &&
                 bn_prev->state && bn_prev->nparts >= 1, "as_trunk_fwd: incomplete as_trunk_bn / skip / a_out");
    a.bn.stat_mean = bn_prev->stat_mean; a.bn.stat_m2 = bn_prev->stat_m2; a.bn.stat_cnt = bn_prev->stat_cnt;
    a.bn.gamma = bn_prev->gamma; a.bn.beta = bn_prev->beta; a.bn.state = bn_prev->state; a.bn.nparts = bn_prev->nparts;
    a.bn.eps = bn_prev->eps;
    hipLaunchKernelGGL(trunk_fwd_kernel<1>, grid, dim3(256), 0, st, a);
  } else {
    a.bn = TrunkBnIn{};
    hipLaunchKernelGGL(trunk_fwd_kernel<0>, grid, dim3(256), 0, st, a);
  }
  AS_CHECK_LAUNCH("as_trunk_fwd");
#ifdef TR_TIMING_BUILD
  if (getenv("AS_TR_TIMING")) {                            // dump THIS launch (synchronises: diagnostic build only)
    hipStreamSynchronize(st);
    static long long hbuf[1024 * 8];
    hipMemcpy(hbuf, tbuf, sizeof(hbuf), hipMemcpyDeviceToHost);
    FILE* f = fopen("gpurun_out/trunk_timing.bin", "wb");
    if (f) { int n = a.tg.gper * ngroups; fwrite(&n, 4, 1, f); fwrite(hbuf, 8, (size_t)n * 8, f); fclose(f); }
  }
#endif
  return AS_OK;
}

extern "C" int as_trunk_finish_fwd(const float* feats_pcl, const as_pcl* g, float* feats_nchw, const float* states, int nlayers,
                                   int ngroups, float* const* running_mean, float* const* running_var, float momentum,
                                   void* stream) {
  TrunkGeom tg;
  if (int e = trunk_geom(&tg, g, ngroups, "as_trunk_finish_fwd")) return e;
  AS_CHECK_ARG(feats_pcl && feats_nchw && nlayers >= 0 && nlayers <= TRUNK_MAX_LAYERS, "as_trunk_finish_fwd: bad argument");
  AS_CHECK_ARG(nlayers == 0 || (states && running_mean && running_var), "as_trunk_finish_fwd: states / running statistics missing");
  TrunkFinishArgs a;
  a.feats_pcl = feats_pcl; a.feats_nchw = feats_nchw; a.g = tg.g; a.states = states; a.nlayers = nlayers; a.ngroups = ngroups;
  a.momentum = momentum;
  for (int l = 0; l < TRUNK_MAX_LAYERS; ++l) {
    a.rm[l] = l < nlayers ? running_mean[l] : nullptr;
    a.rv[l] = l < nlayers ? running_var[l] : nullptr;
    AS_CHECK_ARG(l >= nlayers || (a.rm[l] && a.rv[l]), "as_trunk_finish_fwd: null running statistics of layer %d", l);
  }
  const int blocks = g->B * g->H * ((g->W + 31) / 32) + (nlayers > 0 ? 1 : 0);
  hipLaunchKernelGGL(trunk_finish_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  AS_CHECK_LAUNCH("as_trunk_finish_fwd");
  return AS_OK;
}

extern "C" int as_trunk_begin_bwd(const float* g_a, const float* g_b, int nA, const as_pcl* g, float* out_pcl, void* stream) {
  AS_CHECK_ARG(as_pcl_ok(g) && g->D == 1 && g->pd == 0, "as_trunk_begin_bwd: needs a 2-D PCL geometry");
  AS_CHECK_ARG(g_a && out_pcl && nA >= 1 && nA <= g->B && (nA == g->B || g_b != nullptr), "as_trunk_begin_bwd: bad argument");
  TrunkBeginArgs a;
  a.ga = g_a; a.gb = g_b; a.nA = nA; a.out = out_pcl; a.g = as_make_dev(g);
  hipLaunchKernelGGL(trunk_begin_bwd_kernel, dim3(g->B * g->H * ((g->W + 31) / 32)), dim3(256), 0, (hipStream_t)stream, a);
  AS_CHECK_LAUNCH("as_trunk_begin_bwd");
  return AS_OK;
}

extern "C" int64_t as_trunk_bwd_workspace(const as_pcl* g, int ngroups) {
  TrunkGeom tg;
  if (trunk_geom(&tg, g, ngroups, "as_trunk_bwd_workspace") != AS_OK) return -1;
  return (int64_t)tg.gper * ngroups * (9 * 1024 + 32);
}

extern "C" int as_trunk_bwd(const float* g_a, const float* z, const float* state, const double* sums, int nparts,
                            const float* gamma, float* bn_grads, const float* x, const float* packed_wt, float* g_x,
                            const float* z_next, const float* state_next, double* sums_next, const as_pcl* g, int ngroups,
                            float slope, float* dW, float* db, int accumulate, float* workspace, void* stream) {
  TrunkBwdArgs a;
  if (int e = trunk_geom(&a.tg, g, ngroups, "as_trunk_bwd")) return e;
  AS_CHECK_ARG(g_a && x && packed_wt && g_x && dW && db && workspace, "as_trunk_bwd: null argument");
  AS_CHECK_ARG(g_x != g_a && g_x != x && g_x != z, "as_trunk_bwd: g_x must not alias an input (tiles read their neighbours' voxels)");
  AS_CHECK_ARG((z_next == nullptr) == (state_next == nullptr) && (z_next == nullptr) == (sums_next == nullptr),
               "as_trunk_bwd: z_next / state_next / sums_next go together");
  const int slabs = a.tg.gper * ngroups;
  a.g_a = g_a; a.z = z; a.state = state; a.sums = sums; a.nparts = nparts; a.gamma = gamma; a.bn_grads = bn_grads; a.x = x;
  a.wp_t = packed_wt; a.g_x = g_x; a.z_next = z_next; a.state_next = state_next; a.sums_next = sums_next;
  a.slab = workspace; a.slab_db = workspace + (long)slabs * 9 * 1024; a.slope = slope;
  a.count = (float)((long)a.tg.imgs_per_group * g->H * g->W);
  hipStream_t st = (hipStream_t)stream;
  if (z != nullptr) {
    AS_CHECK_ARG(state && sums && nparts >= 1 && gamma && bn_grads, "as_trunk_bwd: incomplete BatchNorm arguments");
    hipLaunchKernelGGL(trunk_bwd_kernel<1>, dim3(slabs), dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL(trunk_bwd_kernel<0>, dim3(slabs), dim3(256), 0, st, a);
  }
  AS_CHECK_LAUNCH("as_trunk_bwd");
  as_wgrad_reduce_enqueue(st, a.slab, a.slab_db, slabs, 9, dW, db, accumulate);
  AS_CHECK_LAUNCH("as_trunk_bwd (reduce)");
  return AS_OK;
}

extern "C" int as_trunk_finish_bwd(const float* bn_grads, int nlayers, int ngroups, float* const* g_gamma, float* const* g_beta,
                                   int accumulate, void* stream) {
  AS_CHECK_ARG(bn_grads && g_gamma && g_beta && nlayers >= 1 && nlayers <= TRUNK_MAX_LAYERS && ngroups >= 1 &&
               ngroups <= TRUNK_MAX_GROUPS, "as_trunk_finish_bwd: bad argument");
  TrunkBnGradArgs a;
  a.bn_grads = bn_grads; a.nlayers = nlayers; a.ngroups = ngroups; a.accumulate = accumulate;
  for (int l = 0; l < TRUNK_MAX_LAYERS; ++l) {
    a.g_gamma[l] = l < nlayers ? g_gamma[l] : nullptr;
    a.g_beta[l] = l < nlayers ? g_beta[l] : nullptr;
    AS_CHECK_ARG(l >= nlayers || (a.g_gamma[l] && a.g_beta[l]), "as_trunk_finish_bwd: null destination of layer %d", l);
  }
  hipLaunchKernelGGL(trunk_finish_bwd_kernel, dim3(as_div_up(nlayers * 64, 256)), dim3(256), 0, (hipStream_t)stream, a);
  AS_CHECK_LAUNCH("as_trunk_finish_bwd");
  return AS_OK;
}
