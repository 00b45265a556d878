// BatchNorm (train / eval) + LeakyReLU(0.2) (+ BasicBlock skip) around conv32.
// Reference semantics: nn.BatchNorm3d/2d defaults (eps 1e-5, momentum 0.1, affine,
// running stats) + nn.LeakyReLU(0.2, inplace) — adaptive_stereo/models/stereo_net.py
// :17,29,39,94,159; train-mode batch statistics during adaptation (adapt.py:309-314).
//
// Statistics are exact-merge (Chan) in fp64 of the per-workgroup (mean, M2) pairs the
// convolution epilogue emitted from its register tile, mirroring ATen's CPU kernel,
// which accumulates BatchNorm moments in double for float inputs.
// All element-wise passes are HBM-bound: float4 per lane over the PCL interior.
#include "as_common.h"

// ---- finalize: one workgroup, 1024 threads = 32 slices x 32 channels ---------------------
// Exact merge of the per-partial (n, mean, M2) triples in fp64:  mean = sum n_i*mean_i / N;  M2 = sum [ M2_i + n_i*(mean_i - mean)^2 ]
// in its ONE-PASS form (bn_merge.h: with a pivot K = the first partial's mean, S0 = sum n_i, S1 = sum n_i (mean_i - K),
// S2 = sum [ M2_i + n_i (mean_i - K)^2 ] give mean = K + S1/S0 and M2 = S2 - S1^2/S0 — the two-pass result to ~1e-15 relative
// in fp64).  This launch is a single workgroup on the critical path of a layer: the two-pass form it had until round 5 was
// two dependent rounds of loads (7.6 us per launch; 8 launches per step).  Reads are 128-byte rows in a fixed order.
__global__ __launch_bounds__(1024) void bn_finalize_kernel(
    const float* __restrict__ stat_mean, const float* __restrict__ stat_m2, const float* __restrict__ stat_cnt,
    int nparts, const float* __restrict__ gamma, const float* __restrict__ beta, float* running_mean,
    float* running_var, float momentum, float eps, float* save_mean, float* save_invstd, float* scale, float* shift) {
  __shared__ double red[3][32][33];
  const int c = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const double K = (double)stat_mean[c];
  // branch-free and unrolled: the loads of 8 iterations are independent and in flight together
  // (empty partials carry count 0, mean 0, M2 0)
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll 8
  for (int i = sl; i < nparts; i += 32) {
    const double nb = (double)stat_cnt[i];
    const double dm = (double)stat_mean[i * 32 + c] - K;
    s0 += nb; s1 += nb * dm; s2 += (double)stat_m2[i * 32 + c] + nb * dm * dm;
  }
  red[0][sl][c] = s0; red[1][sl][c] = s1; red[2][sl][c] = s2;
  __syncthreads();
  if (sl == 0) {
    double t0 = 0.0, t1 = 0.0, t2 = 0.0;
    for (int j = 0; j < 32; ++j) { t0 += red[0][j][c]; t1 += red[1][j][c]; t2 += red[2][j][c]; }
    const double n = t0;
    const double mean = K + t1 / n;
    const double m2 = fmax(t2 - t1 * t1 / n, 0.0);
    const double var_b = m2 / n;
    const float invstd = (float)(1.0 / sqrt(var_b + (double)eps));
    const float meanf = (float)mean;
    save_mean[c] = meanf;
    save_invstd[c] = invstd;
    const float sc = invstd * gamma[c];
    scale[c] = sc;
    shift[c] = beta[c] - meanf * sc;
    if (running_mean) {
      const double var_u = n > 1.0 ? m2 / (n - 1.0) : var_b;
      running_mean[c] = (float)((double)momentum * mean + (1.0 - (double)momentum) * (double)running_mean[c]);
      running_var[c] = (float)((double)momentum * var_u + (1.0 - (double)momentum) * (double)running_var[c]);
    }
  }
}

__global__ void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, float* save_mean, float* save_invstd, float* scale, float* shift) {
  const int c = threadIdx.x;
  if (c >= 32) return;
  const float invstd = 1.0f / sqrtf(rv[c] + eps);
  const float sc = invstd * gamma[c];
  save_mean[c] = rm[c];
  save_invstd[c] = invstd;
  scale[c] = sc;
  shift[c] = beta[c] - rm[c] * sc;
}

// ---- element-wise passes: row chunks ---------------------------------------------------
// The PCL interior is B*D*H contiguous rows of W voxels (128 B each).  A workgroup takes "row chunks" of 128
// voxels = 1024 float4: thread t owns float4 t, t+256, t+512, t+768 of the chunk (always channel group t&7),
// so one scalar decode per chunk replaces a 64-bit div/mod per element and every thread has four
// independent 16-byte loads per tensor in flight.
#define BN_CHUNK_VOX 128
struct RowChunk { long base; int nf4; };     // float offset of the chunk's first voxel; valid float4 count

__device__ inline RowChunk row_chunk(const PclDev& g, int chunk, int chunks_per_row) {
  const int rowi = chunk / chunks_per_row, cx = chunk - rowi * chunks_per_row;
  const int y = rowi % g.H, t = rowi / g.H;
  const int d = t % g.D, b = t / g.D;
  const int x0 = cx * BN_CHUNK_VOX;
  RowChunk rc;
  rc.base = g.vox(b, d, y, x0) * 32;
  rc.nf4 = min(BN_CHUNK_VOX, g.W - x0) * 8;
  return rc;
}

// Traversal that follows a conv32 LDS producer back to front.  That kernel walks its row segments (the same
// 128-voxel chunks, same numbering) in 8 contiguous bands, one per XCD, all bands advancing together; what
// it wrote last — the tail of every band — is what L2 and the memory-side cache still hold when the consumer
// starts.  Position p of the consumer's schedule -> chunk (band p&7, p>>3 from that band's end).
__device__ inline int band_tail_order(int p, int nchunks) {
  const int tpb = (nchunks + 7) >> 3;
  const int ch = (p & 7) * tpb + (tpb - 1 - (p >> 3));
  return ch;        // may be >= nchunks in the last band (caller skips)
}

__device__ inline f32x4 lrelu4(f32x4 y, f32x4 v, float slope) {   // v where y > 0, v*slope elsewhere
  f32x4 r;
  r.x = y.x > 0.f ? v.x : v.x * slope; r.y = y.y > 0.f ? v.y : v.y * slope;
  r.z = y.z > 0.f ? v.z : v.z * slope; r.w = y.w > 0.f ? v.w : v.w * slope;
  return r;
}

// every eval-mode layer of a forward pass in one launch (the job table lives on the device, built once)
__global__ void bn_eval_affine_batch_kernel(const as_bn_affine_job* __restrict__ jobs, float eps) {
  const as_bn_affine_job j = jobs[blockIdx.x];
  const int c = threadIdx.x;
  if (c >= 32) return;
  const float invstd = 1.0f / sqrtf(j.running_var[c] + eps);
  const float sc = invstd * j.gamma[c];
  j.out[c] = j.running_mean[c];
  j.out[32 + c] = invstd;
  j.out[64 + c] = sc;
  j.out[96 + c] = j.beta[c] - j.running_mean[c] * sc;
}

// ---- a = lrelu(z*scale + shift) (+ residual), interior only -------------------------
template <bool RES>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, float slope,
                                                          const float* __restrict__ residual, float* __restrict__ a,
                                                          PclDev g, int nchunks, int chunks_per_row) {
  const int c4 = threadIdx.x & 7;
  const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c4 * 4);
  const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c4 * 4);
  const int npos = ((nchunks + 7) >> 3) << 3;
  for (int p = blockIdx.x; p < npos; p += gridDim.x) {
    const int ch = band_tail_order(p, nchunks);
    if (ch >= nchunks) continue;
    const RowChunk rc = row_chunk(g, ch, chunks_per_row);
    f32x4 q[4], r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int f = threadIdx.x + 256 * k;
      if (f < rc.nf4) {
        q[k] = *reinterpret_cast<const f32x4*>(z + rc.base + f * 4);
        if (RES) r[k] = *reinterpret_cast<const f32x4*>(residual + rc.base + f * 4);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int f = threadIdx.x + 256 * k;
      if (f < rc.nf4) {
        f32x4 y = q[k] * sc + sh;
        y = lrelu4(y, y, slope);
        if (RES) y += r[k];
        *reinterpret_cast<f32x4*>(a + rc.base + f * 4) = y;
      }
    }
  }
}

// ---- backward ------------------------------------------------------------------------
// stage 1: per-workgroup partial sums of g_y and g_y*(z-mean) per channel (fp64 slabs).
#define BNB_BLOCKS 1024
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ g_a, const float* __restrict__ z,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             const float* __restrict__ mean, float slope,
                                                             double* __restrict__ partial, PclDev g, int nchunks,
                                                             int chunks_per_row) {
  __shared__ double red[2][32][33];
  const int c4 = threadIdx.x & 7, vl = threadIdx.x >> 3;
  const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c4 * 4);
  const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c4 * 4);
  const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c4 * 4);
  f32x4 s_dy = {0.f, 0.f, 0.f, 0.f}, s_dx = {0.f, 0.f, 0.f, 0.f};
  const int npos = ((nchunks + 7) >> 3) << 3;
  for (int p = blockIdx.x; p < npos; p += gridDim.x) {
    const int ch = band_tail_order(p, nchunks);      // g_a comes straight from the data-gradient convolution
    if (ch >= nchunks) continue;
    const RowChunk rc = row_chunk(g, ch, chunks_per_row);
    f32x4 zz[4], gy[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int f = threadIdx.x + 256 * k;
      if (f < rc.nf4) {
        zz[k] = *reinterpret_cast<const f32x4*>(z + rc.base + f * 4);
        gy[k] = *reinterpret_cast<const f32x4*>(g_a + rc.base + f * 4);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int f = threadIdx.x + 256 * k;
      if (f < rc.nf4) {
        const f32x4 gg = lrelu4(zz[k] * sc + sh, gy[k], slope);
        s_dy += gg;
        s_dx += gg * (zz[k] - mu);
      }
    }
  }
  for (int i = 0; i < 4; ++i) {
    red[0][vl][c4 * 4 + i] = (double)s_dy[i];
    red[1][vl][c4 * 4 + i] = (double)s_dx[i];
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int which = threadIdx.x >> 5, c = threadIdx.x & 31;
    double s = 0.0;
    for (int j = 0; j < 32; ++j) s += red[which][j][c];
    partial[(long)blockIdx.x * 64 + which * 32 + c] = s;
  }
}

// stage 2: fixed-order sum of the slabs (1024 threads = 16 slices x 64 sums); emits g_gamma, g_beta and
// the per-channel coefficients of stage 3:  g_z = (g_y - k1 - (z-mean)*k2) * k3.
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const double* __restrict__ partial, int nblocks, long count,
                                                               const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                               int train, float* g_gamma, float* g_beta, float* coef,
                                                               int accumulate) {
  __shared__ double red[16][64];
  const int j = threadIdx.x & 63, sl = threadIdx.x >> 6;      // 16 slices x 64 sums: a slice adds <= 64 slabs
  // every load of a round of 32 slabs in flight together, then added in slab order (the order of the plain loop: the same
  // bits): this kernel is one workgroup on the critical path of a layer's backward, its time is its round trips — with 8
  // loads per round a 512-slab merge was four dependent rounds (7 us), now one
  double s = 0.0;
  for (int i0 = sl; i0 < nblocks; i0 += 16 * 32) {
    double v[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) { const int i = i0 + 16 * u; v[u] = i < nblocks ? partial[(long)i * 64 + j] : 0.0; }
#pragma unroll
    for (int u = 0; u < 32; ++u) s += v[u];
  }
  red[sl][j] = s;
  __syncthreads();
  if (threadIdx.x < 32) {
    const int c = threadIdx.x;
    double sdy = 0.0, sdx = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) { sdy += red[q][c]; sdx += red[q][32 + c]; }
    const double is = (double)invstd[c];
    g_gamma[c] = accumulate ? g_gamma[c] + (float)(sdx * is) : (float)(sdx * is);
    g_beta[c] = accumulate ? g_beta[c] + (float)sdy : (float)sdy;
    coef[c] = train ? (float)(sdy / (double)count) : 0.f;
    coef[32 + c] = train ? (float)(sdx * is * is / (double)count) : 0.f;
    coef[64 + c] = invstd[c] * gamma[c];
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ g_a, const float* __restrict__ z,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ mean, const float* __restrict__ coef,
                                                            float slope, float* __restrict__ g_z, PclDev g, int nchunks,
                                                            int chunks_per_row) {
  const int c4 = threadIdx.x & 7;
  const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c4 * 4);
  const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c4 * 4);
  const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c4 * 4);
  const f32x4 k1 = *reinterpret_cast<const f32x4*>(coef + c4 * 4);
  const f32x4 k2 = *reinterpret_cast<const f32x4*>(coef + 32 + c4 * 4);
  const f32x4 k3 = *reinterpret_cast<const f32x4*>(coef + 64 + c4 * 4);
  const int npos = ((nchunks + 7) >> 3) << 3;
  for (int p = blockIdx.x; p < npos; p += gridDim.x) {
    // stage 1's schedule, last position first: what it streamed last is still cached
    const int ch = band_tail_order(npos - 1 - p, nchunks);
    if (ch >= nchunks) continue;
    const RowChunk rc = row_chunk(g, ch, chunks_per_row);
    f32x4 zz[4], gy[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int f = threadIdx.x + 256 * k;
      if (f < rc.nf4) {
        zz[k] = *reinterpret_cast<const f32x4*>(z + rc.base + f * 4);
        gy[k] = *reinterpret_cast<const f32x4*>(g_a + rc.base + f * 4);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int f = threadIdx.x + 256 * k;
      if (f < rc.nf4) {
        const f32x4 gg = lrelu4(zz[k] * sc + sh, gy[k], slope);
        const f32x4 dx = (zz[k] - mu) * k2;
        *reinterpret_cast<f32x4*>(g_z + rc.base + f * 4) = (gg - k1 - dx) * k3;
      }
    }
  }
}

// ---- host ------------------------------------------------------------------------------
static inline int chunks_per_row(const as_pcl* g) { return (g->W + BN_CHUNK_VOX - 1) / BN_CHUNK_VOX; }
static inline long row_chunks(const as_pcl* g) { return (long)g->B * g->D * g->H * chunks_per_row(g); }
static inline int elementwise_blocks(long nchunks) {
  return (int)(nchunks > 16384 ? 16384 : nchunks);   // grid-stride beyond 64 workgroups per CU
}

extern "C" int as_bn_finalize(const float* stat_mean, const float* stat_m2, const float* stat_cnt, int nparts,
                              const float* gamma, const float* beta, float* running_mean, float* running_var,
                              float momentum, float eps, float* save_mean, float* save_invstd,
                              float* scale, float* shift, void* stream) {
  AS_CHECK_ARG(stat_mean && stat_m2 && stat_cnt && gamma && beta && save_mean && save_invstd && scale && shift,
               "as_bn_finalize: null pointer");
  AS_CHECK_ARG(nparts >= 1, "as_bn_finalize: nparts=%d", nparts);
  AS_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "as_bn_finalize: running stats must pair");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, stat_mean, stat_m2, stat_cnt,
                     nparts, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_invstd,
                     scale, shift);
  AS_CHECK_LAUNCH("as_bn_finalize");
  return AS_OK;
}

extern "C" int as_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                                 const float* running_var, float eps, float* save_mean, float* save_invstd,
                                 float* scale, float* shift, void* stream) {
  AS_CHECK_ARG(gamma && beta && running_mean && running_var && save_mean && save_invstd && scale && shift,
               "as_bn_eval_affine: null pointer");
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, gamma, beta, running_mean,
                     running_var, eps, save_mean, save_invstd, scale, shift);
  AS_CHECK_LAUNCH("as_bn_eval_affine");
  return AS_OK;
}

extern "C" int as_bn_eval_affine_batch(const as_bn_affine_job* jobs, int njobs, float eps, void* stream) {
  AS_CHECK_ARG(jobs && njobs >= 1 && njobs <= 65535, "as_bn_eval_affine_batch: bad job table");
  hipLaunchKernelGGL(bn_eval_affine_batch_kernel, dim3(njobs), dim3(64), 0, (hipStream_t)stream, jobs, eps);
  AS_CHECK_LAUNCH("as_bn_eval_affine_batch");
  return AS_OK;
}

extern "C" int as_bn_act_fwd(const float* z, const float* scale, const float* shift, float slope,
                             const float* residual, float* a, const as_pcl* g, void* stream) {
  AS_CHECK_ARG(as_pcl_ok(g), "as_bn_act_fwd: bad geometry");
  AS_CHECK_ARG(z && scale && shift && a, "as_bn_act_fwd: null pointer");
  const long nch = row_chunks(g);
  AS_CHECK_ARG(nch < (1L << 31), "as_bn_act_fwd: volume too large");
  as_prof_mark(4, (hipStream_t)stream, 1, 0.0);
  if (residual)
    hipLaunchKernelGGL(bn_act_fwd_kernel<true>, dim3(elementwise_blocks(nch)), dim3(256), 0, (hipStream_t)stream, z,
                       scale, shift, slope, residual, a, as_make_dev(g), (int)nch, chunks_per_row(g));
  else
    hipLaunchKernelGGL(bn_act_fwd_kernel<false>, dim3(elementwise_blocks(nch)), dim3(256), 0, (hipStream_t)stream, z,
                       scale, shift, slope, residual, a, as_make_dev(g), (int)nch, chunks_per_row(g));
  as_prof_mark(4, (hipStream_t)stream, 0, (residual ? 3.0 : 2.0) * 128.0 * (double)g->B * g->D * g->H * g->W);
  AS_CHECK_LAUNCH("as_bn_act_fwd");
  return AS_OK;
}

static int bnb_blocks(long nchunks) {
  long nb = (nchunks + 3) / 4;      // at least four row chunks (512 voxels) per workgroup
  if (nb > BNB_BLOCKS) nb = BNB_BLOCKS;
  if (nb < 1) nb = 1;
  return (int)nb;
}

extern "C" int64_t as_bn_bwd_coef_offset(void) { return (int64_t)BNB_BLOCKS * 128; }

extern "C" int64_t as_bn_bwd_workspace(const as_pcl* g) {
  if (!as_pcl_ok(g)) return -1;
  // fp64 slabs [blocks][64] (= 128 floats each) + 96 coefficient floats
  return (int64_t)BNB_BLOCKS * 128 + 96 + 32;
}

static int bn_act_bwd_impl(const float* g_a, const float* z, const float* scale, const float* shift,
                           const float* save_mean, const float* save_invstd, const float* gamma,
                           float slope, int train, float* g_z, float* g_gamma, float* g_beta, int accumulate,
                           float* workspace, const as_pcl* g, int nparts_given, void* stream) {
  AS_CHECK_ARG(as_pcl_ok(g), "as_bn_act_bwd: bad geometry");
  AS_CHECK_ARG(g_a && z && scale && shift && save_mean && save_invstd && gamma && g_gamma && g_beta && workspace,
               "as_bn_act_bwd: null pointer");
  AS_CHECK_ARG(((uintptr_t)workspace & 7) == 0, "as_bn_act_bwd: workspace must be 8-byte aligned");
  const long M = (long)g->B * g->D * g->H * g->W;
  const long nch = row_chunks(g);
  AS_CHECK_ARG(nch < (1L << 31), "as_bn_act_bwd: volume too large");
  AS_CHECK_ARG(nparts_given >= 0 && nparts_given <= BNB_BLOCKS, "as_bn_act_bwd_given: %d partials", nparts_given);
  const int nb = nparts_given > 0 ? nparts_given : bnb_blocks(nch), cpr = chunks_per_row(g);
  double* partial = reinterpret_cast<double*>(workspace);
  float* coef = workspace + (int64_t)BNB_BLOCKS * 128;
  hipStream_t st = (hipStream_t)stream;
  const PclDev gd = as_make_dev(g);
  as_prof_mark(5, st, 1, 0.0);
  if (nparts_given == 0) {
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nb), dim3(256), 0, st, g_a, z, scale, shift, save_mean, slope,
                       partial, gd, (int)nch, cpr);
    AS_CHECK_LAUNCH("as_bn_act_bwd(reduce)");
  }
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(1), dim3(1024), 0, st, partial, nb, M, save_invstd, gamma, train,
                     g_gamma, g_beta, coef, accumulate);
  AS_CHECK_LAUNCH("as_bn_act_bwd(finalize)");
  if (g_z == nullptr) {            // stage 3 is somebody else's (as_conv32_wgrad_bnapply): coefficients stay in the workspace
    as_prof_mark(5, st, 0, (nparts_given == 0 ? 2.0 : 0.0) * 128.0 * (double)M);
    return AS_OK;
  }
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(elementwise_blocks(nch)), dim3(256), 0, st, g_a, z, scale, shift,
                     save_mean, coef, slope, g_z, gd, (int)nch, cpr);
  as_prof_mark(5, st, 0, (nparts_given == 0 ? 5.0 : 3.0) * 128.0 * (double)M);
  AS_CHECK_LAUNCH("as_bn_act_bwd(apply)");
  return AS_OK;
}

// ---- cross-replica ("synchronised") BatchNorm backward: stages 1+2a | all-reduce by the caller | 2b | 3 ----------
// stage 2a: the fixed-order slab sum of bn_bwd_finalize_kernel alone; sums[64] = element count
__global__ __launch_bounds__(1024) void bn_bwd_sums_kernel(const double* __restrict__ partial, int nblocks, long count,
                                                           double* __restrict__ sums) {
  __shared__ double red[16][64];
  const int j = threadIdx.x & 63, sl = threadIdx.x >> 6;
  double s = 0.0;
#pragma unroll 8
  for (int i = sl; i < nblocks; i += 16) s += partial[(long)i * 64 + j];
  red[sl][j] = s;
  __syncthreads();
  if (threadIdx.x < 64) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][threadIdx.x];
    sums[threadIdx.x] = t;
  }
  if (threadIdx.x == 64) sums[64] = (double)count;
}

// stage 2b: parameter gradients from THIS replica's sums (the gradient all-reduce adds the replicas'), stage-3
// coefficients from the sums and the element count of ALL replicas
__global__ void bn_bwd_finalize_synced_kernel(const double* __restrict__ local, const double* __restrict__ global,
                                              const float* __restrict__ invstd, const float* __restrict__ gamma,
                                              float* g_gamma, float* g_beta, float* coef, int accumulate) {
  const int c = threadIdx.x;
  if (c >= 32) return;
  const double is = (double)invstd[c];
  const double ldy = local[c], ldx = local[32 + c];
  g_gamma[c] = accumulate ? g_gamma[c] + (float)(ldx * is) : (float)(ldx * is);
  g_beta[c] = accumulate ? g_beta[c] + (float)ldy : (float)ldy;
  const double count = global[64];
  coef[c] = (float)(global[c] / count);
  coef[32 + c] = (float)(global[32 + c] * is * is / count);
  coef[64 + c] = invstd[c] * gamma[c];
}

extern "C" int as_bn_bwd_sums(const float* g_a, const float* z, const float* scale, const float* shift,
                              const float* save_mean, float slope, float* workspace, const as_pcl* g,
                              int nparts_given, double* sums, void* stream) {
  AS_CHECK_ARG(as_pcl_ok(g), "as_bn_bwd_sums: bad geometry");
  AS_CHECK_ARG(workspace && sums, "as_bn_bwd_sums: null pointer");
  AS_CHECK_ARG(((uintptr_t)workspace & 7) == 0 && ((uintptr_t)sums & 7) == 0, "as_bn_bwd_sums: 8-byte alignment required");
  AS_CHECK_ARG(nparts_given >= 0 && nparts_given <= BNB_BLOCKS, "as_bn_bwd_sums: %d partials", nparts_given);
  const long M = (long)g->B * g->D * g->H * g->W;
  const long nch = row_chunks(g);
  AS_CHECK_ARG(nch < (1L << 31), "as_bn_bwd_sums: volume too large");
  const int nb = nparts_given > 0 ? nparts_given : bnb_blocks(nch);
  double* partial = reinterpret_cast<double*>(workspace);
  hipStream_t st = (hipStream_t)stream;
  if (nparts_given == 0) {
    AS_CHECK_ARG(g_a && z && scale && shift && save_mean, "as_bn_bwd_sums: null pointer");
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nb), dim3(256), 0, st, g_a, z, scale, shift, save_mean, slope,
                       partial, as_make_dev(g), (int)nch, chunks_per_row(g));
    AS_CHECK_LAUNCH("as_bn_bwd_sums(reduce)");
  }
  hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3(1), dim3(1024), 0, st, partial, nb, M, sums);
  AS_CHECK_LAUNCH("as_bn_bwd_sums");
  return AS_OK;
}

extern "C" int as_bn_bwd_finalize_synced(const double* local_sums, const double* global_sums, const float* save_invstd,
                                         const float* gamma, float* g_gamma, float* g_beta, int accumulate,
                                         float* workspace, void* stream) {
  AS_CHECK_ARG(local_sums && global_sums && save_invstd && gamma && g_gamma && g_beta && workspace,
               "as_bn_bwd_finalize_synced: null pointer");
  hipLaunchKernelGGL(bn_bwd_finalize_synced_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, local_sums, global_sums,
                     save_invstd, gamma, g_gamma, g_beta, workspace + (int64_t)BNB_BLOCKS * 128, accumulate);
  AS_CHECK_LAUNCH("as_bn_bwd_finalize_synced");
  return AS_OK;
}

extern "C" int as_bn_bwd_apply(const float* g_a, const float* z, const float* scale, const float* shift,
                               const float* save_mean, float slope, float* g_z, const float* workspace,
                               const as_pcl* g, void* stream) {
  AS_CHECK_ARG(as_pcl_ok(g), "as_bn_bwd_apply: bad geometry");
  AS_CHECK_ARG(g_a && z && scale && shift && save_mean && g_z && workspace, "as_bn_bwd_apply: null pointer");
  const long nch = row_chunks(g);
  AS_CHECK_ARG(nch < (1L << 31), "as_bn_bwd_apply: volume too large");
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(elementwise_blocks(nch)), dim3(256), 0, (hipStream_t)stream, g_a, z, scale,
                     shift, save_mean, workspace + (int64_t)BNB_BLOCKS * 128, slope, g_z, as_make_dev(g), (int)nch,
                     chunks_per_row(g));
  AS_CHECK_LAUNCH("as_bn_bwd_apply");
  return AS_OK;
}

extern "C" int as_bn_act_bwd(const float* g_a, const float* z, const float* scale, const float* shift,
                             const float* save_mean, const float* save_invstd, const float* gamma,
                             float slope, int train, float* g_z, float* g_gamma, float* g_beta, int accumulate,
                             float* workspace, const as_pcl* g, void* stream) {
  return bn_act_bwd_impl(g_a, z, scale, shift, save_mean, save_invstd, gamma, slope, train, g_z, g_gamma, g_beta,
                         accumulate, workspace, g, 0, stream);
}

// Stage 1 (the per-channel sums) was already done by as_conv32_fwd_bnbwd: `workspace` holds `nparts` slabs.
extern "C" int as_bn_act_bwd_given(const float* g_a, const float* z, const float* scale, const float* shift,
                                   const float* save_mean, const float* save_invstd, const float* gamma,
                                   float slope, int train, float* g_z, float* g_gamma, float* g_beta, int accumulate,
                                   float* workspace, const as_pcl* g, int nparts, void* stream) {
  AS_CHECK_ARG(nparts >= 1, "as_bn_act_bwd_given: nparts=%d", nparts);
  return bn_act_bwd_impl(g_a, z, scale, shift, save_mean, save_invstd, gamma, slope, train, g_z, g_gamma, g_beta,
                         accumulate, workspace, g, nparts, stream);
}
