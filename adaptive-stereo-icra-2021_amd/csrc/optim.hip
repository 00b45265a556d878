// a12: global-norm clip + Adam over one flat fp32 arena, and library bookkeeping.
// Reference semantics: nn.utils.clip_grad_norm_(stereo_net.parameters(), 1.0) followed by
// torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8).step() — adapt.py:208-210, 391-393.
// The reference issues one small kernel per parameter tensor per Adam sub-step (~100
// tensors x ~6 ops).  Here parameters, gradients and both moments live at equal offsets
// of flat arenas (the same flat gradient arena is the RCCL all-reduce bucket), so the whole
// update is one sum-of-squares reduction and one element-wise pass.  HBM-bound, 1.7 MB.
#include "as_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void as_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* as_last_error(void) { return g_err; }
extern "C" int as_version(void) { return 1; }

extern "C" int64_t as_pcl_numel(const as_pcl* g) {
  if (!g) return -1;
  return (int64_t)g->B * (g->D + 2 * g->pd) * (g->H + 2 * g->ph) * (g->W + 2 * g->pw) * 32;
}

// ---- measurement hook -------------------------------------------------------------------
#define PROF_IDS AS_PROF_IDS
#define PROF_MAX 8192
static int g_prof_on = 0;
static hipEvent_t g_prof_ev[PROF_IDS][PROF_MAX][2];
static int g_prof_created[PROF_IDS] = {0};
static int g_prof_n[PROF_IDS] = {0};
static double g_prof_flops[PROF_IDS] = {0.0};

void as_prof_mark(int id, hipStream_t st, int begin, double flops) {
  if (!g_prof_on || id < 0 || id >= PROF_IDS) return;
  int n = g_prof_n[id];
  if (n >= PROF_MAX) return;
  if (n >= g_prof_created[id]) {
    if (hipEventCreate(&g_prof_ev[id][n][0]) != hipSuccess) return;
    if (hipEventCreate(&g_prof_ev[id][n][1]) != hipSuccess) return;
    g_prof_created[id] = n + 1;
  }
  if (begin) {
    (void)hipEventRecord(g_prof_ev[id][n][0], st);
  } else {
    (void)hipEventRecord(g_prof_ev[id][n][1], st);
    g_prof_flops[id] += flops;
    g_prof_n[id] = n + 1;
  }
}

extern "C" int as_prof_enable(int on) { g_prof_on = on ? 1 : 0; return AS_OK; }
extern "C" int as_prof_reset(void) {
  for (int i = 0; i < PROF_IDS; ++i) { g_prof_n[i] = 0; g_prof_flops[i] = 0.0; }
  return AS_OK;
}
extern "C" int as_prof_read(int id, int64_t* launches, double* total_ms, double* total_flops) {
  AS_CHECK_ARG(id >= 0 && id < PROF_IDS && launches && total_ms && total_flops, "as_prof_read: bad argument");
  double ms = 0.0;
  for (int i = 0; i < g_prof_n[id]; ++i) {
    float t = 0.f;
    if (hipEventSynchronize(g_prof_ev[id][i][1]) != hipSuccess ||
        hipEventElapsedTime(&t, g_prof_ev[id][i][0], g_prof_ev[id][i][1]) != hipSuccess) {
      as_set_error("as_prof_read: event query failed");
      return AS_ERR_LAUNCH;
    }
    ms += (double)t;
  }
  *launches = g_prof_n[id]; *total_ms = ms; *total_flops = g_prof_flops[id];
  return AS_OK;
}

#define SS_BLOCKS 256
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, double* __restrict__ partial) {
  __shared__ double red[4];
  double s = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const double v = (double)g[i];
    s += v * v;
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// (optionally also the clip coefficient of torch.nn.utils.clip_grad_norm_ and the optimizer's device-side step counter: three
// one-thread jobs in the launch that exists anyway)
__global__ void sumsq_finalize_kernel(const double* __restrict__ partial, int nblk, float* __restrict__ out, float max_norm,
                                      float* __restrict__ coef, float* __restrict__ step_counter) {
  double s = 0.0;                                // one wave; lane-strided, then a shuffle tree: fixed order
  for (int i = threadIdx.x; i < nblk; i += 64) s += partial[i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) {
    out[0] = (float)s;
    if (coef) coef[0] = fminf(max_norm / (sqrtf((float)s) + 1e-6f), 1.0f);      // as clip_coef_kernel, from the stored fp32 value
    if (step_counter) step_counter[0] += 1.0f;
  }
}

// torch.nn.utils.clip_grad_norm_: coef = min(max_norm / (sqrt(sumsq) + 1e-6), 1), in fp32 like ATen
__global__ void clip_coef_kernel(const float* __restrict__ sumsq, float max_norm, float* __restrict__ coef) {
  if (threadIdx.x == 0) coef[0] = fminf(max_norm / (sqrtf(sumsq[0]) + 1e-6f), 1.0f);
}

extern "C" int as_clip_coef(const float* sumsq, float max_norm, float* coef, void* stream) {
  AS_CHECK_ARG(sumsq && coef && max_norm > 0.f, "as_clip_coef: bad argument");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sumsq, max_norm, coef);
  AS_CHECK_LAUNCH("as_clip_coef");
  return AS_OK;
}

// ---- small glue of EdgeAwareRefinement's backward (stereo_net.py:116-121), one launch each instead of torch expressions ----
// g_in = g_out where out > 0, else 0: the final ReLU's backward.
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ g_out, const float* __restrict__ out, long n,
                                                        float* __restrict__ g_in) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    const f32x4 g = *reinterpret_cast<const f32x4*>(g_out + i), o = *reinterpret_cast<const f32x4*>(out + i);
    f32x4 r;
    r.x = o.x > 0.f ? g.x : 0.f; r.y = o.y > 0.f ? g.y : 0.f; r.z = o.z > 0.f ? g.z : 0.f; r.w = o.w > 0.f ? g.w : 0.f;
    *reinterpret_cast<f32x4*>(g_in + i) = r;
  } else {
    for (long k = i; k < n; ++k) g_in[k] = out[k] > 0.f ? g_out[k] : 0.f;
  }
}

extern "C" int as_relu_bwd(const float* g_out, const float* out, int64_t n, float* g_in, void* stream) {
  AS_CHECK_ARG(g_out && out && g_in && n > 0, "as_relu_bwd: bad argument");
  AS_CHECK_ARG((((uintptr_t)g_out | (uintptr_t)out | (uintptr_t)g_in) & 15) == 0, "as_relu_bwd: 16-byte aligned tensors");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(as_div_up(n, 1024)), dim3(256), 0, (hipStream_t)stream, g_out, out, (long)n, g_in);
  AS_CHECK_LAUNCH("as_relu_bwd");
  return AS_OK;
}

// Input channel 0 of a [32][Cin][3][3] weight with mirrored taps, in the two layouts the 32->1 data-gradient kernels read:
// by_tap[t][c] = by_channel[c][t] = w[c][0][8 - t].
__global__ void mirror_ch0_kernel(const float* __restrict__ w, int Cin, float* __restrict__ by_tap, float* __restrict__ by_channel) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 288) return;
  const int c = i / 9, t = i - 9 * c;
  const float v = w[((long)c * Cin) * 9 + (8 - t)];
  if (by_tap) by_tap[t * 32 + c] = v;
  if (by_channel) by_channel[c * 9 + t] = v;
}

extern "C" int as_mirror_taps_ch0(const float* w, int Cin, float* by_tap, float* by_channel, void* stream) {
  AS_CHECK_ARG(w && Cin >= 1 && (by_tap || by_channel), "as_mirror_taps_ch0: bad argument");
  hipLaunchKernelGGL(mirror_ch0_kernel, dim3(2), dim3(256), 0, (hipStream_t)stream, w, Cin, by_tap, by_channel);
  AS_CHECK_LAUNCH("as_mirror_taps_ch0");
  return AS_OK;
}

extern "C" int64_t as_sumsq_workspace(int64_t n) { return n > 0 ? 2 * SS_BLOCKS : -1; }

extern "C" int as_sumsq(const float* g, int64_t n, float* out, float* workspace, void* stream) {
  AS_CHECK_ARG(g && out && workspace && n > 0, "as_sumsq: bad argument");
  AS_CHECK_ARG(((uintptr_t)workspace & 7) == 0, "as_sumsq: workspace must be 8-byte aligned");
  long nb = (n + 255) / 256;
  if (nb > SS_BLOCKS) nb = SS_BLOCKS;
  double* partial = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(sumsq_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, g, (long)n, partial);
  AS_CHECK_LAUNCH("as_sumsq");
  hipLaunchKernelGGL(sumsq_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, (int)nb, out, 0.f,
                     (float*)nullptr, (float*)nullptr);
  AS_CHECK_LAUNCH("as_sumsq(finalize)");
  return AS_OK;
}

extern "C" int as_sumsq_clip(const float* g, int64_t n, float max_norm, float* out, float* coef, float* step_counter,
                             float* workspace, void* stream) {
  AS_CHECK_ARG(g && out && coef && workspace && n > 0 && max_norm > 0.f, "as_sumsq_clip: bad argument");
  AS_CHECK_ARG(((uintptr_t)workspace & 7) == 0, "as_sumsq_clip: workspace must be 8-byte aligned");
  long nb = (n + 255) / 256;
  if (nb > SS_BLOCKS) nb = SS_BLOCKS;
  double* partial = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(sumsq_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, g, (long)n, partial);
  AS_CHECK_LAUNCH("as_sumsq_clip");
  hipLaunchKernelGGL(sumsq_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, (int)nb, out, max_norm, coef,
                     step_counter);
  AS_CHECK_LAUNCH("as_sumsq_clip(finalize)");
  return AS_OK;
}

// torch.optim.Adam (single-tensor path): m = lerp(m, g, 1-b1); v = b2*v + (1-b2)*g*g;
// denom = sqrt(v)/sqrt(bc2) + eps; p -= (lr/bc1) * m / denom.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long n,
                                                    const float* __restrict__ grad_scale, float lr, float b1, float b2,
                                                    float eps, int step_host, const float* __restrict__ step_dev) {
  // bias corrections from the step count: a host integer, or (hipGraph replay) a device counter
  const double step = step_dev ? (double)step_dev[0] : (double)step_host;
  const float bc1 = (float)(1.0 - pow((double)b1, step));
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, step));
  const float gs = grad_scale ? grad_scale[0] : 1.f;
  const float step_size = lr / bc1;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gi = g[i] * gs;
    const float mi = m[i] + (1.f - b1) * (gi - m[i]);
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
  }
}

extern "C" int as_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                            const float* grad_scale_dev, float lr, float beta1, float beta2, float eps,
                            int step, const float* step_dev, void* stream) {
  AS_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0 && (step >= 1 || step_dev), "as_adam_step: bad argument");
  long nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(adam_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                     (long)n, grad_scale_dev, lr, beta1, beta2, eps, step, step_dev);
  AS_CHECK_LAUNCH("as_adam_step");
  return AS_OK;
}
