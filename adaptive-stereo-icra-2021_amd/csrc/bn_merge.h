// BatchNorm finalize done by the CONSUMER of a layer's statistics: every workgroup of the kernel that applies the affine
// merges the producer's per-workgroup (count, mean, M2) partials itself while its first operand planes are still in flight
// (plain cached loads after a kernel boundary: no hand-off protocol, no finalize launch, no ticket), all of them in the same
// fixed order, so every workgroup holds the same bits; workgroup 0 also writes the layer's state (mean, invstd, scale,
// shift: the backward pass reads them) and updates the running statistics.  Arithmetic = bn_finalize_kernel's exact
// two-pass merge in fp64:  mean = sum n_i*mean_i / N;  M2 = sum [ M2_i + n_i*(mean_i - mean)^2 ].
#pragma once
#include "as_common.h"

struct BnMergeDev {
  const float* stat_mean;    // [nparts][32]
  const float* stat_m2;      // [nparts][32]
  const float* stat_cnt;     // [nparts]
  const float* gamma;
  const float* beta;
  float* running_mean;       // may be null (with running_var)
  float* running_var;
  float* save_mean;          // [32] each, written by workgroup 0
  float* save_invstd;
  float* scale;
  float* shift;
  int nparts;
  float momentum, eps;
};

#define BN_MERGE_SCRATCH_BYTES (8 * 33 * 8 + 64 * 4)      // [8][33] doubles + scale[32] + shift[32]

// All 256 threads of the workgroup call (it contains barriers).  On return tab[0..31] = scale, tab[32..63] = shift, where
// tab = (float*)(scratch + 8*33*8).  `publish`: this workgroup writes the layer state and the running statistics.
__device__ inline float* bn_merge_partials(const BnMergeDev& m, char* scratch, bool publish) {
  double* red = reinterpret_cast<double*>(scratch);
  float* tab = reinterpret_cast<float*>(scratch + 8 * 33 * 8);
  const int c = threadIdx.x & 31, slc = threadIdx.x >> 5;
  const int per_slice = (m.nparts + 7) >> 3;
  double s = 0.0, cn = 0.0;
  for (int j0 = 0; j0 < per_slice; j0 += 16) {
    float pn[16], pm[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {                          // sixteen independent loads in flight per round
      const int i = slc + 8 * (j0 + j);
      const bool ok = i < m.nparts;
      const int ii = ok ? i : 0;
      pn[j] = ok ? m.stat_cnt[ii] : 0.f;
      pm[j] = m.stat_mean[ii * 32 + c];
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) { cn += (double)pn[j]; s += (double)pn[j] * (double)pm[j]; }
  }
  red[slc * 33 + c] = s;
  __syncthreads();
  double tot = 0.0;
  for (int j = 0; j < 8; ++j) tot += red[j * 33 + c];
  __syncthreads();
  red[slc * 33 + c] = cn;
  __syncthreads();
  double count = 0.0;
  for (int j = 0; j < 8; ++j) count += red[j * 33 + c];
  __syncthreads();
  const double mean = tot / count;
  double qq = 0.0;
  for (int j0 = 0; j0 < per_slice; j0 += 16) {
    float pn[16], pm[16], pq[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int i = slc + 8 * (j0 + j);
      const bool ok = i < m.nparts;
      const int ii = ok ? i : 0;
      pn[j] = ok ? m.stat_cnt[ii] : 0.f;
      pm[j] = m.stat_mean[ii * 32 + c];
      pq[j] = ok ? m.stat_m2[ii * 32 + c] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) { const double dm = (double)pm[j] - mean; qq += (double)pq[j] + (double)pn[j] * dm * dm; }
  }
  red[slc * 33 + c] = qq;
  __syncthreads();
  if (slc == 0) {
    double m2 = 0.0;
    for (int j = 0; j < 8; ++j) m2 += red[j * 33 + c];
    const double var_b = m2 / count;
    const float invstd = (float)(1.0 / sqrt(var_b + (double)m.eps));
    const float meanf = (float)mean;
    const float scl = invstd * m.gamma[c];
    const float shf = m.beta[c] - meanf * scl;
    tab[c] = scl; tab[32 + c] = shf;
    if (publish) {
      m.save_mean[c] = meanf; m.save_invstd[c] = invstd; m.scale[c] = scl; m.shift[c] = shf;
      if (m.running_mean) {
        const double var_u = count > 1.0 ? m2 / (count - 1.0) : var_b;
        const double mo = (double)m.momentum;
        m.running_mean[c] = (float)(mo * mean + (1.0 - mo) * (double)m.running_mean[c]);
        m.running_var[c] = (float)(mo * var_u + (1.0 - mo) * (double)m.running_var[c]);
      }
    }
  }
  __syncthreads();
  return tab;
}

static inline int bn_merge_fill(BnMergeDev* d, const as_bn_merge* m) {
  if (!(m->stat_mean && m->stat_m2 && m->stat_cnt && m->nparts >= 1 && m->gamma && m->beta && m->save_mean && m->save_invstd &&
        m->scale && m->shift && (m->running_mean == nullptr) == (m->running_var == nullptr)))
    return 0;
  d->stat_mean = m->stat_mean; d->stat_m2 = m->stat_m2; d->stat_cnt = m->stat_cnt; d->nparts = m->nparts;
  d->gamma = m->gamma; d->beta = m->beta; d->running_mean = m->running_mean; d->running_var = m->running_var;
  d->save_mean = m->save_mean; d->save_invstd = m->save_invstd; d->scale = m->scale; d->shift = m->shift;
  d->momentum = m->momentum; d->eps = m->eps;
  return 1;
}
