// BatchNorm finalize done by the CONSUMER of a layer's statistics: every workgroup of the kernel that applies the affine
// merges the producer's per-workgroup (count, mean, M2) partials itself while its first operand planes are still in flight
// (plain cached loads after a kernel boundary: no hand-off protocol, no finalize launch, no ticket), all of them in the same
// fixed order, so every workgroup holds the same bits; workgroup 0 also writes the layer's state (mean, invstd, scale,
// shift: the backward pass reads them) and updates the running statistics.  Arithmetic: bn_finalize_kernel's merge
// (mean = sum n_i*mean_i / N;  M2 = sum [ M2_i + n_i*(mean_i - mean)^2 ]) in its one-pass form, fp64.
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

#define BN_MERGE_SCRATCH_BYTES (8 * 32 * 3 * 8 + 64 * 4)      // [8][32][3] doubles + scale[32] + shift[32]

// All 32 * SLICES threads of the workgroup call (it contains two barriers): 256 by default, 128 with SLICES = 4.  On return tab[0..31] = scale, tab[32..63] = shift,
// where tab = (float*)(scratch + 6144).  `publish`: this workgroup writes the layer state and the running statistics.
// ONE pass over the partials (a consumer waits for this: the two-pass form costs two dependent rounds of loads more):
// with a pivot K (the first partial's mean) the sums  S0 = sum n_i,  S1 = sum n_i (mean_i - K),
// S2 = sum [ M2_i + n_i (mean_i - K)^2 ]  give  mean = K + S1/S0  and  M2 = S2 - S1^2/S0  — algebraically the two-pass
// result, and in fp64 (the terms are fp32 data, the pivot is within the data's range) equal to it to ~1e-15 relative.
template <int BATCH, int SLICES = 8>      // BATCH: partials per thread whose loads are in flight together (3 x BATCH registers)
__device__ inline float* bn_merge_partials(const BnMergeDev& m, char* scratch, bool publish) {
  double* red = reinterpret_cast<double*>(scratch);            // [8][32][3]
  float* tab = reinterpret_cast<float*>(scratch + 8 * 32 * 3 * 8);
  const int c = threadIdx.x & 31, slc = threadIdx.x >> 5;
  const int per_slice = (m.nparts + SLICES - 1) / SLICES;
  const double K = (double)m.stat_mean[c];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (int j0 = 0; j0 < per_slice; j0 += BATCH) {
    float pn[BATCH], pm[BATCH], pq[BATCH];
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {                       // every load of the round in flight together
      const int i = slc + SLICES * (j0 + j);
      const bool ok = i < m.nparts;
      const int ii = ok ? i : 0;
      pn[j] = ok ? m.stat_cnt[ii] : 0.f;
      pm[j] = m.stat_mean[ii * 32 + c];
      pq[j] = ok ? m.stat_m2[ii * 32 + c] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const double n = (double)pn[j], dm = (double)pm[j] - K;
      s0 += n; s1 += n * dm; s2 += (double)pq[j] + n * dm * dm;
    }
  }
  double* mine = red + (slc * 32 + c) * 3;
  mine[0] = s0; mine[1] = s1; mine[2] = s2;
  __syncthreads();
  if (slc == 0) {
    double t0 = 0.0, t1 = 0.0, t2 = 0.0;
    for (int j = 0; j < SLICES; ++j) { const double* r = red + (j * 32 + c) * 3; t0 += r[0]; t1 += r[1]; t2 += r[2]; }
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
