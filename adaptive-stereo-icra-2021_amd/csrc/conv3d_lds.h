// LDS-staged 3x3x3 stride-1 instance of conv32 (conv3d_lds.hip); dispatched from as_conv32_fwd.
#pragma once
#include "as_common.h"
bool conv3d_lds_applicable(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s);
int conv3d_lds_grid(const as_pcl* gout);     // number of workgroups = number of BatchNorm partials
int conv3d_lds_launch(const float* x, const as_pcl* gin, const float* packed_w, const float* bias, float* z,
                      const as_pcl* gout, int epilogue, const float* ep_scale, const float* ep_shift, float slope,
                      const float* residual, float* stat_mean, float* stat_m2, float* stat_cnt, void* stream);

// LDS-staged weight gradient of the same instance; dispatched from as_conv32_wgrad.
bool conv3d_wgrad_lds_applicable(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s);
int conv3d_wgrad_lds_slabs(const as_pcl* gout);        // number of [27][32][32] partial slabs it writes
int conv3d_wgrad_lds_launch(const float* x, const as_pcl* gin, const float* gz, const as_pcl* gout,
                            float* partial, float* partial_db, void* stream);

// Which (chunk, kd) a workgroup of conv3d_wgrad_lds_kernel owns and which extra tile (or -1) it takes on top of its `full` rounds
// of tile = chunk + k * nchunks.  XCD-aware: block ids congruent mod 8 share an XCD; an XCD owns a contiguous range of
// per_xcd = ceil(nchunks / 8) chunks and all three kd of a chunk.  The ntiles - full * nchunks remaining tiles go one each to the
// chunks that are dispatched FIRST on every XCD, counted over REAL chunks only: a padding block (chunk >= nchunks) returns false
// and holds no rank (round 4 numbered the ranks over the padded grid: whenever nchunks was not a multiple of 8 and tiles
// remained, the ranks of padding blocks were lost and their tiles never accumulated — unreachable with the launch's 168 = 8 x 21
// chunks, but nothing enforced that).  Host and device: the CPU suite checks that every tile is covered exactly three times
// (as_conv3d_wgrad_lds_assignment).
__host__ __device__ inline bool conv3d_wgrad_assign(int block, int ntiles, int nchunks, int* chunk, int* kd, int* extra_tile) {
  const int xcd = block & 7, q = block >> 3, i = q / 3;
  const int per_xcd = (nchunks + 7) >> 3;
  *kd = q - 3 * i;
  *chunk = xcd * per_xcd + i;
  *extra_tile = -1;
  if (*chunk >= nchunks || i >= per_xcd) return false;
  int rank = xcd;                                            // real chunks dispatched before this one: rows i' < i, then xcd' < xcd
  for (int r = 0; r < i; ++r) {
    const int n_r = (nchunks - r + per_xcd - 1) / per_xcd;   // XCDs whose chunk of row r is real (they are 0 .. n_r - 1)
    rank += n_r < 8 ? n_r : 8;
  }
  const int full = ntiles / nchunks;
  if (rank < ntiles - full * nchunks) *extra_tile = full * nchunks + rank;
  return true;
}
