// LDS-staged 3x3 stride-1 2-D instance of conv32 (conv32_lds.hip); dispatched from as_conv32_fwd.
#pragma once
#include "as_common.h"
bool conv32_lds_applicable(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s);
int conv32_lds_grid(const as_pcl* gout);     // number of workgroups = number of BatchNorm partials
int conv32_lds_launch(const float* x, const as_pcl* gin, const float* packed_w, const float* bias,
                      float* z, const as_pcl* gout, const as_conv_shape* s,
                      int epilogue, const float* ep_scale, const float* ep_shift, float slope,
                      const float* residual, float* stat_mean, float* stat_m2, float* stat_cnt,
                      const float* bn_z, const float* bn_scale, const float* bn_shift, const float* bn_mean,
                      double* bn_partial, void* stream);   // bn_*: stage 1 of the following BatchNorm backward, or null

// LDS-staged weight gradient of the same instance (conv32_lds.hip); dispatched from as_conv32_wgrad.
int conv32_wgrad_lds_slabs(const as_pcl* gout);        // number of partial slabs it writes
struct WgradBnApply {          // stage 3 of the layer's BatchNorm backward applied to the staged gradient row
  const float* z; const float* scale; const float* shift; const float* mean; const float* coef;
  float* gz_out; float slope;
};
bool conv32_wgrad_bnapply_ok(const as_pcl* gout);      // the fused form exists for this launch size
int conv32_wgrad_lds_launch(const float* x, const as_pcl* gin, const float* gz, const as_pcl* gout,
                            const as_conv_shape* s, float* partial, float* partial_db, const WgradBnApply* bn,
                            void* stream);
