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
