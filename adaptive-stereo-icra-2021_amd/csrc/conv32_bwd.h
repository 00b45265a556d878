// Fused backward of a full-resolution 3x3 refinement layer (conv32_bwd.hip); dispatched from as_conv32_bwd_fused.
#pragma once
#include "as_common.h"
bool conv32_bwd_fused_applicable(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s);
int conv32_bwd_fused_slabs(void);     // workgroups of a launch = weight-gradient slabs = next-BatchNorm partials
int conv32_bwd_fused_launch(const float* x, const float* g_a, const float* z, const as_pcl* g, const as_conv_shape* s,
                            const float* packed_wt, const float* scale, const float* shift, const float* mean,
                            const float* coef, float slope, const float* next_z, const float* next_scale,
                            const float* next_shift, const float* next_mean, float* g_x, float* partial,
                            float* partial_db, double* next_partial, void* stream);
