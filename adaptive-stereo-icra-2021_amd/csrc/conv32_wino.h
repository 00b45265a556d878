// Training forward of a full-resolution layer (operand formed on the way in, as conv32_act.hip) by the minimal-filtering
// algorithm F(2x2, 3x3) (conv32_wino.hip); dispatched from as_conv32_wino_fwd.
#pragma once
#include "as_common.h"
bool conv32_wino_applicable(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s);
int conv32_wino_parts(void);         // workgroups of a launch = BatchNorm partials it writes
int conv32_wino_launch(const float* z_prev, const float* a_prevprev, const float* in_scale, const float* in_shift, float* a_out,
                       const as_pcl* g, const as_conv_shape* s, const float* wino_w, const float* bias, float slope,
                       float* z, float* stat_mean, float* stat_m2, float* stat_cnt, void* stream);
