// Training forward of a full-resolution layer with the previous layer's BatchNorm + LeakyReLU (+ skip) applied to its
// operand in LDS (conv32_act.hip); dispatched from as_conv32_act_fwd.
#pragma once
#include "as_common.h"
bool conv32_act_applicable(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s);
int conv32_act_parts(void);          // workgroups of a launch = BatchNorm partials it writes
int conv32_act_launch(const float* z_prev, const float* a_prevprev, const float* in_scale, const float* in_shift, float* a_out,
                      const as_pcl* g, const as_conv_shape* s, const float* packed_w, const float* bias, float slope,
                      float* z, float* stat_mean, float* stat_m2, float* stat_cnt, void* stream);
