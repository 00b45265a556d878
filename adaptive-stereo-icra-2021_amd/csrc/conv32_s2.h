// 5x5 stride-2 32->32 convolution of the feature towers' head (forward and data gradient) with coalesced row staging through
// wave-private LDS (conv32_s2.hip); dispatched from as_conv32_fwd / as_conv32_dgrad_s2_packed for maps that fill the chip.
#pragma once
#include "as_common.h"
bool conv32_s2_fwd_applicable(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s);
int conv32_s2_fwd_launch(const float* x, const as_pcl* gin, const float* packed_w, const float* bias, float* z,
                         const as_pcl* gout, void* stream);
bool conv32_s2_dgrad_applicable(const as_pcl* ggz, const as_pcl* ggx);
int conv32_s2_dgrad_launch(const float* gz, const as_pcl* ggz, const float* packed, float* gx, const as_pcl* ggx, void* stream);
