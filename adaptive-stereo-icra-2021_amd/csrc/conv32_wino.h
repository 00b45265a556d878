// Training forward of a full-resolution layer (operand formed on the way in, as conv32_act.hip) by the minimal-filtering
// algorithm F(2x2, 3x3) (conv32_wino.hip); dispatched from as_conv32_wino_fwd.
#pragma once
#include "as_common.h"
bool conv32_wino_applicable(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s);
int conv32_wino_parts(void);         // workgroups of a launch = BatchNorm partials it writes
int conv32_wino_launch(const float* z_prev, const float* a_prevprev, const float* in_scale, const float* in_shift, float* a_out,
                       const as_pcl* g, const as_conv_shape* s, const float* wino_w, const float* bias, float slope,
                       float* z, float* stat_mean, float* stat_m2, float* stat_cnt, void* stream);
// Eval-mode forward of a BasicBlock (conv + folded BatchNorm + LeakyReLU + skip connection) by minimal filtering.
int conv32_wino_eval_launch(const float* x, const as_pcl* g, const as_conv_shape* s, const float* wino_w, const float* bias,
                            const float* scale, const float* shift, float slope, int residual, float* out, void* stream);
// Backward of such a layer in two launches: MODE 2 of conv32_wino.hip (g_z, g_x = dgrad(g_z) + g_a, next-BatchNorm sums) and
// conv32_wino_wgrad.hip (weight / bias gradient slabs from x and g_z); dispatched from as_conv32_wino_bwd.
int conv32_wino_dgrad_launch(const float* g_a, const float* z, const as_pcl* g, const as_conv_shape* s, const float* wino_wt,
                             const float* scale, const float* shift, const float* mean, const float* coef, float slope,
                             const float* next_z, const float* next_scale, const float* next_shift, const float* next_mean,
                             float* g_z, float* g_x, double* next_partial, void* stream);
int conv32_wino_dgrad_parts(void);     // workgroups of the data-gradient launch = next-BatchNorm partials it writes
// second generation of that data gradient (conv32_wino_dgrad.hip: waves with roles, raw g_a rows kept in LDS for the skip
// connection, rows staged 64 + 2d voxels wide): same arguments, bit-identical results, the same number of partials
int conv32_wino_dgrad2_launch(const float* g_a, const float* z, const as_pcl* g, const as_conv_shape* s, const float* wino_wt,
                              const float* scale, const float* shift, const float* mean, const float* coef, float slope,
                              const float* next_z, const float* next_scale, const float* next_shift, const float* next_mean,
                              float* g_z, float* g_x, double* next_partial, void* stream);
int conv32_wino_dgrad2_parts(void);
int conv32_wino_wgrad_slabs(void);
int conv32_wino_wgrad_launch(const float* x, const float* g_z, const as_pcl* g, const as_conv_shape* s, float* partial,
                             float* partial_db, void* stream);
// The whole backward in ONE launch (conv32_wino_bwd.hip): both gradients side by side in an 8-wave workgroup per CU, g_z never
// leaves the chip; partial / partial_db: conv32_wino_bwd_fused_parts() slabs, next_partial as many [64]-double partials
int conv32_wino_bwd_fused_parts(void);
int conv32_wino_bwd_fused_launch(const float* x, const float* g_a, const float* z, const as_pcl* g, const as_conv_shape* s,
                                 const float* wino_wt, const float* scale, const float* shift, const float* mean, const float* coef,
                                 float slope, const float* next_z, const float* next_scale, const float* next_shift,
                                 const float* next_mean, float* g_x, float* partial, float* partial_db, double* next_partial,
                                 void* stream);
