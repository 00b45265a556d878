// Rolling-window 3-D aggregation kernels (agg3d.hip).
#pragma once
#include "as_common.h"
bool agg3d_applicable(const as_pcl* g);
int agg3d_units(const as_pcl* g);      // workgroups of a launch = BatchNorm partials it writes
