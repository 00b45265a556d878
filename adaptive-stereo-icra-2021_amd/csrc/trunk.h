// Small-map trunk kernels of the feature extractor (trunk.hip).
#pragma once
#include "as_common.h"
#define TRUNK_MAX_LAYERS 8
#define TRUNK_MAX_GROUPS 2
