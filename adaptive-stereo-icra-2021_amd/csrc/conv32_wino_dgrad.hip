// Data gradient of a full-resolution refinement layer (3x3, dilation 1/2/4/8, stride 1, 32->32: stereo_net.py:10-18, 33-51,
// 97) by minimal filtering F(2x2, 3x3) — second generation of conv32_wino.hip's MODE 2, same arithmetic, same outputs:
//   g_z = stage 3 of the layer's BatchNorm backward applied to (g_a, z) on the way in (written once, for the weight gradient),
//   g_x = dgrad(g_z) + g_a (skip connection), per-workgroup sums of stage 1 of the NEXT BatchNorm backward from the g_x tile.
//
// What the first generation paid for in HBM bytes it did not need (PMC, round 3: 1.70 GB per launch at 4 pairs against 1.19 GB
// algorithmic, 5.6 TB/s on the counters — the kernel was bound by its own excess traffic):
//   * g_a was read TWICE: 80 staged voxels per 64-voxel row for the conversion, and again two to four tiles later at the output
//     pixels for the skip connection (16 + 16 one-dword loads per lane; by then the rows had left the XCD's 4-MB L2);
//   * rows were staged 8 + 64 + 8 voxels wide whatever the dilation, although a tile only reaches d voxels beyond its segment.
// Here the four waves of a workgroup have ROLES (separate instantiations of the whole loop under a wave branch, as in
// conv32_wino_wgrad.hip), which is what makes room for the fix in the 80 KB a workgroup may have at two per CU:
//   all four    matrix phase as before: wave r holds the transformed filters U[r][0..3] and accumulates M[r][c] (64 MFMAs),
//               forms T[j] = (M[r] A)[j]
//   waves 1, 2  ("inner": Y[0][j] and Y[1][j] both need T1 and T2) publish T[0], T[1] in LDS — 16 KB instead of the 32 KB all
//               four published — and do ALL of the row conversion: rows j+3, j+4 of g_a and z arrive in their registers
//               (64 + 2d voxels per row), become g_z in the two freed ring slots and the by-product in HBM, and the RAW g_a
//               chunks of the segment's own 64 voxels go to a small ring of raw rows in LDS
//   waves 0, 3  ("outer": T0 is only needed by Y[0][j], T3 only by Y[1][j]) keep their own T in registers, read T1, T2 and
//               finish output row j (wave 0) / j+1 (wave 3), BOTH column parities: skip connection from the raw ring
//               (ds_read_b32 with immediate offsets: the lane is the channel), next layer's pre-activation from HBM, 32 stores,
//               the next BatchNorm's sums
// Raw rows: the conversion of tile j brings comb rows j+3 (output row of tile j+2: one slot, read before barrier B1 of that tile,
// rewritten after it) and j+4 (output row of tile j+4: two alternating slots) — three slots of 8 KB.
// LDS: ring 4 x (64 + 2d) x 128 B | coefficients 768 B | exchange 16 KB | raw ring 24 KB = 75.5 / 76.5 / 78.6 KB for d = 1 / 2 / 4:
// two workgroups per CU as before.  With d = 8 the ring rows are 80 voxels and the three raw slots miss the 80 KB by the 768 bytes
// of the coefficient table (the coefficients in 24 registers per thread instead: the resident filters go to scratch, measured);
// the instantiation for d = 8 keeps one slot — wave 0 reads its skip rows from HBM as the first generation does — and is SLOWER
// than the first generation there (335 against 288 us at 4 pairs: 64 loads on wave 0): as_conv32_wino_bwd_data launches the
// first generation for d = 8.
//
// Bit-identical to the first generation (tests/test_gpu_kernels.py: g_z, g_x and the next-BatchNorm partials against
// conv32_wino_kernel<2, L>): same element-wise chains, same order in every sum, the same ownership of shared columns.
#include "as_common.h"
#include "conv32_wino.h"
#include "conv32_wino_dev.h"
#include "conv32_wino_dgrad_role.h"

template <int L>
__global__ __launch_bounds__(256, 2) void conv32_wino_dgrad_kernel(DgradArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_dg[];
  {
    float* tab = reinterpret_cast<float*>(smem_dg + DgGeo<L>::COEF_OFF);
    const int i = threadIdx.x;
    if (i < 96) tab[i] = p.bn_coef[i];
    else if (i < 128) tab[i] = p.in_scale[i - 96];
    else if (i < 160) tab[i] = p.in_shift[i - 128];
    else if (i < 192) tab[i] = p.bn_mean[i - 160];
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef DG_ONLY_ROLE                                       // (diagnostic: one role's register count — results are wrong)
  dgrad_role<DG_ONLY_ROLE, L>(p, smem_dg);
#else
  switch (wave) {
    case 0: dgrad_role<0, L>(p, smem_dg); break;
    case 1: dgrad_role<1, L>(p, smem_dg); break;
    case 2: dgrad_role<2, L>(p, smem_dg); break;
    default: dgrad_role<3, L>(p, smem_dg); break;
  }
#endif
}

int conv32_wino_dgrad2_parts(void) { return DG_GRID; }

int conv32_wino_dgrad2_launch(const float* g_a, const float* z, const as_pcl* g, const as_conv_shape* s, const float* wino_wt,
                              const float* scale, const float* shift, const float* mean, const float* coef, float slope,
                              const float* next_z, const float* next_scale, const float* next_shift, const float* next_mean,
                              float* g_z, float* g_x, double* next_partial, void* stream) {
  static AsPerDevice attr_set[4];
  const int L = s->dil == 1 ? 0 : (s->dil == 2 ? 1 : (s->dil == 4 ? 2 : 3));
  const void* fn = L == 0 ? reinterpret_cast<const void*>(conv32_wino_dgrad_kernel<0>)
                 : L == 1 ? reinterpret_cast<const void*>(conv32_wino_dgrad_kernel<1>)
                 : L == 2 ? reinterpret_cast<const void*>(conv32_wino_dgrad_kernel<2>)
                          : reinterpret_cast<const void*>(conv32_wino_dgrad_kernel<3>);
  const int lds = L == 0 ? DgGeo<0>::LDS : (L == 1 ? DgGeo<1>::LDS : (L == 2 ? DgGeo<2>::LDS : DgGeo<3>::LDS));
  if (!attr_set[L].get()) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { as_set_error("as_conv32_wino_bwd_data: %s", hipGetErrorString(e)); return AS_ERR_LAUNCH; }
    attr_set[L].set();
  }
  DgradArgs a;
  a.z = z; a.g_a = g_a; a.g_z = g_z; a.g_x = g_x; a.wq = wino_wt;
  a.in_scale = scale; a.in_shift = shift; a.bn_mean = mean; a.bn_coef = coef;
  a.nz = next_z; a.n_scale = next_scale; a.n_shift = next_shift; a.n_mean = next_mean; a.n_partial = next_partial;
  a.g = as_make_dev(g);
  a.nseg = (g->W + 63) / 64;
  long pairs = 0;
  for (int r = 0; r < s->dil; ++r) pairs += ((g->H - r + s->dil - 1) / s->dil + 1) / 2;
  a.pairs = (int)pairs; a.slope = slope;
  void* kargs[] = {&a};
  hipError_t le = hipLaunchKernel(fn, dim3(DG_GRID), dim3(256), kargs, lds, (hipStream_t)stream);
  if (le != hipSuccess) { as_set_error("as_conv32_wino_bwd_data: launch failed: %s", hipGetErrorString(le)); return AS_ERR_LAUNCH; }
  return AS_OK;
}
