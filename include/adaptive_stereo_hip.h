/*
 * adaptive_stereo_hip.h — C ABI of libadaptive_stereo_hip.so (gfx950 / MI355X).
 *
 * This is the drop-in boundary for the StereoNet online-adaptation hot path
 * (SURVEY.md §8).  The reference (miloknowles/adaptive-stereo-icra-2021) is pure
 * Python/PyTorch and has no FFI of its own: the interfaces these entry points
 * replace are the PyTorch op sequences inside the reference's nn.Module.forward()
 * bodies and loss functions, cited per function below (paths relative to the
 * reference root).  INTEGRATION.md shows the ctypes binding a maintainer of the
 * reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 unless stated otherwise;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, no
 *     entry point synchronises, allocates or frees (safe under hipGraph capture);
 *   - the caller owns every buffer; nothing is retained after return;
 *   - return value: 0 on success, negative on error; as_last_error() returns a
 *     thread-local description of the last failure;
 *   - "NCHW"/"NCDHW" are the reference's contiguous PyTorch layouts;
 *   - "PCL" (padded channel-last) is this library's internal activation layout:
 *       float buf[B][D+2*pd][H+2*ph][W+2*pw][32]
 *     with a zero halo that kernels never write.  Halo zeros implement the
 *     convolutions' zero padding, so no kernel needs bounds predicates, and the
 *     32 channels of a voxel are one aligned 128-byte line.  2-D maps use D=1, pd=0.
 */
#ifndef ADAPTIVE_STEREO_HIP_H_
#define ADAPTIVE_STEREO_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AS_CHANNELS 32
#define AS_MAX_TAPS 32

/* Geometry of one PCL tensor. */
typedef struct as_pcl {
  int32_t B, D, H, W;     /* logical extent */
  int32_t pd, ph, pw;     /* halo width on each side of D, H, W */
} as_pcl;

/* Shape of a 32->32 convolution over PCL tensors. kd=1 for 2-D. */
typedef struct as_conv_shape {
  int32_t kd, kh, kw;     /* kernel extent */
  int32_t pad_d, pad_h, pad_w;   /* logical zero padding (must be <= input halo) */
  int32_t dil;            /* dilation on H and W (and D when kd>1) */
  int32_t stride;         /* stride on H and W (D always 1) */
} as_conv_shape;

const char* as_last_error(void);
int  as_version(void);
/* Number of floats a PCL tensor occupies. */
int64_t as_pcl_numel(const as_pcl* g);

/* ---- a2: difference cost volume — stereo_net.py:173-184 -------------------
 * vol[b,d,y,x,c] = L[b,c,y,x] - R[b,c,y,x-d]  (x >= d), else 0.
 * L, R: NCHW [B,32,H,W].  vol: PCL {B,D,H,W}.  bwd adds nothing: it overwrites gL, gR. */
int as_cost_volume_fwd(const float* L, const float* R, float* vol, const as_pcl* g, void* stream);
int as_cost_volume_bwd(const float* gvol, float* gL, float* gR, const as_pcl* g, void* stream);

/* ---- a3 (and a1/a7 32->32 convs): fp32-MFMA implicit-GEMM convolution ------
 * Replaces nn.Conv3d(32,32,3,padding=1) of convbn_3d (stereo_net.py:21-30,185-186)
 * and the 32->32 nn.Conv2d layers (stereo_net.py:10-18).
 *
 * as_conv32_pack_weights: w is the PyTorch weight [32(out)][32(in)][kd][kh][kw];
 *   packed is [taps][2][32][16] floats in MFMA operand order.  transpose_flip=1
 *   produces the weights of the data-gradient convolution (in/out swapped, taps
 *   mirrored), so dgrad is the same kernel as forward.
 * as_conv32_fwd: z = conv(x) + bias into the interior of PCL z.
 *   epilogue 0: raw output (+ residual[v][c] if given); if stat_mean != NULL also writes
 *               BatchNorm partials: stat_mean/stat_m2 [parts][32], stat_cnt [parts] with
 *               parts = as_conv32_stat_parts() (capacity as_conv32_num_blocks() always suffices);
 *   epilogue 1: z = lrelu(acc*ep_scale[c] + ep_shift[c]) (+ residual[v][c] if given):
 *               eval-mode BatchNorm + LeakyReLU (+ BasicBlock skip) fused.
 *   z must not alias x or residual (2-D 3x3 layers with W >= 128 run on an LDS-staged kernel whose last
 *   row segment overlaps its neighbour: the overlap is computed and stored twice).  The halo of z is never
 *   written.
 * as_conv32_wgrad: dW[o][i][tap] (PyTorch layout) = sum_v x[v+tap][i] * gz[v][o];
 *   workspace must hold as_conv32_wgrad_workspace() floats. db (may be NULL) gets
 *   sum_v gz[v][o].  accumulate=1 adds into dW/db instead of overwriting them (all parameter-gradient
 *   outputs of this library have this flag: it lets a caller point them at a flat gradient arena and
 *   skip one "grad += dW" launch per parameter tensor). */
int as_conv32_pack_weights(const float* w, float* packed, const as_conv_shape* s,
                           int transpose_flip, void* stream);

/* One launch for many packings (every 32->32 convolution weight of a step, both orientations).
 * `jobs` is a DEVICE array; each job is one as_conv32_pack_weights call (taps = kd*kh*kw <= max_taps). */
typedef struct as_pack_job {
  const float* w;
  float* packed;
  int32_t taps;
  int32_t transpose_flip;   /* 0 / 1, or one of the AS_PACK_* kinds below: every weight-derived buffer of a step from ONE launch */
} as_pack_job;
#define AS_PACK_S2_DGRAD 2      /* as_conv32_dgrad_s2_pack (taps = 25) */
#define AS_PACK_CONV4 16        /* + Cin: as_conv4_pack_weights for Cin input channels (packed: taps * 128 floats) */
#define AS_PACK_MIRROR_TAP 32   /* as_mirror_taps_ch0's by_tap [9][32] of a [32][4][3][3] weight */
#define AS_PACK_MIRROR_CH 33    /* ... its by_channel [32][9] */
#define AS_PACK_WINO 34         /* as_conv32_wino_pack_weights(transposed = 0): taps = 16 (packed: 16 * 1024 floats) */
#define AS_PACK_WINO_T 35       /* ... transposed = 1: the data gradient's filter */
int as_conv32_pack_weights_batch(const as_pack_job* jobs, int njobs, int max_taps, void* stream);
int as_conv32_num_blocks(const as_pcl* gout);
int as_conv32_stat_parts(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s);
int as_conv32_fwd(const float* x, const as_pcl* gin, const float* packed_w, const float* bias,
                  float* z, const as_pcl* gout, const as_conv_shape* s,
                  int epilogue, const float* ep_scale, const float* ep_shift, float slope,
                  const float* residual, float* stat_mean, float* stat_m2, float* stat_cnt, void* stream);
/* ---- whole backward of a full-resolution refinement layer in ONE launch (csrc/conv32_bwd.hip): nn.Conv2d(32,32,3,dilation) +
 * BatchNorm2d + LeakyReLU + skip connection of a BasicBlock (stereo_net.py:10-18,33-51,97).  Replaces
 * as_conv32_wgrad_bnapply followed by as_conv32_fwd_bnbwd; g_z never leaves the chip.
 *   x, g_a, z, next_z, g_x   PCL tensors of ONE padded geometry (gin == gout; as_conv32_bwd_fused_ok() == 1)
 *   packed_wt                as_conv32_pack_weights(..., transpose_flip = 1)
 *   scale, shift, mean, coef the layer's BatchNorm state and the stage-3 coefficients [96] (as_bn_act_bwd(_given) with
 *                            g_z = NULL leaves them at workspace + as_bn_bwd_coef_offset())
 *   next_*                   the BatchNorm whose output gradient g_x is: its stage-1 sums go to next_bn_workspace
 *                            (as_bn_bwd_workspace floats; as_conv32_bwd_fused_parts() partials, for as_bn_act_bwd_given)
 *   dW [32,32,3,3], db [32]  set or accumulated; workspace: as_conv32_bwd_fused_workspace() floats */
int as_conv32_bwd_fused_ok(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s);
int as_conv32_bwd_fused_parts(void);
int64_t as_conv32_bwd_fused_workspace(void);
int as_conv32_bwd_fused(const float* x, const as_pcl* gin, const float* g_a, const float* z, const as_pcl* gout,
                        const as_conv_shape* s, const float* packed_wt, const float* scale, const float* shift,
                        const float* mean, const float* coef, float slope, const float* next_z,
                        const float* next_scale, const float* next_shift, const float* next_mean, float* g_x,
                        float* dW, float* db, int accumulate, float* next_bn_workspace, float* workspace, void* stream);

/* ---- deferred weight-gradient reductions.  Every weight-gradient entry point of the 32->32 family (as_conv32_wgrad,
 * as_conv32_wgrad_bnapply, as_conv32_bwd_fused) ends in a small reduction of per-workgroup slabs into dW / db.  Between
 * as_wgrad_defer(1) and as_wgrad_defer_flush() those reductions are only recorded and then run in ONE launch, the jobs of a
 * destination in recording order with the arithmetic of the separate launches.  The caller keeps the workspaces it passed
 * alive until the flush.  as_wgrad_defer returns the previous setting; the record is process-global (autograd runs backward
 * on its own thread). */
int as_wgrad_defer(int on);
int as_wgrad_defer_pending(void);
int as_wgrad_defer_flush(void* stream);

/* ---- training forward of a full-resolution refinement layer that forms its operand on the way in (csrc/conv32_act.hip):
 * the PREVIOUS BasicBlock's BatchNorm2d + LeakyReLU + skip connection (stereo_net.py:10-18,33-51,97) are applied to that
 * block's pre-activation while it is staged, the activated tensor is written back once as a by-product (the backward pass
 * and the next skip connection need it), and the layer's own convolution + BatchNorm moments follow.  Replaces
 * as_bn_act_fwd (previous layer) followed by as_conv32_fwd (this layer).
 *   z_prev, a_prevprev, a_out, z   PCL tensors of ONE padded geometry (gin == gout; as_conv32_act_ok() == 1)
 *   a_out = lrelu(z_prev*in_scale + in_shift) (+ a_prevprev if not NULL);  z = conv(a_out) + bias
 *   packed_w   as_conv32_pack_weights(..., transpose_flip = 0);  stat_*: (count, mean, M2) partials, as_conv32_act_parts()
 *   of them, all three or none */
int as_conv32_act_ok(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s);
int as_conv32_act_parts(void);
int as_conv32_act_fwd(const float* z_prev, const float* a_prevprev, const float* in_scale, const float* in_shift,
                      float* a_out, const as_pcl* gin, const float* packed_w, const float* bias, float slope,
                      float* z, const as_pcl* gout, const as_conv_shape* s, float* stat_mean, float* stat_m2,
                      float* stat_cnt, void* stream);

/* ---- the same layer (same arguments, same outputs) by the minimal-filtering algorithm F(2x2, 3x3) (csrc/conv32_wino.hip):
 * four 32x32 matrix products per output pixel where the direct form has nine; dilation 1, 2, 4 or 8; rows of >= 64 pixels.
 * The result differs from as_conv32_act_fwd's by fp32 rounding only (a different, equally valid association).
 *   wino_w     as_conv32_wino_pack_weights(w, out, transposed = 0) — 16 * 1024 floats — or a batch job of kind AS_PACK_WINO
 *   stat_*     as_conv32_wino_parts() partials */
int as_conv32_wino_ok(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s);
int as_conv32_wino_parts(void);
int as_conv32_wino_pack_weights(const float* w, float* packed, int transposed, void* stream);
int as_conv32_wino_fwd(const float* z_prev, const float* a_prevprev, const float* in_scale, const float* in_shift,
                       float* a_out, const as_pcl* gin, const float* wino_w, const float* bias, float slope,
                       float* z, const as_pcl* gout, const as_conv_shape* s, float* stat_mean, float* stat_m2,
                       float* stat_cnt, void* stream);

/* ---- eval-mode BasicBlock (stereo_net.py:10-18 with BatchNorm folded to an affine) by minimal filtering:
 *   out = lrelu((conv(x) + bias) * scale + shift) (+ x if residual != 0);  x, out: PCL tensors of geometry g (zero halo),
 *   as_conv32_wino_ok(g, g, s) == 1;  wino_w as for as_conv32_wino_fwd;  bias may be NULL */
int as_conv32_wino_eval(const float* x, const as_pcl* g, const as_conv_shape* s, const float* wino_w, const float* bias,
                        const float* scale, const float* shift, float slope, int residual, float* out, void* stream);

/* ---- backward of that layer by minimal filtering, two launches (csrc/conv32_wino.hip MODE 2, csrc/conv32_wino_wgrad.hip):
 * arguments and results of as_conv32_bwd_fused (below), except
 *   wino_wt    as_conv32_wino_pack_weights(w, out, transposed = 1) or a batch job of kind AS_PACK_WINO_T
 *   g_z        a PCL buffer of the layer's geometry with a zero halo: receives the gradient w.r.t. the pre-activation
 *   next_bn_workspace   as_conv32_wino_bwd_parts() partials of [64] doubles;  workspace: as_conv32_wino_bwd_workspace() floats */
int as_conv32_wino_bwd_parts(void);
int64_t as_conv32_wino_bwd_workspace(void);
/* which data-gradient kernel as_conv32_wino_bwd_data launches: 2 (default) = csrc/conv32_wino_dgrad.hip — the waves of a
 * workgroup have roles, the skip connection's g_a rows stay in LDS between conversion and output (one HBM read of g_a instead
 * of two), rows are staged 64 + 2 * dilation voxels wide — for dilation 1, 2, 4 and csrc/conv32_wino.hip MODE 2 for dilation 8
 * (where the second generation's LDS ring does not fit and it is the slower one); 1 = csrc/conv32_wino.hip MODE 2 always;
 * 3 = csrc/conv32_wino_dgrad.hip always (parity tests).  Same bits in every case (what the parity tests hold them to);
 * returns the previous setting, any other argument only queries */
int as_conv32_wino_bwd_generation(int generation);
/* the two launches separately (the weight gradient may go to another stream: only the step's slab reduction waits for it) */
int as_conv32_wino_bwd_data(const float* g_a, const float* z, const as_pcl* g, const as_conv_shape* s,
                            const float* wino_wt, const float* scale, const float* shift, const float* mean,
                            const float* coef, float slope, const float* next_z, const float* next_scale,
                            const float* next_shift, const float* next_mean, float* g_z, float* g_x,
                            float* next_bn_workspace, void* stream);
int as_conv32_wino_bwd_filter(const float* x, const float* g_z, const as_pcl* g, const as_conv_shape* s, float* dW,
                              float* db, int accumulate, float* workspace, void* stream);
int as_conv32_wino_bwd(const float* x, const as_pcl* gin, const float* g_a, const float* z, const as_pcl* gout,
                       const as_conv_shape* s, const float* wino_wt, const float* scale, const float* shift,
                       const float* mean, const float* coef, float slope, const float* next_z,
                       const float* next_scale, const float* next_shift, const float* next_mean, float* g_z,
                       float* g_x, float* dW, float* db, int accumulate, float* next_bn_workspace,
                       float* workspace, void* stream);
/* ---- the same backward in ONE launch (csrc/conv32_wino_bwd.hip): one 8-wave workgroup per CU, waves 0-3 the data gradient,
 * waves 4-7 the weight gradient on the g_z rows the first four keep in LDS — g_z never goes to HBM (x, g_a, z, next_z read, g_x
 * written: 5 tensor passes for the two launches' 7).  Arguments of as_conv32_wino_bwd without g_z; g_x equals the two-launch
 * result bit for bit, dW / db to rounding (another order of the sum over tiles).
 *   next_bn_workspace   as_conv32_wino_bwd_fused_parts() partials of [64] doubles;  workspace: as_conv32_wino_bwd_fused_workspace() floats
 * Replaces, per layer of EdgeAwareRefinement (reference models/stereo_net.py:10-18, 33-51, 97), autograd's conv2d backward +
 * BatchNorm backward + LeakyReLU backward + the residual add. */
int as_conv32_wino_bwd_fused_parts(void);
int64_t as_conv32_wino_bwd_fused_workspace(void);
int as_conv32_wino_bwd_fused(const float* x, const as_pcl* gin, const float* g_a, const float* z, const as_pcl* gout,
                             const as_conv_shape* s, const float* wino_wt, const float* scale, const float* shift,
                             const float* mean, const float* coef, float slope, const float* next_z,
                             const float* next_scale, const float* next_shift, const float* next_mean, float* g_x,
                             float* dW, float* db, int accumulate, float* next_bn_workspace, float* workspace, void* stream);

/* ---- the 5x5 stride-2 32->32 layers of the feature head (stereo_net.py:59-72): as_conv32_fwd and as_conv32_dgrad_s2(_packed)
 * run them on csrc/conv32_s2.hip (coalesced row staging through wave-private LDS) when the map fills the chip; 0 switches back to
 * the generic direct-load kernels (same bits: the parity tests compare them).  Returns the previous setting; any other argument
 * only queries */
int as_conv32_s2_enable(int on);
/* the same switch for the towers' FIRST layer, nn.Conv2d(3, 32, 5, stride=2, padding=2) (stereo_net.py:61-69): as_conv4_fwd runs
 * it on conv4_s2_fwd_kernel (persistent waves, rows staged through wave-private LDS) when the epilogue is plain and the map fills
 * the chip; bit-identical to conv4_fwd_kernel<25> */
int as_conv4_s2_enable(int on);

/* ---- the refinement's output layer with the last BasicBlock's activation on the way in (csrc/refine_out.hip;
 * stereo_net.py:44-51, 102, 116-121):  out = relu?(conv2d_out(a) + bias + add_src)  where
 *   scale != NULL:  a = lrelu(x * scale + shift) (+ skip) — x is the block's pre-activation, scale / shift its BatchNorm as an
 *                   affine; a is also written to a_out (the backward pass and nothing else reads it)
 *   scale == NULL:  a = x (already activated: inference); skip, shift, a_out must be NULL
 *   x, skip, a_out  PCL tensors of geometry g (as_refine_out_ok(g) == 1: 2-D, halo >= 1);  w [32][9], bias [1] or NULL
 *   add_src, out    dense [B][H][W] */
int as_refine_out_ok(const as_pcl* g);
int as_refine_out_fwd(const float* x, const float* skip, const float* scale, const float* shift, float slope, float* a_out,
                      const as_pcl* g, const float* w, const float* bias, const float* add_src, int relu, float* out,
                      void* stream);

/* ---- a3, second generation: one 3x3x3 stride-1 32->32 aggregation layer (stereo_net.py:21-30,155-161,185-186) or its data
 * gradient, walking down the disparity axis with a rolling window of planes in LDS (csrc/agg3d.hip).
 *   x, z, a_out      PCL tensors of geometry g (halo 1 in d, h, w; as_agg3d_ok(g) == 1)
 *   in_scale/shift   non-null: x is the PREVIOUS layer's raw convolution output and lrelu(x*in_scale + in_shift) — that
 *                    layer's BatchNorm + LeakyReLU(slope) — is applied on the fly; a_out (optional) receives the activated
 *                    tensor (what nn.Sequential would have materialised: the backward pass needs it)
 *   in_bn            alternative to in_scale/shift: the previous layer's BatchNorm is still in PARTIALS (what its own launch
 *                    wrote through stat_*): every workgroup merges them (as_bn_finalize's arithmetic) while its first planes
 *                    are in flight, workgroup 0 writes that layer's state and running statistics — no finalize launch
 *   epilogue         0: z = conv + bias, and (count, mean, M2) BatchNorm partials if stat_* are given (as_agg3d_parts(g)
 *                    partials); 1: z = lrelu((conv + bias)*ep_scale + ep_shift) (eval mode); 2: as 0 without moments */
typedef struct as_bn_merge {
  const float* stat_mean;   /* [nparts][32] */
  const float* stat_m2;     /* [nparts][32] */
  const float* stat_cnt;    /* [nparts] */
  const float* gamma;
  const float* beta;
  float* running_mean;      /* may be NULL (with running_var) */
  float* running_var;
  float* save_mean;         /* outputs, [32] each */
  float* save_invstd;
  float* scale;
  float* shift;
  int32_t nparts;
  float momentum;
  float eps;
} as_bn_merge;
int as_agg3d_ok(const as_pcl* g);
int as_agg3d_parts(const as_pcl* g);
int as_agg3d_fwd(const float* x, const as_pcl* g, const float* packed_w, const float* bias,
                 const float* in_scale, const float* in_shift, const as_bn_merge* in_bn, float* a_out,
                 float* z, int epilogue, const float* ep_scale, const float* ep_shift, float slope,
                 float* stat_mean, float* stat_m2, float* stat_cnt, void* stream);

/* ---- a4 + a5 + a8 in one launch: conv3d_alone (stereo_net.py:162,187) -> logits -> soft-argmax (:190-192,124-134),
 * arg-max index and feature-contrast score (utils/feature_contrast.py:12-23), csrc/agg_tail.hip.
 *   x                PCL volume of geometry g (halo 1; as_agg_tail_ok(g) == 1): the last aggregation layer's ACTIVATED output,
 *                    or — with in_scale/in_shift — its RAW convolution output, whose BatchNorm + LeakyReLU(slope) is then
 *                    applied on the fly (a_out, optional, receives the activated tensor for the backward pass); in_bn: as for
 *                    as_agg3d_fwd (the layer's BatchNorm still in partials, merged by every workgroup)
 *   w, bias          conv3d_alone.weight [1,32,3,3,3] (PyTorch order) and bias [1] (may be NULL)
 *   logits           dense [B,D,H,W];  pred, fcs: dense float [B,H,W];  argmax: int32 [B,H,W] (first maximum, as torch.argmax);
 *                    argmax and fcs may be NULL */
int as_agg_tail_ok(const as_pcl* g);
int as_agg_tail_fwd(const float* x, const as_pcl* g, const float* in_scale, const float* in_shift,
                    const as_bn_merge* in_bn, float* a_out,
                    const float* w, const float* bias, float slope, float* logits, float* pred,
                    int32_t* argmax, float* fcs, void* stream);

/* ---- a1, the 1/2^k-resolution trunk of FeatureExtractorNetwork in train mode (csrc/trunk.hip): six BasicBlocks
 * a_l = lrelu(BN_l(conv3x3_l(a_{l-1}))) + a_{l-1} (stereo_net.py:33-51, only conv1 is executed) and conv_alone (:85), called
 * once per image of a pair (adapt.py:72) — here ONE launch per layer and direction for both images: the batch dimension of
 * every PCL tensor holds `ngroups` statistics groups of B/ngroups images each (left images, then right images), BatchNorm
 * moments and gradients are taken per group, as the two separate calls of the reference take them.
 * Replaces, per BasicBlock and image: as_conv32_fwd + as_bn_finalize + as_bn_act_fwd forward and as_bn_act_bwd (three kernels) +
 * as_conv32_wgrad + as_conv32_fwd(dgrad) backward.
 *   geometry g       2-D PCL (D = 1, halo >= 1 in h and w), any H, W
 *   as_trunk_parts   workgroups per group of a launch = BatchNorm partials per group = weight-gradient slabs / ngroups
 * as_trunk_fwd: z = conv3x3(operand) + bias and, if stat_* are given, per-workgroup (count, mean, M2) moments of z
 *   [ngroups * parts] (+[32]).  bn_prev == NULL: the operand is `src`.  bn_prev != NULL: the operand is
 *   a_prev = lrelu(BN_prev(src)) + skip with BN_prev still in the partials its own launch wrote; every workgroup merges its
 *   group's partials (as_bn_finalize's arithmetic, fp64), the first workgroup of a group writes
 *   state[group] = {mean, invstd, scale, shift, unbiased variance}[32], and a_prev is written to a_out.
 * as_trunk_finish_fwd: features PCL -> NCHW, and running_mean/var of `nlayers` BatchNorm layers updated from `states`
 *   ([nlayers][ngroups][5][32]) group after group (feature_net(left), then feature_net(right)); running_mean / running_var are
 *   HOST arrays of nlayers device pointers.
 * as_trunk_bwd: backward of one layer.  z != NULL (a BasicBlock): g_a = dL/da_l; g_z = BatchNorm-backward(lrelu'(.) g_a) with
 *   the per-group sums merged from `sums` ([ngroups][nparts][64] doubles: what the launch above it left in sums_next);
 *   bn_grads [ngroups][2][32] receives the group's (g_gamma, g_beta); g_x = g_a + dgrad(g_z).  z == NULL (conv_alone): g_a is
 *   the gradient of the convolution's output itself, g_x = dgrad(g_a).  Both: dW [32,32,3,3] / db [32] set or accumulated
 *   (workspace: as_trunk_bwd_workspace floats, kept alive until as_wgrad_defer_flush inside a deferral region), and, if z_next is
 *   given, the stage-1 sums of the layer below (its pre-activation z_next, its state) into sums_next
 *   ([ngroups * parts][64] doubles).
 * as_trunk_finish_bwd: g_gamma[l] / g_beta[l] (HOST arrays of device pointers) set or accumulated from bn_grads
 *   ([nlayers][ngroups][2][32]), groups in call order. */
typedef struct as_trunk_bn {
  const float* stat_mean;   /* [ngroups][nparts][32] */
  const float* stat_m2;     /* [ngroups][nparts][32] */
  const float* stat_cnt;    /* [ngroups][nparts] */
  const float* gamma;
  const float* beta;
  float* state;             /* out: [ngroups][5][32] */
  int32_t nparts;           /* partials per group */
  float eps;
} as_trunk_bn;
int as_trunk_parts(const as_pcl* g, int ngroups);
int as_trunk_fwd(const float* src, const float* skip, const as_trunk_bn* bn_prev, float* a_out, const as_pcl* g, int ngroups,
                 const float* packed_w, const float* bias, float slope, float* z, float* stat_mean, float* stat_m2,
                 float* stat_cnt, void* stream);
int as_trunk_finish_fwd(const float* feats_pcl, const as_pcl* g, float* feats_nchw, const float* states, int nlayers,
                        int ngroups, float* const* running_mean, float* const* running_var, float momentum, void* stream);
 /* as_trunk_begin_bwd: the features' gradient(s), NCHW — one tensor [B,32,H,W] (g_b NULL, nA = B) or the left and right
 * halves of a pair pass separately ([nA,...] and [B-nA,...]) — into the interior of one PCL buffer of geometry g. */
int as_trunk_begin_bwd(const float* g_a, const float* g_b, int nA, const as_pcl* g, float* out_pcl, void* stream);
int64_t as_trunk_bwd_workspace(const as_pcl* g, int ngroups);
int as_trunk_bwd(const float* g_a, const float* z, const float* state, const double* sums, int nparts, const float* gamma,
                 float* bn_grads, const float* x, const float* packed_wt, float* g_x, const float* z_next,
                 const float* state_next, double* sums_next, const as_pcl* g, int ngroups, float slope, float* dW, float* db,
                 int accumulate, float* workspace, void* stream);
int as_trunk_finish_bwd(const float* bn_grads, int nlayers, int ngroups, float* const* g_gamma, float* const* g_beta,
                        int accumulate, void* stream);

/* Data gradient of nn.Conv2d(32,32,5,stride=2,padding=2) (FeatureExtractorNetwork.downsample[1..k-1],
 * stereo_net.py:61-69): four parity phases of a transposed convolution, each a gather over gz.
 * gz: PCL of the convolution's output extent (halo >= 1); gx: PCL of its input extent; w: PyTorch
 * [32][32][5][5]; workspace: as_conv32_dgrad_s2_workspace() floats. */
int64_t as_conv32_dgrad_s2_workspace(void);
int as_conv32_dgrad_s2(const float* gz, const as_pcl* ggz, const float* w, float* gx, const as_pcl* ggx,
                       float* workspace, void* stream);
/* The same in two halves: the phase-major packing of the weights (25 * 1024 floats; constant while the weights are) and the
 * convolution on packed weights — a caller that packs once per step (or through as_conv32_pack_weights_batch, AS_PACK_S2_DGRAD)
 * saves the packing launch of every call. */
int as_conv32_dgrad_s2_pack(const float* w, float* packed, void* stream);
int as_conv32_dgrad_s2_packed(const float* gz, const as_pcl* ggz, const float* packed, float* gx, const as_pcl* ggx,
                              void* stream);
int64_t as_conv32_wgrad_workspace(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s);
int as_conv32_wgrad(const float* x, const as_pcl* gin, const float* gz, const as_pcl* gout,
                    const as_conv_shape* s, float* dW, float* db, int accumulate, float* workspace, void* stream);
/* Host-side view of the work assignment of the LDS-staged 3-D weight gradient (csrc/conv3d_lds.hip): chunk, kd and extra tile
 * (-1: none) of workgroup `block` for a launch over ntiles tiles in nchunks chunks; returns 1 (working block) / 0 (padding
 * block).  Runs on the host — the function the kernel itself calls — so that a test without a GPU can check that every
 * tile is covered exactly three times for any (ntiles, nchunks). */
int as_conv3d_wgrad_lds_assignment(int ntiles, int nchunks, int block, int* chunk, int* kd, int* extra_tile);

/* ---- BatchNorm (train/eval) + LeakyReLU around the convolution --------------
 * nn.BatchNorm3d / nn.BatchNorm2d (eps, momentum as given) + nn.LeakyReLU(0.2)
 * (stereo_net.py:17,29,39,94,159).
 * as_bn_finalize: merges the conv's (count, mean, M2) partials (Chan, fp64),
 *   writes save_mean/save_invstd, scale = gamma*invstd, shift = beta - mean*scale,
 *   and updates running_mean / running_var (unbiased) with `momentum`.
 * as_bn_eval_affine: scale/shift from the running statistics (eval mode).
 * as_bn_act_fwd: a = lrelu(z*scale+shift) (+ residual) on the interior.
 * as_bn_act_bwd: given g_a, z: writes g_z (PCL interior), g_gamma, g_beta; train=1 uses
 *   batch statistics (full BatchNorm backward), train=0 treats mean/invstd as constants.
 *   workspace: as_bn_bwd_workspace() floats. If g_res != NULL it receives nothing (the
 *   skip connection's gradient is g_a itself and is handled by the caller). */
int as_bn_finalize(const float* stat_mean, const float* stat_m2, const float* stat_cnt, int nparts,
                   const float* gamma, const float* beta, float* running_mean, float* running_var,
                   float momentum, float eps, float* save_mean, float* save_invstd,
                   float* scale, float* shift, void* stream);
int as_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, float* save_mean, float* save_invstd,
                      float* scale, float* shift, void* stream);
/* as_bn_eval_affine for many layers in one launch; `jobs` is a DEVICE array, out = [mean, invstd, scale, shift][32]. */
typedef struct as_bn_affine_job {
  const float* gamma;
  const float* beta;
  const float* running_mean;
  const float* running_var;
  float* out;
} as_bn_affine_job;
int as_bn_eval_affine_batch(const as_bn_affine_job* jobs, int njobs, float eps, void* stream);
int as_bn_act_fwd(const float* z, const float* scale, const float* shift, float slope,
                  const float* residual, float* a, const as_pcl* g, void* stream);
int64_t as_bn_bwd_workspace(const as_pcl* g);
int as_bn_act_bwd(const float* g_a, const float* z, const float* scale, const float* shift,
                  const float* save_mean, const float* save_invstd, const float* gamma,
                  float slope, int train, float* g_z, float* g_gamma, float* g_beta, int accumulate,
                  float* workspace, const as_pcl* g, void* stream);
/* Data gradient of a convolution fused with stage 1 (the per-channel sums) of the BatchNorm backward that consumes
 * its output: z_out = conv(x) (+ residual) IS the g_a of the layer whose pre-activation is bn_z, so
 * sum g_y and sum g_y*(bn_z - mean), g_y = z_out * lrelu'(bn_z*scale + shift), are taken from the output tile in
 * registers and written to bn_workspace (an as_bn_bwd_workspace() buffer) as as_conv32_bnbwd_parts() slabs;
 * as_bn_act_bwd_given then runs stages 2 and 3 only.  as_conv32_bnbwd_parts() == 0: not available for the
 * configuration (use as_conv32_fwd + as_bn_act_bwd). */
int as_conv32_bnbwd_parts(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s);
int as_conv32_fwd_bnbwd(const float* x, const as_pcl* gin, const float* packed_w, float* z_out, const as_pcl* gout,
                        const as_conv_shape* s, const float* residual, const float* bn_z,
                        const float* bn_scale, const float* bn_shift, const float* bn_mean, float slope,
                        float* bn_workspace, void* stream);
/* Stage 3 (g_z = (g_a*lrelu'(z*scale+shift) - k1 - (z-mean)*k2)*k3) fused into the weight gradient of the same layer:
 * call as_bn_act_bwd / as_bn_act_bwd_given with g_z = NULL (stages 1-2 only; the coefficients k1,k2,k3 stay in the
 * workspace at float offset as_bn_bwd_coef_offset()), then as_conv32_wgrad_bnapply, which stages g_a and z rows, applies
 * stage 3 in LDS, accumulates dW/db from the result and writes g_z (PCL interior) for the data gradient that follows.
 * Available when as_conv32_wgrad_bnapply_ok() == 1 (launches that fill the chip). */
int64_t as_bn_bwd_coef_offset(void);
int as_conv32_wgrad_bnapply_ok(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s);
int as_conv32_wgrad_bnapply(const float* x, const as_pcl* gin, const float* g_a, const float* z, const as_pcl* gout,
                            const as_conv_shape* s, const float* scale, const float* shift, const float* mean,
                            const float* coef, float slope, float* g_z, float* dW, float* db, int accumulate,
                            float* workspace, void* stream);
int as_bn_act_bwd_given(const float* g_a, const float* z, const float* scale, const float* shift,
                        const float* save_mean, const float* save_invstd, const float* gamma,
                        float slope, int train, float* g_z, float* g_gamma, float* g_beta, int accumulate,
                        float* workspace, const as_pcl* g, int nparts, void* stream);

/* Cross-replica ("synchronised") BatchNorm in data-parallel adaptation (the reference's train-mode
 * BatchNorm, stereo_net.py:17,29, sees the whole batch).  Forward needs nothing new: the replicas exchange their
 * convolution partials (count, mean, M2) and every one calls as_bn_finalize on the concatenation.  Backward splits
 * as_bn_act_bwd at the point where the replicas must meet:
 *   as_bn_bwd_sums: stage 1 (skipped when nparts_given > 0: the slabs are already in the workspace, see
 *     as_conv32_fwd_bnbwd) + the fixed-order slab sum -> sums[65] doubles = [sum g_y][32], [sum g_y*(z-mean)][32],
 *     element count.  The caller all-reduces (sum) a COPY of it.
 *   as_bn_bwd_finalize_synced: g_gamma / g_beta from this replica's sums (the parameter-gradient all-reduce adds the
 *     replicas' later), stage-3 coefficients from the all-reduced sums and count, left in the workspace at
 *     as_bn_bwd_coef_offset().
 *   as_bn_bwd_apply: stage 3 alone (g_z on the PCL interior); as_conv32_wgrad_bnapply / as_conv4_wgrad_bnapply take the
 *     same coefficients instead when stage 3 rides on the weight gradient. */
int as_bn_bwd_sums(const float* g_a, const float* z, const float* scale, const float* shift,
                   const float* save_mean, float slope, float* workspace, const as_pcl* g,
                   int nparts_given, double* sums, void* stream);
int as_bn_bwd_finalize_synced(const double* local_sums, const double* global_sums, const float* save_invstd,
                              const float* gamma, float* g_gamma, float* g_beta, int accumulate,
                              float* workspace, void* stream);
int as_bn_bwd_apply(const float* g_a, const float* z, const float* scale, const float* shift,
                    const float* save_mean, float slope, float* g_z, const float* workspace,
                    const as_pcl* g, void* stream);

/* ---- a4 + a5 + a8: conv3d 32->1, soft-argmax, arg-max index, FCS ---------------
 * nn.Conv3d(32,1,3,padding=1) (stereo_net.py:162,187), F.softmax(dim=1) +
 * DisparityRegression (stereo_net.py:190-192,124-134), feature_contrast_mean
 * (utils/feature_contrast.py:12-23).
 * w: [32][27] (PyTorch [1][32][3][3][3]).  logits: [B][D][H][W] (the reference's
 * "cost_volume_{side}/{scale}" output).  pred/fcs: [B][H][W]; argmax: int32 [B][H][W]. */
int as_conv3d_out_fwd(const float* a, const as_pcl* g, const float* w, const float* bias,
                      float* logits, void* stream);
int64_t as_conv3d_out_bwd_workspace(const as_pcl* g);
int as_conv3d_out_bwd(const float* g_logits, const float* a, const as_pcl* g, const float* w,
                      float* g_a, float* g_w, float* g_bias, int accumulate, float* workspace, void* stream);
int as_softargmax_fwd(const float* logits, int B, int D, int H, int W,
                      float* pred, int32_t* argmax, float* fcs, void* stream);
/* g_logits[d] = p_d*(d - pred)*g_pred (+ g_logits_in[d] when not NULL). */
int as_softargmax_bwd(const float* logits, const float* g_pred, const float* g_logits_in,
                      int B, int D, int H, int W, float* g_logits, void* stream);

/* The backward of a5 + a4 in ONE launch (csrc/agg_tail_bwd.hip): the logits gradient p_d*(d - pred)*g_pred (+ g_logits_in) is
 * formed in LDS per (row, 4 disparity planes) and feeds the data gradient g_a and the weight / bias gradient of conv3d_alone from
 * one read of the activation a; it never reaches HBM.  g_pred [B][H][W], g_logits_in [B][D][H][W]: either may be NULL (zero).
 * a, g_a: PCL of geometry g; g_w [32][27], g_bias [1] (may be NULL): overwritten, or added to with accumulate != 0.
 * workspace: as_agg_tail_bwd_workspace(g) floats.  as_agg_tail_bwd_ok(g) == 0: use as_softargmax_bwd + as_conv3d_out_bwd. */
int as_agg_tail_bwd_ok(const as_pcl* g);
int64_t as_agg_tail_bwd_workspace(const as_pcl* g);
int as_agg_tail_bwd(const float* logits, const float* g_pred, const float* g_logits_in, const float* a, const as_pcl* g,
                    const float* w, float* g_a, float* g_w, float* g_bias, int accumulate, float* workspace, void* stream);

/* ---- generic 32 -> 1 convolution (3x3x3 or 3x3, 'same' padding, any dilation) ----------
 * The 3-D instance is a4 above; the 2-D instance is EdgeAwareRefinement.conv2d_out =
 * nn.Conv2d(32,1,3,padding=1) with the block's tail fused: out = relu(add_src + conv(a) + bias)
 * (stereo_net.py:102,121).  It also serves as the data-gradient of conv2d_feature w.r.t. its
 * disparity channel (a 32->1 convolution with mirrored taps).  w: [32][taps]; out/add_src/g_out:
 * dense [B][D][H][W].  bwd: g_a (PCL, may be NULL), g_w [32][taps] (may be NULL), g_bias [1] (may be NULL). */
int as_conv32to1_fwd(const float* a, const as_pcl* g, const as_conv_shape* s, const float* w,
                     const float* bias, const float* add_src, int relu, float* out, void* stream);
int64_t as_conv32to1_bwd_workspace(const as_pcl* g, const as_conv_shape* s);
int as_conv32to1_bwd(const float* g_out, const float* a, const as_pcl* g, const as_conv_shape* s,
                     const float* w, float* g_a, float* g_w, float* g_bias, int accumulate, float* workspace,
                     void* stream);
/* Data gradient of the 2-D 3x3 32->1 output layer (conv2d_out, stereo_net.py:102) that also produces stage 1 of the
 * BatchNorm backward consuming it — per-channel sums of g_a*lrelu'(z*scale+shift) and of that times (z - mean) — as
 * as_conv32to1_bnsums_parts(g) fp64 slabs in bn_workspace (as_bn_bwd_workspace floats), for as_bn_act_bwd_given.
 * Replaces as_conv32to1_bwd(g_a only) + the as_bn_bwd_sums pass over g_a and z. */
int as_conv32to1_bnsums_ok(const as_pcl* g, const as_conv_shape* s);
int as_conv32to1_bnsums_parts(const as_pcl* g);
int as_conv32to1_dgrad_bnsums(const float* g_out, const as_pcl* g, const as_conv_shape* s, const float* w, float* g_a,
                              const float* bn_z, const float* bn_scale, const float* bn_shift, const float* bn_mean,
                              float slope, float* bn_workspace, void* stream);

/* ---- thin-input convolution: Cin <= 4 -> 32 on "PCL4" ([B][H+2ph][W+2pw][4], zero halo) -----
 * EdgeAwareRefinement.conv2d_feature = nn.Conv2d(4,32,3,padding=1) over cat([disparity, rgb])
 * (stereo_net.py:89-94,116-118) and FeatureExtractorNetwork.downsample[0] = nn.Conv2d(3,32,5,
 * stride=2,padding=2) (:61-69).  as_pack_in4 builds the PCL4 image from an optional dense plane
 * ch0 [B,1,H,W] followed by the C channels of img [B,C,H,W] (this is the reference's torch.cat).
 * Epilogue arguments as for as_conv32_fwd.  wgrad: dW in PyTorch layout [32][Cin][kh][kw]. */
int64_t as_pcl4_numel(const as_pcl* g);
int as_pack_in4(const float* ch0, const float* img, int C, float* x4, const as_pcl* g, void* stream);
int as_conv4_pack_weights(const float* w, int Cin, float* packed, const as_conv_shape* s, void* stream);
/* Number of (mean, M2, count) partials as_conv4_fwd writes for this configuration (size of stat_*). */
int as_conv4_stat_parts(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s);
int as_conv4_fwd(const float* x4, const as_pcl* gin, const float* packed_w, const float* bias,
                 float* z, const as_pcl* gout, const as_conv_shape* s,
                 int epilogue, const float* ep_scale, const float* ep_shift, float slope,
                 float* stat_mean, float* stat_m2, float* stat_cnt, void* stream);
/* as_conv4_wgrad with stage 3 of the layer's BatchNorm backward applied on the fly (gradient operand = g_a, g_z written as
 * a by-product; coefficients from as_bn_act_bwd(g_z = NULL), see as_conv32_wgrad_bnapply).  3x3 stride-1 layers only. */
int as_conv4_wgrad_bnapply_ok(const as_pcl* gin, const as_pcl* gout, const as_conv_shape* s);
int as_conv4_wgrad_bnapply(const float* x4, const as_pcl* gin, const float* g_a, const float* z, const as_pcl* gout,
                           const as_conv_shape* s, int Cin, const float* scale, const float* shift, const float* mean,
                           const float* coef, float slope, float* g_z, float* dW, float* db, int accumulate,
                           float* workspace, void* stream);
/* as_conv4_wgrad_bnapply without the g_z write: what goes out are the nine per-tap projections
 * h[b][t][y][x] = sum_co g_z[b][y][x][co] * w_proj[t][co]  (t = 3*kh + kw, w_proj [9][32]) — all that the 3x3 32->1 data
 * gradient of conv2d_feature's disparity channel (stereo_net.py:116-118) needs; as_tap_gather then forms
 * out[b][y][x] = residual[b][y][x] + sum_t h[b][t][y + kh - 1][x + kw - 1] (zero outside the image; residual may be NULL). */
int as_conv4_wgrad_bnapply_proj(const float* x4, const as_pcl* gin, const float* g_a, const float* z, const as_pcl* gout,
                                const as_conv_shape* s, int Cin, const float* scale, const float* shift, const float* mean,
                                const float* coef, float slope, const float* w_proj, float* h, float* dW, float* db,
                                int accumulate, float* workspace, void* stream);
int as_tap_gather(const float* h, const float* residual, float* out, int B, int H, int W, void* stream);
int64_t as_conv4_wgrad_workspace(const as_pcl* gout, const as_conv_shape* s);
int as_conv4_wgrad(const float* x4, const as_pcl* gin, const float* gz, const as_pcl* gout,
                   const as_conv_shape* s, int Cin, float* dW, float* db, int accumulate, float* workspace,
                   void* stream);

/* ---- a6/a7 head: bilinear up-sampling, align_corners=False -------------------
 * F.interpolate(pred.unsqueeze(1), size=(H,W), mode="bilinear") * gain
 * (stereo_net.py:106-114, 201-202).  src: [B][h][w], dst: [B][1][H][W]. */
int as_upsample_bilinear_fwd(const float* src, int B, int h, int w, float* dst, int H, int W,
                             float gain, void* stream);
int as_upsample_bilinear_bwd(const float* g_dst, int B, int H, int W, float* g_src, int h, int w,
                             float gain, void* stream);

/* ---- a9: LinearWarping — models/linear_warping.py:18-57 -----------------------
 * grid_sample(bilinear, border, align_corners=False) at (x -/+ disp, y) after the
 * 2x/w-1 normalisation => samples at (x -/+ d - 0.5, y - 0.5).  img: NCHW [B,C,H,W];
 * disp: [B,1,H,W]; warped: [B,C,H,W]; mask: uint8 [B,1,H,W].  bwd: gradient w.r.t.
 * disp only (the image has no gradient in the adaptation loss). */
int as_warp_fwd(const float* img, const float* disp, int B, int C, int H, int W, int right_to_left,
                float* warped, uint8_t* mask, void* stream);
int as_warp_bwd(const float* g_warped, const float* img, const float* disp, int B, int C, int H, int W,
                int right_to_left, float* g_disp, void* stream);
/* LinearWarping.forward(mode="nearest") (models/linear_warping.py:57 forwards `mode` to F.grid_sample): the nearest tap of the
 * clipped sample position, ties to even; same validity mask.  No gradient w.r.t. the disparity exists in this mode. */
int as_warp_nearest_fwd(const float* img, const float* disp, int B, int C, int H, int W, int right_to_left,
                        float* warped, uint8_t* mask, void* stream);
/* The same with g_disp = (gradient through the warp) + add_src[i] (add_src may be NULL): the disparity receives its gradient
 * from the loss map directly and through the warped image (adapt.py:78-86) — the sum autograd would form in a launch of its own. */
int as_warp_bwd_add(const float* g_warped, const float* img, const float* disp, const float* add_src, int B, int C,
                    int H, int W, int right_to_left, float* g_disp, void* stream);

/* ---- a10: monodepth photometric loss — utils/loss_functions.py:41-138 ---------
 * total = 0.85*mean_c(SSIM dist) + 0.15*mean_c|I - I~| + sw*smooth(d/(mean d + 1e-7), I).
 * pred: [B,1,H,W]; img, warped: [B,3,H,W]; the four outputs: [B,1,H,W].
 * workspace: as_monodepth_workspace() floats.
 * bwd: g_* may be NULL (treated as zero).  Writes g_pred [B,1,H,W], g_warped [B,3,H,W]. */
int64_t as_monodepth_workspace(int B, int H, int W);
int as_monodepth_loss_fwd(const float* pred, const float* img, const float* warped, int B, int H, int W,
                          float smoothness_weight, float* total, float* l1, float* ssim, float* smooth,
                          float* workspace, void* stream);
int as_monodepth_loss_bwd(const float* g_total, const float* g_l1, const float* g_ssim, const float* g_smooth,
                          const float* pred, const float* img, const float* warped, int B, int H, int W,
                          float smoothness_weight, float* g_pred, float* g_warped,
                          float* workspace, void* stream);
/* Backward of loss = total[mask].mean() / total[mask].sum() (adapt.py:81-83) without a dense gradient map: every valid pixel
 * of `total` carries g_sum[0] + g_mean[0] / sum_count[1] (either pointer may be NULL; sum_count = as_masked_sum's out2 of the
 * forward pass), every other pixel 0; the l1 / ssim / smooth maps carry no gradient.  fwd_workspace (may be NULL): the workspace
 * as_monodepth_loss_fwd ran with on the same pred — its per-image mean disparity is reused instead of being summed again. */
int as_monodepth_loss_bwd_masked(const uint8_t* mask, const float* g_sum, const float* g_mean, const float* sum_count,
                                 const float* pred, const float* img, const float* warped, int B, int H, int W,
                                 float smoothness_weight, float* g_pred, float* g_warped, float* workspace,
                                 const float* fwd_workspace, void* stream);

/* ---- a9 + a10 + adapt.py:81-83 in one pass each way (csrc/photometric_rows.hip) -----------------------------------------
 * monodepth_single_loss (adapt.py:78-86): warp the right image with the predicted disparity (models/linear_warping.py:18-57),
 * monodepth loss map (utils/loss_functions.py:106-138), mean over the valid pixels — as row-walking strips: every input value is
 * loaded once, the 3x3 windows come from neighbouring lanes and from registers, no intermediate map touches HBM.  Bit for bit
 * what as_warp_fwd -> as_monodepth_loss_fwd -> as_masked_sum_mean and as_monodepth_loss_bwd_masked -> as_warp_bwd_add give.
 * pred [B,1,H,W]; left, right, warped [B,3,H,W]; mask uint8 [B,1,H,W]; out4 = {masked sum, count, mean, count}.
 * workspace: as_photometric_chain_workspace() floats, 16-byte aligned; the backward call takes the forward call's workspace as
 * fwd_workspace (per-image mean disparity) and its out4.  g_sum / g_mean: device scalars, either may be NULL. */
int64_t as_photometric_chain_workspace(int B, int H, int W);
int as_photometric_chain_fwd(const float* pred, const float* left, const float* right, int B, int H, int W,
                             float smoothness_weight, float* warped, uint8_t* mask, float* out4, float* workspace, void* stream);
int as_photometric_chain_bwd(const float* g_sum, const float* g_mean, const float* out4, const float* pred, const float* left,
                             const float* right, int B, int H, int W, float smoothness_weight, float* g_pred, float* workspace,
                             const float* fwd_workspace, void* stream);
/* The four loss maps from a given warped image, same strips (any output may be NULL); workspace as above. */
int as_monodepth_loss_rows_fwd(const float* pred, const float* img, const float* warped, int B, int H, int W,
                               float smoothness_weight, float* total, float* l1, float* ssim, float* smooth,
                               float* workspace, void* stream);

/* ---- loss[mask].mean() without a host sync — adapt.py:81-83 --------------------
 * out[0] = sum(v*m), out[1] = count(m); value = out[0]/out[1] is formed by the caller
 * on device.  workspace: as_masked_sum_workspace(n) floats. */
int64_t as_masked_sum_workspace(int64_t n);
int as_masked_sum(const float* v, const uint8_t* mask, int64_t n, float* out2, float* workspace, void* stream);
/* The same with out4 = [sum, count, sum / count, count]: the mean and a second copy of the count come out of the finalize
 * launch instead of two element-wise launches of the caller. */
int as_masked_sum_mean(const float* v, const uint8_t* mask, int64_t n, float* out4, float* workspace, void* stream);

/* ---- a13: khamis_robust_loss — utils/loss_functions.py:6-15 (ER modes, adapt.py:339-349) -------
 * out2[0] = sum_{gt>0}(sqrt((gt-pred)^2+4)/2 - 1) / max(count(gt>0),1), out2[1] = max(count,1).
 * as_khamis_bwd: g_pred = *g_loss * d out2[0] / d pred (g_loss: device scalar; out2 from the forward).
 * pred, gt: any shape with n elements.  workspace: as_khamis_workspace(n) floats, 8-byte aligned. */
int64_t as_khamis_workspace(int64_t n);
int as_khamis_fwd(const float* pred, const float* gt, int64_t n, float* out2, float* workspace, void* stream);
int as_khamis_bwd(const float* pred, const float* gt, const float* g_loss, const float* out2, int64_t n,
                  float* g_pred, void* stream);

/* ---- evaluation reductions — train.py:98-106 ------------------------------------
 * out6 = [sum |pred-gt| over gt>0, count(gt>0), count(gt>0 & |err|>2), >3, >4, >5]; EPE = out6[0]/out6[1],
 * D1_all_tpx = out6[t]/out6[1].  pred, gt: any shape with n elements.  workspace: as_eval_metrics_workspace(n). */
int64_t as_eval_metrics_workspace(int64_t n);
int as_eval_metrics(const float* pred, const float* gt, int64_t n, float* out6, float* workspace, void* stream);

/* ---- a12: clip_grad_norm_ + Adam, multi-tensor — adapt.py:208-210,391-393 ------
 * One flat fp32 arena holds params / grads / exp_avg / exp_avg_sq at equal offsets.
 * as_sumsq: out[0] = sum(g[0:n]^2) (deterministic two-stage).
 * as_adam_step: g is first multiplied by *grad_scale_dev (a device scalar, e.g. the clip
 * coefficient; NULL = 1), then the torch.optim.Adam update (no weight decay / amsgrad).  The step
 * count for the bias corrections is `step`, or *step_dev (a device float) when step_dev != NULL —
 * the latter lets a captured hipGraph replay successive steps. */
int64_t as_sumsq_workspace(int64_t n);
int as_sumsq(const float* g, int64_t n, float* out, float* workspace, void* stream);
/* coef[0] = min(max_norm / (sqrt(sumsq[0]) + 1e-6), 1): the scale clip_grad_norm_ applies (adapt.py:391). */
/* Glue of EdgeAwareRefinement's backward (stereo_net.py:116-121) as single launches:
 * as_relu_bwd: g_in[i] = out[i] > 0 ? g_out[i] : 0 (the final ReLU; n floats, 16-byte aligned);
 * as_mirror_taps_ch0: input channel 0 of a [32][Cin][3][3] weight with mirrored taps — by_tap [9][32] and by_channel [32][9]
 * (either may be NULL): the 32->1 data gradient of conv2d_feature towards the disparity channel. */
int as_relu_bwd(const float* g_out, const float* out, int64_t n, float* g_in, void* stream);
int as_mirror_taps_ch0(const float* w, int Cin, float* by_tap, float* by_channel, void* stream);
int as_clip_coef(const float* sumsq, float max_norm, float* coef, void* stream);
/* as_sumsq + as_clip_coef in the launches as_sumsq has anyway; step_counter (may be NULL) += 1: the optimizer's device-side
 * step count, read by as_adam_step under hipGraph replay. */
int as_sumsq_clip(const float* g, int64_t n, float max_norm, float* out, float* coef, float* step_counter,
                  float* workspace, void* stream);
int as_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                 const float* grad_scale_dev, float lr, float beta1, float beta2, float eps,
                 int step, const float* step_dev, void* stream);

/* ---- dataset layer on the device ----------------------------------------------------
 * What datasets/stereo_dataset.py:49-96 and utils/dataset_utils.py:26-57 do per sample on the host (ToTensor,
 * flip_stereo_pair, crop, disparity decoding), over the raw decoded file contents uploaded once:
 * as_decode_rgb8: src = uint8 [H0][W0][3] (PIL RGB) -> dst = float [3][H][W] = src/255 of the window (i0, j0, H, W);
 *   hflip = 1: the image was mirrored BEFORE cropping (the caller also swaps left and right, dataset_utils.py:19-23).
 * as_decode_plane: one-channel sample -> float [H][W]; dtype 0 = f32, 1 = u16, 2 = u8; vflip = 1 for bottom-up
 *   files (PFM, utils/io.py:73); value = v*scale (KITTI png 1/256, KITTI-raw npy 1/128, PFM 1) or scale/v
 *   (reciprocal = 1: Virtual KITTI depth in cm -> disparity, scale = baseline*focal/0.01).
 * The multi-scale pyramid (stereo_dataset.py:98-135) is as_upsample_bilinear_fwd (a general align_corners=False
 * resize) with gain 1 for colour and 1/2^s for disparity. */
int as_decode_rgb8(const uint8_t* src, int H0, int W0, int i0, int j0, int H, int W, int hflip, float* dst,
                   void* stream);
int as_decode_plane(const void* src, int dtype, int H0, int W0, int i0, int j0, int H, int W, int hflip,
                    int vflip, float scale, int reciprocal, float* dst, void* stream);

/* ---- measurement hook (bench.py roofline leg) -----------------------------------
 * When enabled, as_conv32_fwd and as_conv32_wgrad bracket their main kernel with HIP events on the
 * launch stream and account its algorithmic FLOPs (2 * voxels * 32 * 32 * taps).  Kernel ids:
 * 0 conv32_fwd_kernel<taps> (direct-load forward/dgrad), 1 conv32_wgrad_kernel<..> (direct-load wgrad),
 * 2 conv32_lds_kernel (LDS-staged 3x3 forward/dgrad), 3 conv32_wgrad_lds_kernel.
 * HBM-bound passes account algorithmic BYTES in the same slot: 4 as_bn_act_fwd (2 or 3 tensors x interior
 * bytes), 5 as_bn_act_bwd (its kernels together: 5 tensor passes, fewer when stages are fused elsewhere).
 * 6 conv32_lds_kernel<3,*> (the data gradient that also carries stage 1 of the next BatchNorm backward); id 2 then
 * counts the other flavours only.
 * as_prof_read synchronises on the recorded events.
 * Disabled by default; must stay disabled under hipGraph capture. */
int as_prof_enable(int on);
int as_prof_reset(void);
int as_prof_read(int kernel_id, int64_t* launches, double* total_ms, double* total_flops);

#ifdef __cplusplus
}
#endif
#endif  /* ADAPTIVE_STEREO_HIP_H_ */
