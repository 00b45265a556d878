"""ctypes binding of libadaptive_stereo_hip.so (C ABI: include/adaptive_stereo_hip.h).

The library is the product: there is no CPU or eager-PyTorch fallback for the
operators it provides.  If it is missing (not built, wrong path) every operator
raises, loudly, with the build command.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libadaptive_stereo_hip.so")
BUILD_HINT = "build it with: make -C %s" % os.path.join(os.path.dirname(_HERE), "csrc")

c_int, c_float, c_i64, c_vp = ctypes.c_int, ctypes.c_float, ctypes.c_int64, ctypes.c_void_p


class Pcl(ctypes.Structure):
  """as_pcl: geometry of a padded channel-last tensor [B][D+2pd][H+2ph][W+2pw][32]."""
  _fields_ = [(n, ctypes.c_int32) for n in ("B", "D", "H", "W", "pd", "ph", "pw")]

  def numel(self):
    return self.B * (self.D + 2 * self.pd) * (self.H + 2 * self.ph) * (self.W + 2 * self.pw) * 32

  def voxels(self):
    return self.B * self.D * self.H * self.W

  def key(self):
    return (self.B, self.D, self.H, self.W, self.pd, self.ph, self.pw)


class ConvShape(ctypes.Structure):
  """as_conv_shape."""
  _fields_ = [(n, ctypes.c_int32) for n in ("kd", "kh", "kw", "pad_d", "pad_h", "pad_w", "dil", "stride")]

  def taps(self):
    return self.kd * self.kh * self.kw


class BnMerge(ctypes.Structure):
  """as_bn_merge: a layer's BatchNorm still in per-workgroup partials + where its finalized state goes; the consumer
  kernel (as_agg3d_fwd / as_agg_tail_fwd) merges them itself."""
  _fields_ = [(n, c_vp) for n in ("stat_mean", "stat_m2", "stat_cnt", "gamma", "beta", "running_mean", "running_var",
                                 "save_mean", "save_invstd", "scale", "shift")] + \
             [("nparts", ctypes.c_int32), ("momentum", c_float), ("eps", c_float)]


class TrunkBn(ctypes.Structure):
  """as_trunk_bn: a BatchNorm of the small-map trunk still in per-workgroup partials, per statistics group."""
  _fields_ = [(n, c_vp) for n in ("stat_mean", "stat_m2", "stat_cnt", "gamma", "beta", "state")] + \
             [("nparts", ctypes.c_int32), ("eps", c_float)]


_P = ctypes.POINTER
_SIGNATURES = {
  # name: (restype, [argtypes])
  "as_last_error": (ctypes.c_char_p, []),
  "as_version": (c_int, []),
  "as_pcl_numel": (c_i64, [_P(Pcl)]),
  "as_cost_volume_fwd": (c_int, [c_vp, c_vp, c_vp, _P(Pcl), c_vp]),
  "as_cost_volume_bwd": (c_int, [c_vp, c_vp, c_vp, _P(Pcl), c_vp]),
  "as_conv32_pack_weights": (c_int, [c_vp, c_vp, _P(ConvShape), c_int, c_vp]),
  "as_conv32_pack_weights_batch": (c_int, [c_vp, c_int, c_int, c_vp]),
  "as_conv32_num_blocks": (c_int, [_P(Pcl)]),
  "as_conv32_stat_parts": (c_int, [_P(Pcl), _P(Pcl), _P(ConvShape)]),
  "as_conv32_fwd": (c_int, [c_vp, _P(Pcl), c_vp, c_vp, c_vp, _P(Pcl), _P(ConvShape), c_int, c_vp, c_vp, c_float,
                            c_vp, c_vp, c_vp, c_vp, c_vp]),
  "as_conv32_dgrad_s2_workspace": (c_i64, []),
  "as_conv32_dgrad_s2": (c_int, [c_vp, _P(Pcl), c_vp, c_vp, _P(Pcl), c_vp, c_vp]),
  "as_conv32_dgrad_s2_pack": (c_int, [c_vp, c_vp, c_vp]),
  "as_conv32_dgrad_s2_packed": (c_int, [c_vp, _P(Pcl), c_vp, c_vp, _P(Pcl), c_vp]),
  "as_conv32_wgrad_workspace": (c_i64, [_P(Pcl), _P(Pcl), _P(ConvShape)]),
  "as_conv32_wgrad": (c_int, [c_vp, _P(Pcl), c_vp, _P(Pcl), _P(ConvShape), c_vp, c_vp, c_int, c_vp, c_vp]),
  "as_bn_finalize": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_float, c_float, c_vp, c_vp, c_vp,
                             c_vp, c_vp]),
  "as_bn_eval_affine": (c_int, [c_vp, c_vp, c_vp, c_vp, c_float, c_vp, c_vp, c_vp, c_vp, c_vp]),
  "as_bn_eval_affine_batch": (c_int, [c_vp, c_int, c_float, c_vp]),
  "as_bn_act_fwd": (c_int, [c_vp, c_vp, c_vp, c_float, c_vp, c_vp, _P(Pcl), c_vp]),
  "as_bn_bwd_workspace": (c_i64, [_P(Pcl)]),
  "as_bn_act_bwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_float, c_int, c_vp, c_vp, c_vp, c_int, c_vp,
                            _P(Pcl), c_vp]),
  "as_bn_act_bwd_given": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_float, c_int, c_vp, c_vp, c_vp, c_int, c_vp,
                                  _P(Pcl), c_int, c_vp]),
  "as_bn_bwd_coef_offset": (c_i64, []),
  "as_bn_bwd_sums": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_float, c_vp, _P(Pcl), c_int, c_vp, c_vp]),
  "as_bn_bwd_finalize_synced": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
  "as_bn_bwd_apply": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_float, c_vp, c_vp, _P(Pcl), c_vp]),
  "as_conv32_wgrad_bnapply_ok": (c_int, [_P(Pcl), _P(Pcl), _P(ConvShape)]),
  "as_conv32_wgrad_bnapply": (c_int, [c_vp, _P(Pcl), c_vp, c_vp, _P(Pcl), _P(ConvShape), c_vp, c_vp, c_vp, c_vp, c_float,
                                      c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
  "as_conv32to1_bnsums_ok": (c_int, [_P(Pcl), _P(ConvShape)]),
  "as_conv32to1_bnsums_parts": (c_int, [_P(Pcl)]),
  "as_conv32to1_dgrad_bnsums": (c_int, [c_vp, _P(Pcl), _P(ConvShape), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_float, c_vp, c_vp]),
  "as_wgrad_defer": (c_int, [c_int]),
  "as_wgrad_defer_pending": (c_int, []),
  "as_wgrad_defer_flush": (c_int, [c_vp]),
  "as_conv32_act_ok": (c_int, [_P(Pcl), _P(Pcl), _P(ConvShape)]),
  "as_conv32_act_parts": (c_int, []),
  "as_conv32_act_fwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, _P(Pcl), c_vp, c_vp, c_float, c_vp, _P(Pcl), _P(ConvShape), c_vp, c_vp,
                                c_vp, c_vp]),
  "as_conv32_wino_ok": (c_int, [_P(Pcl), _P(Pcl), _P(ConvShape)]),
  "as_conv32_wino_parts": (c_int, []),
  "as_conv32_wino_pack_weights": (c_int, [c_vp, c_vp, c_int, c_vp]),
  "as_conv32_wino_fwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, _P(Pcl), c_vp, c_vp, c_float, c_vp, _P(Pcl), _P(ConvShape), c_vp, c_vp,
                                 c_vp, c_vp]),
  "as_conv32_wino_eval": (c_int, [c_vp, _P(Pcl), _P(ConvShape), c_vp, c_vp, c_vp, c_vp, c_float, c_int, c_vp, c_vp]),
  "as_conv32_wino_bwd_data": (c_int, [c_vp, c_vp, _P(Pcl), _P(ConvShape), c_vp, c_vp, c_vp, c_vp, c_vp, c_float, c_vp, c_vp, c_vp, c_vp,
                                      c_vp, c_vp, c_vp, c_vp]),
  "as_conv32_wino_bwd_filter": (c_int, [c_vp, c_vp, _P(Pcl), _P(ConvShape), c_vp, c_vp, c_int, c_vp, c_vp]),
  "as_conv32_wino_bwd_parts": (c_int, []),
  "as_conv32_wino_bwd_workspace": (c_i64, []),
  "as_conv32_wino_bwd_generation": (c_int, [c_int]),
  "as_conv32_wino_bwd": (c_int, [c_vp, _P(Pcl), c_vp, c_vp, _P(Pcl), _P(ConvShape), c_vp, c_vp, c_vp, c_vp, c_vp, c_float, c_vp,
                                 c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp]),
  "as_conv32_wino_bwd_fused_parts": (c_int, []),
  "as_conv32_wino_bwd_fused_workspace": (c_i64, []),
  "as_conv32_wino_bwd_fused": (c_int, [c_vp, _P(Pcl), c_vp, c_vp, _P(Pcl), _P(ConvShape), c_vp, c_vp, c_vp, c_vp, c_vp, c_float, c_vp,
                                       c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp]),
  "as_conv3d_wgrad_lds_assignment": (c_int, [c_int, c_int, c_int, c_vp, c_vp, c_vp]),
  "as_conv32_s2_enable": (c_int, [c_int]),
  "as_conv4_s2_enable": (c_int, [c_int]),
  "as_refine_out_ok": (c_int, [_P(Pcl)]),
  "as_refine_out_fwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_float, c_vp, _P(Pcl), c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
  "as_conv32_bwd_fused_ok": (c_int, [_P(Pcl), _P(Pcl), _P(ConvShape)]),
  "as_conv32_bwd_fused_parts": (c_int, []),
  "as_conv32_bwd_fused_workspace": (c_i64, []),
  "as_conv32_bwd_fused": (c_int, [c_vp, _P(Pcl), c_vp, c_vp, _P(Pcl), _P(ConvShape), c_vp, c_vp, c_vp, c_vp, c_vp, c_float, c_vp, c_vp,
                                  c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp]),
  "as_conv32_bnbwd_parts": (c_int, [_P(Pcl), _P(Pcl), _P(ConvShape)]),
  "as_conv32_fwd_bnbwd": (c_int, [c_vp, _P(Pcl), c_vp, c_vp, _P(Pcl), _P(ConvShape), c_vp, c_vp, c_vp, c_vp, c_vp, c_float,
                                  c_vp, c_vp]),
  "as_agg3d_ok": (c_int, [_P(Pcl)]),
  "as_agg3d_parts": (c_int, [_P(Pcl)]),
  "as_agg3d_fwd": (c_int, [c_vp, _P(Pcl), c_vp, c_vp, c_vp, c_vp, _P(BnMerge), c_vp, c_vp, c_int, c_vp, c_vp, c_float, c_vp, c_vp,
                           c_vp, c_vp]),
  "as_trunk_parts": (c_int, [_P(Pcl), c_int]),
  "as_trunk_fwd": (c_int, [c_vp, c_vp, _P(TrunkBn), c_vp, _P(Pcl), c_int, c_vp, c_vp, c_float, c_vp, c_vp, c_vp, c_vp, c_vp]),
  "as_trunk_finish_fwd": (c_int, [c_vp, _P(Pcl), c_vp, c_vp, c_int, c_int, _P(c_vp), _P(c_vp), c_float, c_vp]),
  "as_trunk_begin_bwd": (c_int, [c_vp, c_vp, c_int, _P(Pcl), c_vp, c_vp]),
  "as_trunk_bwd_workspace": (c_i64, [_P(Pcl), c_int]),
  "as_trunk_bwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, _P(Pcl), c_int, c_float,
                           c_vp, c_vp, c_int, c_vp, c_vp]),
  "as_trunk_finish_bwd": (c_int, [c_vp, c_int, c_int, _P(c_vp), _P(c_vp), c_int, c_vp]),
  "as_agg_tail_ok": (c_int, [_P(Pcl)]),
  "as_agg_tail_fwd": (c_int, [c_vp, _P(Pcl), c_vp, c_vp, _P(BnMerge), c_vp, c_vp, c_vp, c_float, c_vp, c_vp, c_vp, c_vp, c_vp]),
  "as_conv3d_out_fwd": (c_int, [c_vp, _P(Pcl), c_vp, c_vp, c_vp, c_vp]),
  "as_conv3d_out_bwd_workspace": (c_i64, [_P(Pcl)]),
  "as_conv3d_out_bwd": (c_int, [c_vp, c_vp, _P(Pcl), c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
  "as_conv32to1_fwd": (c_int, [c_vp, _P(Pcl), _P(ConvShape), c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
  "as_conv32to1_bwd_workspace": (c_i64, [_P(Pcl), _P(ConvShape)]),
  "as_conv32to1_bwd": (c_int, [c_vp, c_vp, _P(Pcl), _P(ConvShape), c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
  "as_pcl4_numel": (c_i64, [_P(Pcl)]),
  "as_pack_in4": (c_int, [c_vp, c_vp, c_int, c_vp, _P(Pcl), c_vp]),
  "as_conv4_pack_weights": (c_int, [c_vp, c_int, c_vp, _P(ConvShape), c_vp]),
  "as_conv4_wgrad_bnapply_ok": (c_int, [_P(Pcl), _P(Pcl), _P(ConvShape)]),
  "as_conv4_wgrad_bnapply_proj": (c_int, [c_vp, _P(Pcl), c_vp, c_vp, _P(Pcl), _P(ConvShape), c_int, c_vp, c_vp, c_vp, c_vp, c_float,
                                          c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
  "as_tap_gather": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
  "as_conv4_wgrad_bnapply": (c_int, [c_vp, _P(Pcl), c_vp, c_vp, _P(Pcl), _P(ConvShape), c_int, c_vp, c_vp, c_vp, c_vp, c_float,
                                     c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
  "as_conv4_stat_parts": (c_int, [_P(Pcl), _P(Pcl), _P(ConvShape)]),
  "as_conv4_fwd": (c_int, [c_vp, _P(Pcl), c_vp, c_vp, c_vp, _P(Pcl), _P(ConvShape), c_int, c_vp, c_vp, c_float,
                           c_vp, c_vp, c_vp, c_vp]),
  "as_conv4_wgrad_workspace": (c_i64, [_P(Pcl), _P(ConvShape)]),
  "as_conv4_wgrad": (c_int, [c_vp, _P(Pcl), c_vp, _P(Pcl), _P(ConvShape), c_int, c_vp, c_vp, c_int, c_vp, c_vp]),
  "as_softargmax_fwd": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
  "as_softargmax_bwd": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp]),
  "as_agg_tail_bwd_ok": (c_int, [_P(Pcl)]),
  "as_agg_tail_bwd_workspace": (c_i64, [_P(Pcl)]),
  "as_agg_tail_bwd": (c_int, [c_vp, c_vp, c_vp, c_vp, _P(Pcl), c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
  "as_upsample_bilinear_fwd": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_int, c_int, c_float, c_vp]),
  "as_upsample_bilinear_bwd": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_int, c_int, c_float, c_vp]),
  "as_warp_fwd": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
  "as_warp_nearest_fwd": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
  "as_warp_bwd": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp]),
  "as_warp_bwd_add": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp]),
  "as_monodepth_loss_bwd_masked": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_float, c_vp, c_vp, c_vp,
                                           c_vp, c_vp]),
  "as_monodepth_workspace": (c_i64, [c_int, c_int, c_int]),
  "as_monodepth_loss_fwd": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_float, c_vp, c_vp, c_vp, c_vp, c_vp,
                                    c_vp]),
  "as_monodepth_loss_bwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_float, c_vp,
                                    c_vp, c_vp, c_vp]),
  "as_photometric_chain_workspace": (c_i64, [c_int, c_int, c_int]),
  "as_photometric_chain_fwd": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_float, c_vp, c_vp, c_vp, c_vp, c_vp]),
  "as_photometric_chain_bwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_float, c_vp, c_vp, c_vp, c_vp]),
  "as_monodepth_loss_rows_fwd": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_float, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
  "as_masked_sum_workspace": (c_i64, [c_i64]),
  "as_masked_sum": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
  "as_masked_sum_mean": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
  "as_khamis_workspace": (c_i64, [c_i64]),
  "as_khamis_fwd": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
  "as_khamis_bwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp]),
  "as_eval_metrics_workspace": (c_i64, [c_i64]),
  "as_eval_metrics": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
  "as_relu_bwd": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp]),
  "as_mirror_taps_ch0": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp]),
  "as_sumsq_workspace": (c_i64, [c_i64]),
  "as_clip_coef": (c_int, [c_vp, c_float, c_vp, c_vp]),
  "as_sumsq_clip": (c_int, [c_vp, c_i64, c_float, c_vp, c_vp, c_vp, c_vp, c_vp]),
  "as_sumsq": (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp]),
  "as_adam_step": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_float, c_float, c_float, c_float, c_int, c_vp,
                           c_vp]),
  "as_decode_rgb8": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp]),
  "as_decode_plane": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_vp,
                              c_vp]),
  "as_prof_enable": (c_int, [c_int]),
  "as_prof_reset": (c_int, []),
  "as_prof_read": (c_int, [c_int, _P(c_i64), _P(ctypes.c_double), _P(ctypes.c_double)]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES.keys())

_lib = None


def load():
  """Returns the loaded library; raises RuntimeError if it cannot be loaded."""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise RuntimeError("adaptive_stereo: HIP library %s is missing; %s" % (LIB_PATH, BUILD_HINT))
    try:
      lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
      raise RuntimeError("adaptive_stereo: cannot load %s (%s); %s" % (LIB_PATH, e, BUILD_HINT))
    for name, (res, args) in _SIGNATURES.items():
      fn = getattr(lib, name)       # AttributeError here = header/library mismatch
      fn.restype = res
      fn.argtypes = args
    _lib = lib
  return _lib


def call(name, *args):
  """Invokes an int-returning entry point and raises on a non-zero status."""
  lib = load()
  rc = getattr(lib, name)(*args)
  if rc != 0:
    raise RuntimeError("%s failed (%d): %s" % (name, rc, lib.as_last_error().decode()))


def ptr(t):
  """Device pointer of a tensor (None -> NULL)."""
  if t is None:
    return None
  return c_vp(t.data_ptr())


def stream():
  return c_vp(torch.cuda.current_stream().cuda_stream)


def require_gpu(*tensors):
  for t in tensors:
    if t is None:
      continue
    if not t.is_cuda:
      raise RuntimeError("adaptive_stereo: tensors must live on the GPU (got %s); there is no CPU path" % t.device)
    if t.dtype not in (torch.float32, torch.uint8, torch.int32, torch.bool):
      raise RuntimeError("adaptive_stereo: unsupported dtype %s" % t.dtype)


def f32c(t):
  """fp32 contiguous view/copy of a GPU tensor."""
  if t is None:
    return None
  require_gpu(t)
  if t.dtype != torch.float32:
    t = t.float()
  return t.contiguous()
