"""Loss functions with the reference's signatures (adaptive_stereo/utils/loss_functions.py).

monodepth_loss is one fused HIP stencil kernel forward and two backward (hip_ops.MonodepthLossFn)
instead of ~20 element-wise/pooling launches.  khamis_robust_loss is one masked HIP reduction (forward) and one element-wise HIP pass (backward),
used by the experience-replay modes (adapt.py:339-349) and the supervised multiscale loss.
"""
import torch

from .. import _native as nat
from ..hip_ops import MonodepthLossFn, KhamisLossFn


def monodepth_loss(pred_disp, true_img, warped_img, smoothness_weight=0.001):
  """-> (L_total, photo_l1, photo_ssim, L_smooth), each [B,1,H,W] (reference :106-138)."""
  nat.require_gpu(pred_disp, true_img, warped_img)
  return MonodepthLossFn.apply(pred_disp, true_img, warped_img, float(smoothness_weight))


def SSIM(x, y):
  """Per-channel SSIM distance clamp((1-SSIM)/2, 0, 1), [B,3,H,W] (reference :41-72).
  Not used by the adaptation step itself (monodepth_loss fuses it); provided for callers that
  want the per-channel map.  Only 3-channel images are supported by the kernel, and the channel
  mean is what it returns, so this wrapper evaluates it one channel at a time."""
  nat.require_gpu(x, y)
  b, c, h, w = x.shape
  zeros = torch.zeros(b, 1, h, w, dtype=torch.float32, device=x.device)
  outs = []
  for ch in range(c):
    xc = x[:, ch:ch + 1].expand(-1, 3, -1, -1)
    yc = y[:, ch:ch + 1].expand(-1, 3, -1, -1)
    outs.append(MonodepthLossFn.apply(zeros + 1.0, xc, yc, 0.0)[2])
  return torch.cat(outs, dim=1)


def khamis_robust_loss(pred_disp, gt_disp):
  """sum_{gt>0}(sqrt((gt-pred)^2+4)/2 - 1) / max(n,1) (reference :6-15): one HIP reduction pass forward, one
  element-wise pass backward (hip_ops.KhamisLossFn), no boolean-index sync."""
  nat.require_gpu(pred_disp, gt_disp)
  return KhamisLossFn.apply(pred_disp, gt_disp.detach())


def khamis_robust_loss_multiscale(inputs, outputs, scales=[0], gt_disp_scale=0):
  """Reference :18-38: equal-weight sum of the robust loss over the listed prediction scales."""
  losses = {"total_loss": 0}
  gt = inputs["gt_disp_l/{}".format(gt_disp_scale)]
  for scale in scales:
    this = khamis_robust_loss(outputs["pred_disp_l/{}".format(scale)], gt)
    losses["khamis_robust_loss/{}".format(scale)] = this
    losses["total_loss"] = losses["total_loss"] + this
  return losses


def monodepth_leftright_loss(left_img, right_img, outputs, warper, scale):
  """The reference's left-right consistency loss is dead code: it overwrites its ``outputs``
  argument and then raises KeyError (reference :154-157), and adapt.py:319 misspells the option that
  would reach it.  Kept as a name so ``from ... import monodepth_leftright_loss`` works."""
  raise KeyError("pred_disp_l/{}".format(scale))
