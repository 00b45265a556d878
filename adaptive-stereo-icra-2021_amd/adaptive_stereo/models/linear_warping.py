"""LinearWarping with the reference's call signature, computed by as_warp_fwd/bwd.

Reference: adaptive_stereo/models/linear_warping.py:6-57.  The reference builds a
sampling grid tensor per call (expand + clone + two normalisation passes + grid_sample);
here the grid never exists: one kernel computes the sample position from (x, y, disp)
in registers, including the reference's half-pixel quirk (it normalises with 2x/w - 1
but samples with align_corners=False, so the tap is at (x -/+ d - 0.5, y - 0.5)).
"""
import torch.nn as nn

from .. import _native as nat
from ..hip_ops import LinearWarpFn, LinearWarpNearestFn


class LinearWarping(nn.Module):
  def __init__(self, height, width, device=None):
    super().__init__()
    self._height = height
    self._width = width

  def forward(self, img, positive_disp, mode="bilinear", right_to_left=True):
    """img [B,C,H,W], positive_disp [B,1,H,W] -> (warped [B,C,H,W], valid_mask bool [B,1,H,W]).
    right_to_left=True synthesises the left view from a right image: L'(x,y) = R(x - d(x,y), y)."""
    nat.require_gpu(img, positive_disp)
    if mode not in ("bilinear", "nearest"):
      raise NotImplementedError("LinearWarping: mode %r (the reference forwards it to F.grid_sample; bilinear and nearest "
                                "are implemented, no caller uses another)" % (mode,))
    b, c, h, w = img.shape
    assert h == self._height
    assert w == self._width
    if mode == "nearest":
      return LinearWarpNearestFn.apply(img, positive_disp, bool(right_to_left))
    return LinearWarpFn.apply(img, positive_disp, bool(right_to_left))
