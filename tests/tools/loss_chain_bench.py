"""The loss tail at 375x1242 (warp + monodepth loss + masked mean, forward and backward): the row-walking chain
(csrc/photometric_rows.hip) against the one-thread-per-pixel functions stitched by autograd.  HIP-event times per call.
usage: python tests/tools/loss_chain_bench.py [B ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "adaptive-stereo-icra-2021_amd"))
import torch
from adaptive_stereo import hip_ops as ops

DEV = "cuda:0"
H, W = 375, 1242


def timed(fn, n=30, warm=5):
  for _ in range(warm):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n):
    fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / n


for B in [int(a) for a in sys.argv[1:]] or [1, 4, 16, 32]:
  g = torch.Generator().manual_seed(3)
  left = torch.rand(B, 3, H, W, generator=g).to(DEV)
  right = (left.roll(-17, -1) + 0.02 * torch.rand(B, 3, H, W, generator=g).to(DEV)).contiguous()
  pred0 = (torch.rand(B, 1, H, W, generator=g) * 30.0 + 2.0).to(DEV)
  px = B * H * W

  def chain_fwd():
    with torch.no_grad():
      return ops.MaskedPhotometricFn.apply(pred0, left, right, 1e-3)

  def chain_both():
    p = pred0.clone().requires_grad_(True)
    ops.MaskedPhotometricFn.apply(p, left, right, 1e-3)[0].backward()

  def old_fwd():
    with torch.no_grad():
      warped, mask = ops.LinearWarpFn.apply(right, pred0, True)
      total = ops.MonodepthLossFn.apply(pred0, left, warped, 1e-3)[0]
      return ops.masked_mean(total, mask)

  def old_both():
    p = pred0.clone().requires_grad_(True)
    warped, mask = ops.LinearWarpFn.apply(right, p, True)
    total = ops.MonodepthLossFn.apply(p, left, warped, 1e-3)[0]
    ops.masked_mean(total, mask).backward()

  tf, tb = timed(chain_fwd), timed(chain_both)
  of, ob = timed(old_fwd), timed(old_both)
  fb, bb = px * 41.0, px * 32.0      # algorithmic bytes: fwd reads pred, left, right, writes warped + mask; bwd reads pred, left, right, writes g_pred
  print("B=%2d  chain fwd %7.1f us (%5.0f GB/s algorithmic)  fwd+bwd %7.1f us (bwd alone ~%7.1f us, %5.0f GB/s) | pixel kernels fwd %7.1f  fwd+bwd %7.1f us"
        % (B, tf, fb / tf / 1e3, tb, tb - tf, bb / max(tb - tf, 1e-3) / 1e3, of, ob), flush=True)
