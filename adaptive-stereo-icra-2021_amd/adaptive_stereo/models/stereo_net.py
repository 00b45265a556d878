"""StereoNet / FeatureExtractorNetwork with the reference's public contract, computed on MI355X.

Drop-in surface (reference: adaptive_stereo/models/stereo_net.py):
  FeatureExtractorNetwork(k)(rgb[B,3,H,W]) -> [B,32,Hc,Wc]                       (:54-85)
  StereoNet(k, r, input_scale, maxdisp=192)(left_img, left_features, right_features,
            side, output_cost_volume=False) -> {"cost_volume_{side}/{s+k}",      (:137-207)
            "pred_disp_{side}/{s+k}", "pred_disp_{side}/{s}"}
  identical ``state_dict`` keys and shapes, including the never-executed
  ``BasicBlock.conv2`` tensors (:40) — checkpoints interchange with the reference.

The modules below are parameter containers: their nested ModuleLists exist to reproduce
the reference's key names.  ``forward`` never calls an nn.Conv3d/BatchNorm3d: the cost
volume, the 3-D aggregation, the soft-argmax, the up-sampling and their backward passes
run as hand-written HIP kernels behind ``libadaptive_stereo_hip.so`` (hip_ops.py), and so do
the edge-aware refinement (a7) and the whole feature extractor (a1).  No MIOpen/rocBLAS call is
issued anywhere on the path.
There is no CPU path: tensors must be on the GPU and the HIP library must be built.
"""
import torch
import torch.nn as nn

from .. import _native as nat
from .. import hip_ops

LEAKY_SLOPE = hip_ops.LEAKY_SLOPE
REFINE_DILATIONS = (1, 2, 4, 8, 1, 1)


class _ConvBN(nn.ModuleList):
  """[conv, batchnorm] pair; children are named "0" and "1" like the reference's nn.Sequential."""

  def __init__(self, dims, cin, cout, ksize, dilation=1):
    conv_t, bn_t = (nn.Conv3d, nn.BatchNorm3d) if dims == 3 else (nn.Conv2d, nn.BatchNorm2d)
    pad = dilation if dilation > 1 else 1
    super().__init__([conv_t(cin, cout, ksize, stride=1, padding=pad, dilation=dilation), bn_t(cout)])

  @property
  def conv(self):
    return self[0]

  @property
  def bn(self):
    return self[1]


class _ResidualBlock2d(nn.Module):
  """Parameters of x + LeakyReLU(BN(Conv3x3_dilated(x))).  ``conv2`` holds tensors the reference
  allocates but never uses (stereo_net.py:40, 44-51); they receive no gradient.  The arithmetic
  lives in hip_ops.block_forward / block_backward."""

  def __init__(self, dilation):
    super().__init__()
    self.dilation = dilation
    self.conv1 = nn.ModuleList([_ConvBN(2, 32, 32, 3, dilation)])
    self.conv2 = _ConvBN(2, 32, 32, 3, dilation)

  def live(self):
    return self.conv1[0]


def _block_params(blocks):
  params, buffers = [], []
  for cb in blocks:
    params += [cb.conv.weight, cb.conv.bias, cb.bn.weight, cb.bn.bias]
    buffers.append((cb.bn.running_mean, cb.bn.running_var))
  return params, buffers


def _count_batches(blocks):
  for cb in blocks:
    hip_ops.count_batch(cb.bn)


class FeatureExtractorNetwork(nn.Module):
  def __init__(self, k):
    super().__init__()
    self.k = k
    self.downsample = nn.ModuleList(
        [nn.Conv2d(3 if i == 0 else 32, 32, kernel_size=5, stride=2, padding=2) for i in range(k)])
    self.residual_blocks = nn.ModuleList([_ResidualBlock2d(1) for _ in range(6)])
    self.conv_alone = nn.Conv2d(32, 32, kernel_size=3, stride=1, padding=1)

  def forward(self, rgb_img):
    return self._run(rgb_img, 1)

  def forward_pair(self, left_img, right_img):
    """feature_net(left), feature_net(right) of the reference (adapt.py:72) as ONE pass over [left; right] with two
    statistics groups: in train mode BatchNorm moments, running-statistics updates (left first, then right) and
    gradients are taken per image batch, exactly as two calls take them; half the launches of a latency-bound chain.
    Returns (left_features, right_features)."""
    if left_img.shape != right_img.shape:
      raise RuntimeError("FeatureExtractorNetwork.forward_pair: left %s and right %s differ in shape" % (
          tuple(left_img.shape), tuple(right_img.shape)))
    n = left_img.shape[0]
    if self.training and (not hip_ops.trunk_enabled()):
      return self._run(left_img, 1), self._run(right_img, 1)
    if self.training:
      return self._run(hip_ops.adjacent_or_cat(left_img, right_img), 2)       # two outputs
    both = self._run(hip_ops.adjacent_or_cat(left_img, right_img), 1)
    return both[:n], both[n:]

  def _run(self, rgb_img, groups):
    nat.require_gpu(rgb_img)
    params = []
    for conv in self.downsample:          # no activation between the strided convs (:81-82)
      params += [conv.weight, conv.bias]
    live = [b.live() for b in self.residual_blocks]
    block_params, buffers = _block_params(live)
    params += block_params + [self.conv_alone.weight, self.conv_alone.bias]
    out = hip_ops.FeatureExtractorFn.apply(rgb_img, self.k, groups, self.training, torch.is_grad_enabled(), buffers,
                                           hip_ops.grad_sinks(params), *params)
    if self.training:
      for _ in range(groups):
        _count_batches(live)
    return out


class EdgeAwareRefinement(nn.Module):
  def __init__(self, in_channels):
    super().__init__()
    self.conv2d_feature = nn.ModuleList([_ConvBN(2, in_channels, 32, 3, 1)])
    self.residual_astrous_blocks = nn.ModuleList([_ResidualBlock2d(d) for d in REFINE_DILATIONS])
    self.conv2d_out = nn.Conv2d(32, 1, kernel_size=3, stride=1, padding=1)

  def forward(self, coarse_disparity, guidance_rgb):
    live = [self.conv2d_feature[0]] + [b.live() for b in self.residual_astrous_blocks]
    params, buffers = _block_params(live)
    params += [self.conv2d_out.weight, self.conv2d_out.bias]
    out = hip_ops.EdgeRefineFn.apply(coarse_disparity, guidance_rgb, self.training, torch.is_grad_enabled(), buffers,
                                     hip_ops.grad_sinks(params), *params)
    if self.training:
      _count_batches(live)
    return out


class DisparityRegression(nn.Module):
  """sum_d d * x[:, d] (stereo_net.py:124-134).  Exported only because callers import the name
  (evaluation/ood_analysis.py:13); it is NOT on the hot path — StereoNet.forward uses the fused
  soft-argmax kernel (as_softargmax_fwd), which never materialises the probabilities."""

  def __init__(self, maxdisp):
    super().__init__()
    self.maxdisp = maxdisp

  def forward(self, x):
    nat.require_gpu(x)
    idx = torch.arange(self.maxdisp, dtype=x.dtype, device=x.device).view(1, -1, 1, 1)
    return (x * idx).sum(dim=1)


class StereoNet(nn.Module):
  def __init__(self, k, r, input_scale, maxdisp=192):
    super().__init__()
    self.maxdisp = maxdisp
    self.k = k
    self.r = r
    self.input_scale = input_scale
    self.filter = nn.ModuleList([nn.ModuleList([_ConvBN(3, 32, 32, 3)]) for _ in range(4)])
    self.conv3d_alone = nn.Conv3d(32, 1, kernel_size=3, stride=1, padding=1)
    self.edge_aware_refinements = nn.ModuleList([EdgeAwareRefinement(4)])

  def coarse_max_disp(self):
    return (self.maxdisp + 1) // (2 ** (self.input_scale + self.k))

  def forward(self, left_img, left_features, right_features, side, output_cost_volume=False):
    nat.require_gpu(left_img, left_features, right_features)
    params, buffers = [], []
    for f in self.filter:
      cb = f[0]
      params += [cb.conv.weight, cb.conv.bias, cb.bn.weight, cb.bn.bias]
      buffers.append((cb.bn.running_mean, cb.bn.running_var))
    params += [self.conv3d_alone.weight, self.conv3d_alone.bias]

    logits, pred, argmax, fcs = hip_ops.CostAggregationFn.apply(
        left_features, right_features, self.coarse_max_disp(), self.training, torch.is_grad_enabled(), buffers,
        hip_ops.grad_sinks(params),
        *params)
    if self.training:
      for f in self.filter:
        hip_ops.count_batch(f[0].bn)

    # By-products of the fused soft-argmax kernel; feature_contrast_mean() picks the FCS up
    # from the logits tensor instead of sorting the volume again.
    logits._as_fcs = fcs
    logits._as_fcs_version = logits._version     # feature_contrast_mean uses the by-product only while the logits are unmodified
    logits._as_argmax = argmax

    coarse_scale = self.input_scale + self.k
    outputs = {}
    if output_cost_volume:
      outputs["cost_volume_{}/{}".format(side, coarse_scale)] = logits
    H, W = left_img.shape[-2:]
    outputs["pred_disp_{}/{}".format(side, coarse_scale)] = hip_ops.UpsampleBilinearFn.apply(
        pred, H, W, float(2 ** self.k))
    outputs["pred_disp_{}/{}".format(side, self.input_scale)] = self.edge_aware_refinements[0](pred, left_img)
    return outputs
