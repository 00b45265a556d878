"""ORACLE — CPU restatement of the reference's StereoNet adaptation hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE. Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it. The product (``adaptive-stereo-icra-2021_amd/adaptive_stereo``) never
routes through this file; it fails loudly when the HIP library is missing.

What it is: a functional (state_dict in, tensors out), device-agnostic, pure
PyTorch restatement of the reference's arithmetic for SURVEY.md §8 rows a1-a13.
The reference computes this path with ATen CPU ops in fp32; so does this file,
so agreement with the reference is expected to be bit-exact or within a few
ulp. Differences in *form* from the reference: no ``nn.Module`` objects, no
hard-coded ``.cuda()``, the cost volume is built with one vectorised gather
instead of a Python loop over disparities, and BatchNorm state is explicit.

Parity pin: ``tests/golden/*.npz`` were produced by importing the reference's
own modules from /root/reference in the build container
(``tests/golden/make_golden.py``); ``tests/test_oracle_golden.py`` checks this
file against every one of them.

Reference citations are to files under /root/reference.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

LEAKY_SLOPE = 0.2      # stereo_net.py:39,94,159
BN_EPS = 1e-5          # nn.BatchNorm{2,3}d defaults, stereo_net.py:17,29
BN_MOMENTUM = 0.1
REFINE_DILATIONS = (1, 2, 4, 8, 1, 1)   # stereo_net.py:97


# --------------------------------------------------------------------------
# State handling
# --------------------------------------------------------------------------
def is_buffer_key(key: str) -> bool:
  return key.endswith(("running_mean", "running_var", "num_batches_tracked"))


def conv2_is_dead(key: str) -> bool:
  """BasicBlock.conv2 is constructed but never called (stereo_net.py:40 vs 44-51)."""
  return ".conv2." in key


def make_params(state_dict, requires_grad: bool):
  """Clones a state_dict into leaf tensors. Buffers never require grad."""
  out = OrderedDict()
  for k, v in state_dict.items():
    t = v.detach().clone()
    if requires_grad and not is_buffer_key(k) and t.is_floating_point():
      t.requires_grad_(True)
    out[k] = t
  return out


def _bn(p, prefix, x, train: bool):
  """BatchNorm with batch statistics in train mode (running stats updated in
  place, unbiased variance, momentum 0.1) and running statistics in eval mode."""
  if train:
    p[prefix + ".num_batches_tracked"] += 1
  return F.batch_norm(x, p[prefix + ".running_mean"], p[prefix + ".running_var"],
                      p[prefix + ".weight"], p[prefix + ".bias"],
                      training=train, momentum=BN_MOMENTUM, eps=BN_EPS)


def _lrelu(x):
  return F.leaky_relu(x, LEAKY_SLOPE)


# --------------------------------------------------------------------------
# a1 — feature extractor (stereo_net.py:54-85)
# --------------------------------------------------------------------------
def _residual_block_2d(p, prefix, x, dilation: int, train: bool):
  """x + LeakyReLU(BN(Conv3x3(x))); padding = dilation if dilation > 1 else 1
  (stereo_net.py:10-18, 33-51)."""
  pad = dilation if dilation > 1 else 1
  y = F.conv2d(x, p[prefix + ".conv1.0.0.weight"], p[prefix + ".conv1.0.0.bias"],
               stride=1, padding=pad, dilation=dilation)
  y = _lrelu(_bn(p, prefix + ".conv1.0.1", y, train))
  return x + y


def feature_extractor(p, rgb, k: int, train: bool):
  """k stride-2 5x5 convs with NO activation in between, six residual blocks,
  one plain 3x3 conv (stereo_net.py:79-85)."""
  x = rgb
  for i in range(k):
    x = F.conv2d(x, p["downsample.%d.weight" % i], p["downsample.%d.bias" % i], stride=2, padding=2)
  for i in range(6):
    x = _residual_block_2d(p, "residual_blocks.%d" % i, x, 1, train)
  return F.conv2d(x, p["conv_alone.weight"], p["conv_alone.bias"], stride=1, padding=1)


# --------------------------------------------------------------------------
# a2 — difference cost volume (stereo_net.py:173-184)
# --------------------------------------------------------------------------
def coarse_disparities(maxdisp: int, input_scale: int, k: int) -> int:
  return (maxdisp + 1) // (2 ** (input_scale + k))     # stereo_net.py:169


def cost_volume(fl, fr, num_disp: int):
  """cost[b,c,d,y,x] = L[b,c,y,x] - R[b,c,y,x-d] for x >= d, else 0."""
  b, c, h, w = fl.shape
  x = torch.arange(w, device=fl.device)
  d = torch.arange(num_disp, device=fl.device)
  src = x[None, :] - d[:, None]                         # [D, W]
  valid = (src >= 0)
  src = src.clamp(min=0)
  shifted = fr[:, :, :, src]                            # [B, C, H, D, W]
  shifted = shifted.permute(0, 1, 3, 2, 4)              # [B, C, D, H, W]
  diff = fl.unsqueeze(2) - shifted
  return (diff * valid[None, None, :, None, :].to(diff.dtype)).contiguous()


# --------------------------------------------------------------------------
# a3/a4 — 3D aggregation (stereo_net.py:185-187)
# --------------------------------------------------------------------------
def aggregate(p, volume, train: bool, taps=None):
  x = volume
  for i in range(4):
    x = F.conv3d(x, p["filter.%d.0.0.weight" % i], p["filter.%d.0.0.bias" % i], stride=1, padding=1)
    x = _lrelu(_bn(p, "filter.%d.0.1" % i, x, train))
    if taps is not None:
      taps["filter%d" % i] = x
  logits = F.conv3d(x, p["conv3d_alone.weight"], p["conv3d_alone.bias"], stride=1, padding=1)
  return logits.squeeze(1)                              # [B, D, H, W]


# --------------------------------------------------------------------------
# a5 — soft-argmax (stereo_net.py:190-192, 124-134). softmax(+cost): no negation.
# --------------------------------------------------------------------------
def soft_argmax(logits):
  prob = F.softmax(logits, dim=1)
  idx = torch.arange(logits.shape[1], dtype=logits.dtype, device=logits.device)
  return (prob * idx.view(1, -1, 1, 1)).sum(dim=1)      # [B, H, W]


# --------------------------------------------------------------------------
# a7 — edge-aware refinement (stereo_net.py:104-121)
# --------------------------------------------------------------------------
def refine(p, coarse, guidance_rgb, train: bool, prefix="edge_aware_refinements.0"):
  up = F.interpolate(coarse.unsqueeze(1), size=guidance_rgb.shape[-2:], mode="bilinear", align_corners=False)
  up = up * (guidance_rgb.shape[-1] / coarse.shape[-1])   # float ratio, e.g. 1242/78 (stereo_net.py:113)
  x = torch.cat([up, guidance_rgb], dim=1)                 # disparity is channel 0 (stereo_net.py:116-117)
  x = F.conv2d(x, p[prefix + ".conv2d_feature.0.0.weight"], p[prefix + ".conv2d_feature.0.0.bias"], padding=1)
  x = _lrelu(_bn(p, prefix + ".conv2d_feature.0.1", x, train))
  for i, dil in enumerate(REFINE_DILATIONS):
    x = _residual_block_2d(p, "%s.residual_astrous_blocks.%d" % (prefix, i), x, dil, train)
  res = F.conv2d(x, p[prefix + ".conv2d_out.weight"], p[prefix + ".conv2d_out.bias"], padding=1)
  return F.relu(up + res)


# --------------------------------------------------------------------------
# StereoNet.forward (stereo_net.py:168-207)
# --------------------------------------------------------------------------
def stereo_forward(p, left_img, fl, fr, k: int, input_scale: int, maxdisp: int, side: str,
                   train: bool, output_cost_volume: bool = False, taps=None):
  num_disp = coarse_disparities(maxdisp, input_scale, k)
  vol = cost_volume(fl, fr, num_disp)
  if taps is not None:
    taps["volume"] = vol
  logits = aggregate(p, vol, train, taps)
  pred = soft_argmax(logits)
  coarse_scale = input_scale + k
  out = OrderedDict()
  if output_cost_volume:
    out["cost_volume_%s/%d" % (side, coarse_scale)] = logits
  out["pred_disp_%s/%d" % (side, coarse_scale)] = (2 ** k) * F.interpolate(
      pred.unsqueeze(1), size=left_img.shape[-2:], mode="bilinear", align_corners=False)
  out["pred_disp_%s/%d" % (side, input_scale)] = refine(p, pred, left_img, train)
  if taps is not None:
    taps["pred"] = pred
  return out


# --------------------------------------------------------------------------
# a8 — feature-contrast score (utils/feature_contrast.py:12-23)
# --------------------------------------------------------------------------
def feature_contrast_mean(logits):
  with torch.no_grad():
    s = torch.sort(logits, dim=1, descending=True)[0]
    return s[:, 0] - s[:, 2:].mean(dim=1)


# --------------------------------------------------------------------------
# a9 — LinearWarping (models/linear_warping.py:18-57). Normalises with 2x/w - 1
# but samples with align_corners=False, i.e. at (x - d - 0.5, y - 0.5).
# --------------------------------------------------------------------------
def linear_warp(img, disp, right_to_left: bool = True, mode: str = "bilinear"):
  b, c, h, w = img.shape
  ys, xs = torch.meshgrid(torch.arange(h, device=img.device), torch.arange(w, device=img.device), indexing="ij")
  gx = xs.float().unsqueeze(0).expand(b, -1, -1)
  gy = ys.float().unsqueeze(0).expand(b, -1, -1)
  d = disp[:, 0]
  gx = gx - d if right_to_left else gx + d
  nx = 2 * gx / w - 1.0
  ny = 2 * gy / h - 1.0
  grid = torch.stack([nx, ny], dim=-1)
  valid = ((nx >= -1.0) & (nx <= 1.0) & (ny >= -1.0) & (ny <= 1.0)).unsqueeze(1)
  warped = F.grid_sample(img, grid, mode=mode, padding_mode="border", align_corners=False)   # (linear_warping.py:57)
  return warped, valid


# --------------------------------------------------------------------------
# a10 — photometric loss (utils/loss_functions.py:41-138)
# --------------------------------------------------------------------------
def ssim_distance(x, y):
  c1, c2 = 0.01 ** 2, 0.03 ** 2
  pool = lambda t: F.avg_pool2d(t, 3, stride=1, padding=1)   # zero pad, count_include_pad=True
  mu_x, mu_y = pool(x), pool(y)
  sig_x = pool(x ** 2) - mu_x ** 2
  sig_y = pool(y ** 2) - mu_y ** 2
  sig_xy = pool(x * y) - mu_x * mu_y
  n = (2 * mu_x * mu_y + c1) * (2 * sig_xy + c2)
  d = (mu_x ** 2 + mu_y ** 2 + c1) * (sig_x + sig_y + c2)
  return ((1 - n / d) / 2).clamp(min=0, max=1)


def edge_aware_smoothness(disp, img):
  gdx = (disp[:, :, :, :-1] - disp[:, :, :, 1:]).abs()
  gdy = (disp[:, :, :-1, :] - disp[:, :, 1:, :]).abs()
  gix = (img[:, :, :, :-1] - img[:, :, :, 1:]).abs().mean(1, keepdim=True)
  giy = (img[:, :, :-1, :] - img[:, :, 1:, :]).abs().mean(1, keepdim=True)
  gdx = F.pad(gdx * torch.exp(-gix), (0, 1))
  gdy = F.pad(gdy * torch.exp(-giy), (0, 0, 0, 1))
  return gdx + gdy


def monodepth_loss(pred_disp, true_img, warped_img, smoothness_weight: float = 1e-3):
  photo_ssim = ssim_distance(true_img, warped_img).mean(dim=1, keepdim=True)
  photo_l1 = (true_img - warped_img).abs().mean(dim=1, keepdim=True)
  photo = 0.85 * photo_ssim + 0.15 * photo_l1
  mean_disp = pred_disp.mean(2, True).mean(3, True)
  smooth = edge_aware_smoothness(pred_disp / (mean_disp + 1e-7), true_img)
  return photo + smoothness_weight * smooth, photo_l1, photo_ssim, smooth


def monodepth_single_loss(left, right, pred_disp):
  """adapt.py:78-86: warp right->left, loss map, mean over valid pixels of the batch."""
  warped, mask = linear_warp(right, pred_disp, True)
  total = monodepth_loss(pred_disp, left, warped, 1e-3)[0]
  return total[mask].mean(), warped, mask


# a13 — utils/loss_functions.py:6-15
def khamis_robust_loss(pred, gt):
  mask = gt > 0
  n = max(int(mask.sum()), 1)
  return (torch.sqrt((gt[mask] - pred[mask]) ** 2 + 4) / 2 - 1).sum() / n


# --------------------------------------------------------------------------
# a12 — clip (stereo_net only) + Adam (adapt.py:208-210, 391-393)
# --------------------------------------------------------------------------
def clip_grad_norm(grads, max_norm: float = 1.0):
  """torch.nn.utils.clip_grad_norm_: scale by max_norm / (total_norm + 1e-6), clamped to 1."""
  gs = [g for g in grads if g is not None]
  total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in gs]))
  coef = (max_norm / (total + 1e-6)).clamp(max=1.0)
  for g in gs:
    g.mul_(coef)
  return total


def adam_step(param, grad, state, lr, beta1=0.9, beta2=0.999, eps=1e-8):
  """torch.optim.Adam single-tensor update, no weight decay, no amsgrad."""
  state["step"] = state.get("step", 0) + 1
  m = state.setdefault("exp_avg", torch.zeros_like(param))
  v = state.setdefault("exp_avg_sq", torch.zeros_like(param))
  m.lerp_(grad, 1 - beta1)
  v.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
  bc1 = 1 - beta1 ** state["step"]
  bc2 = 1 - beta2 ** state["step"]
  denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
  param.addcdiv_(m, denom, value=-lr / bc1)


# --------------------------------------------------------------------------
# The harness the build reproduces: one NONSTOP adaptation step (adapt.py:304-396)
# --------------------------------------------------------------------------
def forward_only(feat_sd, stereo_sd, left, right, k, input_scale=0, maxdisp=192):
  """evaluate_model.py:52-60 / train.py:94-96 — eval mode, no grad, + FCS."""
  fp = make_params(feat_sd, False)
  sp = make_params(stereo_sd, False)
  with torch.no_grad():
    fl = feature_extractor(fp, left, k, False)
    fr = feature_extractor(fp, right, k, False)
    out = stereo_forward(sp, left, fl, fr, k, input_scale, maxdisp, "l", False, True)
    fcs = feature_contrast_mean(out["cost_volume_l/%d" % (input_scale + k)])
  return out, fcs


def adapt_step(feat_p, stereo_p, opt_state, left, right, k, input_scale=0, maxdisp=192,
               lr=5e-5, clip=True, taps=None):
  """feat_p / stereo_p are make_params(..., True) dicts and are updated in place."""
  for p in list(feat_p.values()) + list(stereo_p.values()):
    p.grad = None
  fl = feature_extractor(feat_p, left, k, True)
  fr = feature_extractor(feat_p, right, k, True)
  if taps is not None:
    fl.retain_grad(); fr.retain_grad()
    taps["fl"], taps["fr"] = fl, fr
  out = stereo_forward(stereo_p, left, fl, fr, k, input_scale, maxdisp, "l", True, True, taps)
  pred = out["pred_disp_l/%d" % input_scale]
  loss, warped, mask = monodepth_single_loss(left, right, pred)
  fcs = feature_contrast_mean(out["cost_volume_l/%d" % (input_scale + k)]).mean()
  loss.backward()
  if clip:
    clip_grad_norm([p.grad for kk, p in stereo_p.items() if p.requires_grad])
  with torch.no_grad():
    # Param-group order: stereo_net first, then feature_net (adapt.py:208-209).
    for name, group in (("stereo", stereo_p), ("feature", feat_p)):
      for kk, p in group.items():
        if p.requires_grad and p.grad is not None:
          adam_step(p, p.grad, opt_state.setdefault((name, kk), {}), lr)
  return {"loss": loss.detach(), "fcs": fcs, "outputs": out, "warped": warped, "mask": mask}
