"""torch.autograd.Function wrappers over the C ABI (include/adaptive_stereo_hip.h).

Each Function is a hand-written forward AND backward: autograd only stitches them
together.  Nothing here computes on the CPU; every call goes through
``_native.call`` on the current HIP stream with raw device pointers.
"""
import torch

from . import _native as nat
from ._native import Pcl, ConvShape, call, ptr, stream, f32c

LEAKY_SLOPE = 0.2
BN_EPS = 1e-5
BN_MOMENTUM = 0.1

CONV3D_333 = ConvShape(3, 3, 3, 1, 1, 1, 1, 1)


# ----------------------------------------------------------------------------------------
# PCL helpers (allocation + layout conversion; conversion is host plumbing used at the
# boundary with NCHW tensors and by tests)
# ----------------------------------------------------------------------------------------
def pcl_zeros(g: Pcl, device):
  return torch.zeros(g.numel(), dtype=torch.float32, device=device)


def pcl_view(buf, g: Pcl):
  return buf.view(g.B, g.D + 2 * g.pd, g.H + 2 * g.ph, g.W + 2 * g.pw, 32)


def pcl_interior(buf, g: Pcl):
  v = pcl_view(buf, g)
  return v[:, g.pd:g.pd + g.D, g.ph:g.ph + g.H, g.pw:g.pw + g.W, :]


def pcl_to_ncdhw(buf, g: Pcl):
  return pcl_interior(buf, g).permute(0, 4, 1, 2, 3).contiguous()


def ncdhw_to_pcl(t, g: Pcl):
  buf = pcl_zeros(g, t.device)
  pcl_interior(buf, g).copy_(t.permute(0, 2, 3, 4, 1))
  return buf


def _empty(n, device, dtype=torch.float32):
  return torch.empty(int(n), dtype=dtype, device=device)


# ----------------------------------------------------------------------------------------
# Thin call helpers
# ----------------------------------------------------------------------------------------
def pack_weights(w, shape: ConvShape, transpose_flip: bool):
  w = f32c(w)
  packed = _empty(shape.taps() * 1024, w.device)
  call("as_conv32_pack_weights", ptr(w), ptr(packed), shape, int(transpose_flip), stream())
  return packed


def conv32(x, gin: Pcl, packed_w, bias, gout: Pcl, shape: ConvShape, out=None, epilogue=0, scale=None,
           shift=None, residual=None, stats=None):
  """Returns the PCL output buffer. ``stats`` = (mean_partials, m2_partials) to fill."""
  z = out if out is not None else pcl_zeros(gout, x.device)
  sm, s2 = stats if stats is not None else (None, None)
  call("as_conv32_fwd", ptr(x), gin, ptr(packed_w), ptr(bias), ptr(z), gout, shape, int(epilogue),
       ptr(scale), ptr(shift), LEAKY_SLOPE, ptr(residual), ptr(sm), ptr(s2), stream())
  return z


def conv32_wgrad(x, gin: Pcl, gz, gout: Pcl, shape: ConvShape, want_bias=True):
  lib = nat.load()
  dev = x.device
  ws = _empty(lib.as_conv32_wgrad_workspace(gin, gout, shape), dev)
  taps = shape.taps()
  if shape.kd > 1:
    dW = _empty(32 * 32 * taps, dev).view(32, 32, shape.kd, shape.kh, shape.kw)
  else:
    dW = _empty(32 * 32 * taps, dev).view(32, 32, shape.kh, shape.kw)
  db = _empty(32, dev) if want_bias else None
  call("as_conv32_wgrad", ptr(x), gin, ptr(gz), gout, shape, ptr(dW), ptr(db), ptr(ws), stream())
  return dW, db


class BnState(object):
  """Per-layer BatchNorm scalars living on the device: scale/shift (the affine the
  activation pass applies) and mean/invstd (what backward needs)."""

  def __init__(self, device):
    buf = torch.empty(4, 32, dtype=torch.float32, device=device)
    self.mean, self.invstd, self.scale, self.shift = buf[0], buf[1], buf[2], buf[3]


def bn_train_stats(stats, count, gamma, beta, running_mean, running_var, nblocks):
  st = BnState(gamma.device)
  call("as_bn_finalize", ptr(stats[0]), ptr(stats[1]), int(nblocks), int(count), ptr(gamma), ptr(beta),
       ptr(running_mean), ptr(running_var), BN_MOMENTUM, BN_EPS, ptr(st.mean), ptr(st.invstd),
       ptr(st.scale), ptr(st.shift), stream())
  return st


def bn_eval_stats(gamma, beta, running_mean, running_var):
  st = BnState(gamma.device)
  call("as_bn_eval_affine", ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), BN_EPS,
       ptr(st.mean), ptr(st.invstd), ptr(st.scale), ptr(st.shift), stream())
  return st


def bn_act(z, st: BnState, g: Pcl, residual=None, out=None):
  a = out if out is not None else pcl_zeros(g, z.device)
  call("as_bn_act_fwd", ptr(z), ptr(st.scale), ptr(st.shift), LEAKY_SLOPE, ptr(residual), ptr(a), g, stream())
  return a


def bn_act_bwd(g_a, z, st: BnState, gamma, g: Pcl, train: bool):
  lib = nat.load()
  dev = z.device
  g_z = pcl_zeros(g, dev)
  g_gamma, g_beta = _empty(32, dev), _empty(32, dev)
  ws = _empty(lib.as_bn_bwd_workspace(g), dev)
  call("as_bn_act_bwd", ptr(g_a), ptr(z), ptr(st.scale), ptr(st.shift), ptr(st.mean), ptr(st.invstd), ptr(gamma),
       LEAKY_SLOPE, int(train), ptr(g_z), ptr(g_gamma), ptr(g_beta), ptr(ws), g, stream())
  return g_z, g_gamma, g_beta


# ----------------------------------------------------------------------------------------
# a2-a5 (+a8): cost volume -> 4x(conv3d+BN+LeakyReLU) -> conv3d 32->1 -> soft-argmax
# Reference: StereoNet.forward, adaptive_stereo/models/stereo_net.py:173-192
# ----------------------------------------------------------------------------------------
class CostAggregationFn(torch.autograd.Function):
  """inputs : fl, fr [B,32,H,W]; 4 x (conv weight, conv bias, bn weight, bn bias); out conv weight, bias
     outputs: logits [B,D,H,W], pred [B,H,W], argmax int32 [B,H,W], fcs [B,H,W]"""

  @staticmethod
  def forward(ctx, fl, fr, num_disp, train, bn_buffers, *params):
    assert len(params) == 18
    fl, fr = f32c(fl), f32c(fr)
    params = [f32c(p) for p in params]
    B, C, H, W = fl.shape
    if C != 32 or fr.shape != fl.shape:
      raise RuntimeError("CostAggregationFn: features must be [B,32,H,W] and equal in shape")
    dev = fl.device
    D = int(num_disp)
    g = Pcl(B, D, H, W, 1, 1, 1)
    need_bwd = any(ctx.needs_input_grad)
    count = g.voxels()
    nblocks = nat.load().as_conv32_num_blocks(g)

    vol = pcl_zeros(g, dev)
    call("as_cost_volume_fwd", ptr(fl), ptr(fr), ptr(vol), g, stream())

    x = vol
    xs, zs, sts = [vol], [], []
    for l in range(4):
      w, b, gamma, beta = params[4 * l:4 * l + 4]
      rm, rv = bn_buffers[l]
      wp = pack_weights(w, CONV3D_333, False)
      if train:
        stats = (_empty(nblocks * 32, dev), _empty(nblocks * 32, dev))
        z = conv32(x, g, wp, b, g, CONV3D_333, stats=stats)
        st = bn_train_stats(stats, count, gamma, beta, rm, rv, nblocks)
        a = bn_act(z, st, g)
      else:
        st = bn_eval_stats(gamma, beta, rm, rv)
        if need_bwd:
          z = conv32(x, g, wp, b, g, CONV3D_333)
          a = bn_act(z, st, g)
        else:
          z = None    # eval inference: BatchNorm + LeakyReLU fused into the conv epilogue
          a = conv32(x, g, wp, b, g, CONV3D_333, epilogue=1, scale=st.scale, shift=st.shift)
      zs.append(z); sts.append(st); xs.append(a)
      x = a

    w_out, b_out = params[16], params[17]
    logits = torch.empty(B, D, H, W, dtype=torch.float32, device=dev)
    call("as_conv3d_out_fwd", ptr(x), g, ptr(w_out), ptr(b_out), ptr(logits), stream())
    pred = torch.empty(B, H, W, dtype=torch.float32, device=dev)
    argmax = torch.empty(B, H, W, dtype=torch.int32, device=dev)
    fcs = torch.empty(B, H, W, dtype=torch.float32, device=dev)
    call("as_softargmax_fwd", ptr(logits), B, D, H, W, ptr(pred), ptr(argmax), ptr(fcs), stream())

    if need_bwd:
      ctx.g = g
      ctx.train = bool(train)
      ctx.xs, ctx.zs, ctx.sts = xs, zs, sts
      ctx.save_for_backward(logits, *params)
    ctx.mark_non_differentiable(argmax, fcs)
    return logits, pred, argmax, fcs

  @staticmethod
  def backward(ctx, g_logits_in, g_pred, _g_argmax, _g_fcs):
    logits, *params = ctx.saved_tensors
    g = ctx.g
    dev = logits.device
    B, D, H, W = logits.shape
    lib = nat.load()

    g_logits = torch.empty_like(logits)
    g_pred, g_logits_in = f32c(g_pred), f32c(g_logits_in)      # keep any contiguous copies alive past the launch
    call("as_softargmax_bwd", ptr(logits), ptr(g_pred), ptr(g_logits_in), B, D, H, W, ptr(g_logits), stream())

    grads = [None] * 18
    w_out = params[16]
    g_a = pcl_zeros(g, dev)
    g_wout = torch.empty_like(w_out)
    g_bout = _empty(1, dev)
    ws = _empty(lib.as_conv3d_out_bwd_workspace(g), dev)
    call("as_conv3d_out_bwd", ptr(g_logits), ptr(ctx.xs[4]), g, ptr(w_out), ptr(g_a), ptr(g_wout), ptr(g_bout),
         ptr(ws), stream())
    grads[16], grads[17] = g_wout, g_bout

    need_feat = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
    for l in range(3, -1, -1):
      w, b, gamma, beta = params[4 * l:4 * l + 4]
      g_z, g_gamma, g_beta = bn_act_bwd(g_a, ctx.zs[l], ctx.sts[l], gamma, g, ctx.train)
      dW, db = conv32_wgrad(ctx.xs[l], g, g_z, g, CONV3D_333)
      grads[4 * l:4 * l + 4] = [dW, db, g_gamma, g_beta]
      if l > 0 or need_feat:
        wp_t = pack_weights(w, CONV3D_333, True)
        g_a = conv32(g_z, g, wp_t, None, g, CONV3D_333)
    g_fl = g_fr = None
    if need_feat:
      g_fl = torch.empty(B, 32, H, W, dtype=torch.float32, device=dev)
      g_fr = torch.empty_like(g_fl)
      call("as_cost_volume_bwd", ptr(g_a), ptr(g_fl), ptr(g_fr), g, stream())
    ctx.xs = ctx.zs = ctx.sts = None
    return (g_fl, g_fr, None, None, None) + tuple(grads)


# ----------------------------------------------------------------------------------------
# a6: bilinear up-sampling (stereo_net.py:106-114, 201-202)
# ----------------------------------------------------------------------------------------
class UpsampleBilinearFn(torch.autograd.Function):
  """src [B,h,w] -> [B,1,H,W] * gain, align_corners=False."""

  @staticmethod
  def forward(ctx, src, H, W, gain):
    src = f32c(src)
    B, h, w = src.shape
    dst = torch.empty(B, 1, H, W, dtype=torch.float32, device=src.device)
    call("as_upsample_bilinear_fwd", ptr(src), B, h, w, ptr(dst), int(H), int(W), float(gain), stream())
    ctx.dims = (B, h, w, int(H), int(W), float(gain))
    return dst

  @staticmethod
  def backward(ctx, g_dst):
    B, h, w, H, W, gain = ctx.dims
    g_dst = f32c(g_dst)
    g_src = torch.empty(B, h, w, dtype=torch.float32, device=g_dst.device)
    call("as_upsample_bilinear_bwd", ptr(g_dst), B, H, W, ptr(g_src), h, w, gain, stream())
    return g_src, None, None, None


# ----------------------------------------------------------------------------------------
# a9: LinearWarping (models/linear_warping.py:18-57)
# ----------------------------------------------------------------------------------------
class LinearWarpFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, img, disp, right_to_left):
    img, disp = f32c(img), f32c(disp)
    B, C, H, W = img.shape
    if tuple(disp.shape) != (B, 1, H, W):
      raise RuntimeError("LinearWarpFn: disparity must be [B,1,H,W]")
    warped = torch.empty_like(img)
    mask = torch.empty(B, 1, H, W, dtype=torch.uint8, device=img.device)
    call("as_warp_fwd", ptr(img), ptr(disp), B, C, H, W, int(bool(right_to_left)), ptr(warped), ptr(mask), stream())
    ctx.save_for_backward(img, disp)
    ctx.r2l = int(bool(right_to_left))
    mask = mask.bool()
    ctx.mark_non_differentiable(mask)
    return warped, mask

  @staticmethod
  def backward(ctx, g_warped, _g_mask):
    img, disp = ctx.saved_tensors
    if ctx.needs_input_grad[0]:
      raise NotImplementedError("LinearWarpFn: gradient w.r.t. the image is not part of the adaptation path")
    B, C, H, W = img.shape
    g_disp = torch.empty_like(disp)
    g_warped = f32c(g_warped)
    call("as_warp_bwd", ptr(g_warped), ptr(img), ptr(disp), B, C, H, W, ctx.r2l, ptr(g_disp), stream())
    return None, g_disp, None


# ----------------------------------------------------------------------------------------
# a10: monodepth photometric loss (utils/loss_functions.py:106-138)
# ----------------------------------------------------------------------------------------
class MonodepthLossFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, pred, img, warped, smoothness_weight):
    pred, img, warped = f32c(pred), f32c(img), f32c(warped)
    B, C, H, W = img.shape
    if C != 3 or tuple(pred.shape) != (B, 1, H, W) or warped.shape != img.shape:
      raise RuntimeError("MonodepthLossFn: expected pred [B,1,H,W], img/warped [B,3,H,W]")
    dev = img.device
    outs = [torch.empty(B, 1, H, W, dtype=torch.float32, device=dev) for _ in range(4)]
    ws = _empty(nat.load().as_monodepth_workspace(B, H, W), dev)
    call("as_monodepth_loss_fwd", ptr(pred), ptr(img), ptr(warped), B, H, W, float(smoothness_weight),
         ptr(outs[0]), ptr(outs[1]), ptr(outs[2]), ptr(outs[3]), ptr(ws), stream())
    ctx.save_for_backward(pred, img, warped)
    ctx.sw = float(smoothness_weight)
    return tuple(outs)

  @staticmethod
  def backward(ctx, g_total, g_l1, g_ssim, g_smooth):
    pred, img, warped = ctx.saved_tensors
    if ctx.needs_input_grad[1]:
      raise NotImplementedError("MonodepthLossFn: gradient w.r.t. the true image is not part of the adaptation path")
    B, C, H, W = img.shape
    dev = img.device
    g_pred = torch.empty_like(pred) if ctx.needs_input_grad[0] else None
    g_warped = torch.empty_like(warped) if ctx.needs_input_grad[2] else None
    ws = _empty(nat.load().as_monodepth_workspace(B, H, W), dev)
    g_total, g_l1, g_ssim, g_smooth = f32c(g_total), f32c(g_l1), f32c(g_ssim), f32c(g_smooth)
    call("as_monodepth_loss_bwd", ptr(g_total), ptr(g_l1), ptr(g_ssim), ptr(g_smooth),
         ptr(pred), ptr(img), ptr(warped), B, H, W, ctx.sw, ptr(g_pred), ptr(g_warped), ptr(ws), stream())
    return g_pred, None, g_warped, None


# ----------------------------------------------------------------------------------------
# loss[mask].mean() without the boolean-index host sync (adapt.py:81-83)
# ----------------------------------------------------------------------------------------
class MaskedMeanFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, values, mask):
    values = f32c(values)
    m8 = mask.to(torch.uint8).contiguous()
    n = values.numel()
    out2 = _empty(2, values.device)
    ws = _empty(nat.load().as_masked_sum_workspace(n), values.device)
    call("as_masked_sum", ptr(values), ptr(m8), n, ptr(out2), ptr(ws), stream())
    ctx.save_for_backward(m8, out2)
    ctx.shape = values.shape
    return out2[0] / out2[1]

  @staticmethod
  def backward(ctx, g):
    m8, out2 = ctx.saved_tensors
    return (m8.to(torch.float32) * (g / out2[1])).view(ctx.shape), None


def masked_mean(values, mask):
  return MaskedMeanFn.apply(values, mask)
