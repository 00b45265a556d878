"""Kernel-level parity: every C-ABI entry point against the oracle / a plain PyTorch fp32
CPU evaluation of the same op, on seeded inputs with ragged (non-multiple-of-tile) extents.

All tests here need a real MI355X (``-m gpu``) and call the HIP library through the C ABI.
Tolerances are written next to each assertion; integer/index outputs must be exact.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from adaptive_stereo import _native as nat
from adaptive_stereo import hip_ops as ops
from adaptive_stereo.hip_ops import Pcl, ConvShape
from oracle import stereo_oracle as orc

DEV = "cuda:0"


def rnd(*shape, seed=0, scale=1.0):
  g = torch.Generator().manual_seed(seed)
  return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def close(got, exp, atol, rtol=0.0, what=""):
  got, exp = got.detach().cpu().double(), exp.detach().cpu().double()
  assert got.shape == exp.shape, "%s shape %s vs %s" % (what, tuple(got.shape), tuple(exp.shape))
  err = (got - exp).abs()
  tol = atol + rtol * exp.abs()
  assert bool((err <= tol).all()), "%s: max err %.3e, %d/%d over tol" % (what, float(err.max()), int((err > tol).sum()), err.numel())


# ----------------------------------------------------------------------------- a2
@pytest.mark.parametrize("B,D,H,W", [(1, 8, 5, 9), (2, 12, 7, 78), (1, 24, 3, 131), (1, 1, 1, 1), (1, 4, 2, 3)])
def test_cost_volume_fwd_is_bit_exact(B, D, H, W):
  fl, fr = rnd(B, 32, H, W, seed=1), rnd(B, 32, H, W, seed=2)
  g = Pcl(B, D, H, W, 1, 1, 1)
  vol = ops.pcl_zeros(g, DEV)
  fld, frd = fl.to(DEV), fr.to(DEV)     # named: a temporary's memory would be recycled before the launch
  nat.call("as_cost_volume_fwd", nat.ptr(fld), nat.ptr(frd), nat.ptr(vol), g, nat.stream())
  exp = orc.cost_volume(fl, fr, D)
  assert torch.equal(ops.pcl_to_ncdhw(vol, g).cpu(), exp)          # one fp32 subtract: exact
  # the halo must still be zero
  full = ops.pcl_view(vol, g).clone()
  ops.pcl_interior(full, g).zero_()
  assert float(full.abs().max()) == 0.0


@pytest.mark.parametrize("B,D,H,W", [(1, 8, 5, 9), (2, 12, 4, 78), (1, 24, 3, 70)])
def test_cost_volume_bwd(B, D, H, W):
  fl = rnd(B, 32, H, W, seed=1).requires_grad_(True)
  fr = rnd(B, 32, H, W, seed=2).requires_grad_(True)
  gv = rnd(B, 32, D, H, W, seed=3)
  orc.cost_volume(fl, fr, D).backward(gv)
  g = Pcl(B, D, H, W, 1, 1, 1)
  gbuf = ops.ncdhw_to_pcl(gv.to(DEV), g)
  gl, gr = torch.empty(B, 32, H, W, device=DEV), torch.empty(B, 32, H, W, device=DEV)
  nat.call("as_cost_volume_bwd", nat.ptr(gbuf), nat.ptr(gl), nat.ptr(gr), g, nat.stream())
  # sums of <= D terms: fp32 reassociation only
  close(gl, fl.grad, 1e-5, 1e-5, "gL"); close(gr, fr.grad, 1e-5, 1e-5, "gR")


# ----------------------------------------------------------------------------- a3
def _conv_ref(x, w, b, shape):
  if shape.kd > 1:
    return F.conv3d(x, w, b, stride=1, padding=(shape.pad_d, shape.pad_h, shape.pad_w), dilation=shape.dil)
  return F.conv2d(x[:, :, 0], w, b, stride=shape.stride, padding=(shape.pad_h, shape.pad_w), dilation=shape.dil).unsqueeze(2)


CONV_CASES = [
  # B, D, H, W, shape, halo
  (1, 8, 5, 9, ConvShape(3, 3, 3, 1, 1, 1, 1, 1), (1, 1, 1)),
  (2, 12, 6, 19, ConvShape(3, 3, 3, 1, 1, 1, 1, 1), (1, 1, 1)),
  (1, 3, 24, 78, ConvShape(3, 3, 3, 1, 1, 1, 1, 1), (1, 1, 1)),
  # LDS-staged 3-D instance (conv3d_lds.hip): 128-position tiles over the flattened padded plane; a ragged plane whose
  # last tile is shifted back; the k=4 and k=3 cost-volume geometries; first / last plane of the buffer (D = 1)
  (1, 5, 9, 40, ConvShape(3, 3, 3, 1, 1, 1, 1, 1), (1, 1, 1)),
  (2, 12, 24, 78, ConvShape(3, 3, 3, 1, 1, 1, 1, 1), (1, 1, 1)),
  (1, 4, 47, 156, ConvShape(3, 3, 3, 1, 1, 1, 1, 1), (1, 1, 1)),
  (2, 1, 4, 33, ConvShape(3, 3, 3, 1, 1, 1, 1, 1), (1, 1, 1)),
  (2, 1, 17, 37, ConvShape(1, 3, 3, 0, 1, 1, 1, 1), (0, 1, 1)),
  (1, 1, 21, 45, ConvShape(1, 3, 3, 0, 2, 2, 2, 1), (0, 8, 8)),
  (1, 1, 33, 40, ConvShape(1, 3, 3, 0, 8, 8, 8, 1), (0, 8, 8)),
  # LDS-staged instance: several 128-pixel tiles per row with a ragged tail, more tiles than workgroups
  (2, 1, 19, 300, ConvShape(1, 3, 3, 0, 1, 1, 1, 1), (0, 8, 8)),
  (1, 1, 23, 257, ConvShape(1, 3, 3, 0, 4, 4, 4, 1), (0, 8, 8)),
  (3, 1, 130, 131, ConvShape(1, 3, 3, 0, 8, 8, 8, 1), (0, 8, 8)),
  (1, 1, 375, 1242, ConvShape(1, 3, 3, 0, 2, 2, 2, 1), (0, 8, 8)),
  # >= 4096 row segments: the double-buffered one-workgroup-per-CU weight-gradient kernel, ragged last segment
  (1, 1, 421, 1250, ConvShape(1, 3, 3, 0, 4, 4, 4, 1), (0, 8, 8)),
]


@pytest.mark.parametrize("B,D,H,W,shape,halo", CONV_CASES)
def test_conv32_fwd_dgrad_wgrad(B, D, H, W, shape, halo):
  taps = shape.taps()
  wshape = (32, 32, 3, 3, 3) if shape.kd > 1 else (32, 32, 3, 3)
  x = rnd(B, 32, D, H, W, seed=1).requires_grad_(True)
  w = rnd(*wshape, seed=2, scale=1.0 / (32 * taps) ** 0.5).requires_grad_(True)
  b = rnd(32, seed=3, scale=0.1).requires_grad_(True)
  wr = w if shape.kd > 1 else w
  z_ref = _conv_ref(x, wr, b, shape)
  gz = rnd(*z_ref.shape, seed=4)
  z_ref.backward(gz)

  g = Pcl(B, D, H, W, *halo)
  xb = ops.ncdhw_to_pcl(x.detach().to(DEV), g)
  wd, bd = w.detach().to(DEV), b.detach().to(DEV)
  wp = ops.pack_weights(wd, shape, False)
  stats = ops.conv32_stat_parts(g, g, shape, DEV)
  zb = ops.conv32(xb, g, wp, bd, g, shape, stats=stats)
  # fp32 fma chain over K = 32*taps products of O(1/sqrt(K)) terms
  close(ops.pcl_to_ncdhw(zb, g), z_ref, 2e-5, 1e-5, "conv fwd")
  full = ops.pcl_view(zb, g).clone(); ops.pcl_interior(full, g).zero_()
  assert float(full.abs().max()) == 0.0, "conv wrote into the halo"

  # train-mode BatchNorm statistics from the epilogue partials
  gamma, beta = rnd(32, seed=5) * 0.5 + 1.0, rnd(32, seed=6) * 0.2
  rm, rv = rnd(32, seed=7) * 0.1, rnd(32, seed=8).abs() + 0.5
  rm_d, rv_d = rm.to(DEV).clone(), rv.to(DEV).clone()
  st = ops.bn_train_stats(stats, gamma.to(DEV), beta.to(DEV), rm_d, rv_d)
  zr = z_ref.detach()
  mean_ref = zr.mean(dim=(0, 2, 3, 4))
  var_ref = zr.var(dim=(0, 2, 3, 4), unbiased=False)
  close(st.mean, mean_ref, 2e-6, 1e-5, "bn mean")
  close(st.invstd, 1.0 / torch.sqrt(var_ref + 1e-5), 0, 2e-5, "bn invstd")
  rm_ref, rv_ref = rm.clone(), rv.clone()
  F.batch_norm(zr, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)
  close(rm_d, rm_ref, 2e-6, 1e-5, "running_mean"); close(rv_d, rv_ref, 2e-6, 2e-5, "running_var")

  # data gradient = same kernel on mirrored/transposed weights
  gzb = ops.ncdhw_to_pcl(gz.to(DEV), g)
  wpt = ops.pack_weights(wd, shape, True)
  gxb = ops.conv32(gzb, g, wpt, None, g, shape)
  close(ops.pcl_to_ncdhw(gxb, g), x.grad, 3e-5, 1e-5, "conv dgrad")

  # weight / bias gradient: fp32 sums over B*D*H*W voxels of O(1) terms (up to 465,750 of them): the
  # summation order differs from oneDNN's, so the bar is relative to the result's scale
  dW, db = ops.conv32_wgrad(xb, g, gzb, g, shape)
  for name, got, exp in (("conv wgrad", dW, w.grad), ("conv bias grad", db, b.grad)):
    scale = float(exp.abs().max())
    close(got, exp, 2e-4 * scale + 1e-5, 0, name)
    rel = float((got.cpu().double() - exp.double()).norm() / exp.double().norm())
    assert rel < 1e-4, "%s: relative L2 error %.2e" % (name, rel)


def test_conv32_fused_eval_epilogue_and_residual():
  B, D, H, W = 1, 1, 19, 23
  shape, halo = ConvShape(1, 3, 3, 0, 2, 2, 2, 1), (0, 8, 8)
  x, w, b = rnd(B, 32, D, H, W, seed=1), rnd(32, 32, 3, 3, seed=2, scale=0.06), rnd(32, seed=3, scale=0.1)
  sc, sh = rnd(32, seed=4) * 0.5 + 1.0, rnd(32, seed=5) * 0.3
  ref = x + F.leaky_relu(_conv_ref(x, w, b, shape) * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1), 0.2)
  g = Pcl(B, D, H, W, *halo)
  xb = ops.ncdhw_to_pcl(x.to(DEV), g)
  out = ops.conv32(xb, g, ops.pack_weights(w.to(DEV), shape, False), b.to(DEV), g, shape, epilogue=1,
                   scale=sc.to(DEV), shift=sh.to(DEV), residual=xb)
  close(ops.pcl_to_ncdhw(out, g), ref, 2e-5, 1e-5, "fused epilogue")


# ----------------------------------------------------------------------------- BN + LeakyReLU
@pytest.mark.parametrize("train", [True, False])
def test_bn_act_fwd_bwd(train):
  B, D, H, W = 2, 5, 7, 11
  g = Pcl(B, D, H, W, 1, 1, 1)
  z = rnd(B, 32, D, H, W, seed=1, scale=2.0).requires_grad_(True)
  gamma = (rnd(32, seed=2) * 0.5 + 1.0).requires_grad_(True)
  beta = (rnd(32, seed=3) * 0.2).requires_grad_(True)
  rm, rv = rnd(32, seed=4) * 0.1, rnd(32, seed=5).abs() + 0.5
  a_ref = F.leaky_relu(F.batch_norm(z, rm.clone(), rv.clone(), gamma, beta, train, 0.1, 1e-5), 0.2)
  ga = rnd(B, 32, D, H, W, seed=6)
  a_ref.backward(ga)

  zb = ops.ncdhw_to_pcl(z.detach().to(DEV), g)
  gd, bd = gamma.detach().to(DEV), beta.detach().to(DEV)
  if train:
    # statistics through the public path: an identity "convolution" is not available, so feed exact stats
    zr = z.detach()
    st = ops.BnState(DEV)
    st.mean.copy_(zr.mean(dim=(0, 2, 3, 4)))
    st.invstd.copy_(1.0 / torch.sqrt(zr.var(dim=(0, 2, 3, 4), unbiased=False) + 1e-5))
    st.scale.copy_(st.invstd * gd); st.shift.copy_(bd - st.mean * st.scale)
  else:
    st = ops.bn_eval_stats(gd, bd, rm.to(DEV), rv.to(DEV))
  ab = ops.bn_act(zb, st, g)
  close(ops.pcl_to_ncdhw(ab, g), a_ref, 1e-5, 1e-5, "bn+lrelu fwd")
  gab = ops.ncdhw_to_pcl(ga.to(DEV), g)
  g_z, g_gamma, g_beta = ops.bn_act_bwd(gab, zb, st, gd, g, train)
  close(ops.pcl_to_ncdhw(g_z, g), z.grad, 2e-5, 1e-4, "bn bwd g_z")
  close(g_gamma, gamma.grad, 2e-4, 1e-4, "bn bwd g_gamma")
  close(g_beta, beta.grad, 2e-4, 1e-4, "bn bwd g_beta")


# ----------------------------------------------------------------------------- a4 + a5 + a8
@pytest.mark.parametrize("B,D,H,W,gain", [(1, 8, 5, 9, 1.0), (2, 12, 6, 19, 50.0), (1, 24, 4, 33, 300.0)])
def test_out_conv_softargmax_fcs(B, D, H, W, gain):
  g = Pcl(B, D, H, W, 1, 1, 1)
  a = rnd(B, 32, D, H, W, seed=1).requires_grad_(True)
  w = (rnd(1, 32, 3, 3, 3, seed=2, scale=0.03) * gain).requires_grad_(True)
  b = (rnd(1, seed=3, scale=0.1) * gain).requires_grad_(True)
  logits_ref = F.conv3d(a, w, b, padding=1).squeeze(1)
  pred_ref = orc.soft_argmax(logits_ref)
  gp = rnd(B, H, W, seed=4)
  gl_in = rnd(B, D, H, W, seed=5, scale=0.1)
  (pred_ref * gp).sum().backward(retain_graph=True)
  ga_pred_only = a.grad.clone()

  ab = ops.ncdhw_to_pcl(a.detach().to(DEV), g)
  wd, bd = w.detach().to(DEV).contiguous(), b.detach().to(DEV)
  logits = torch.empty(B, D, H, W, device=DEV)
  nat.call("as_conv3d_out_fwd", nat.ptr(ab), g, nat.ptr(wd), nat.ptr(bd), nat.ptr(logits), nat.stream())
  close(logits, logits_ref, 3e-6 * max(1.0, gain), 1e-5, "logits")

  # soft-argmax on the ORACLE's logits so that index equality is a statement about this kernel alone
  lr = logits_ref.detach().to(DEV).contiguous()
  pred = torch.empty(B, H, W, device=DEV); am = torch.empty(B, H, W, dtype=torch.int32, device=DEV)
  fcs = torch.empty(B, H, W, device=DEV)
  nat.call("as_softargmax_fwd", nat.ptr(lr), B, D, H, W, nat.ptr(pred), nat.ptr(am), nat.ptr(fcs), nat.stream())
  assert torch.equal(am.cpu().long(), torch.argmax(logits_ref.detach(), dim=1)), "arg-max indices must be bit-exact"
  close(pred, pred_ref, 2e-5, 1e-5, "soft-argmax")
  close(fcs, orc.feature_contrast_mean(logits_ref.detach()), 1e-5 * max(1.0, gain), 1e-5, "fcs")

  g_logits = torch.empty(B, D, H, W, device=DEV)
  gpd, gl_ind = gp.to(DEV), gl_in.to(DEV)
  nat.call("as_softargmax_bwd", nat.ptr(lr), nat.ptr(gpd), nat.ptr(gl_ind), B, D, H, W,
           nat.ptr(g_logits), nat.stream())
  l2 = logits_ref.detach().clone().requires_grad_(True)
  ((orc.soft_argmax(l2) * gp).sum() + (l2 * gl_in).sum()).backward()
  close(g_logits, l2.grad, 2e-6, 1e-4, "soft-argmax bwd")

  # backward of the 32->1 convolution, driven with the reference's d(pred)/d(logits)
  a.grad = None; w.grad = None; b.grad = None
  gl = rnd(B, D, H, W, seed=7)
  F.conv3d(a, w, b, padding=1).squeeze(1).backward(gl)
  lib = nat.load()
  g_a = ops.pcl_zeros(g, DEV); g_w = torch.empty_like(wd); g_b = torch.empty(1, device=DEV)
  ws = torch.empty(lib.as_conv3d_out_bwd_workspace(g), device=DEV)
  gld = gl.to(DEV)
  nat.call("as_conv3d_out_bwd", nat.ptr(gld), nat.ptr(ab), g, nat.ptr(wd), nat.ptr(g_a), nat.ptr(g_w),
           nat.ptr(g_b), 0, nat.ptr(ws), nat.stream())
  close(ops.pcl_to_ncdhw(g_a, g), a.grad, 2e-6 * max(1.0, gain), 1e-5, "out-conv dgrad")
  n = g.voxels()
  close(g_w, w.grad, 2e-6 * n ** 0.5, 1e-5, "out-conv wgrad")
  close(g_b, b.grad, 2e-6 * n ** 0.5, 1e-5, "out-conv bias grad")
  assert ga_pred_only is not None


@pytest.mark.parametrize("B,D,H,W,with_gin", [(1, 8, 5, 9, True), (2, 12, 6, 19, False), (1, 24, 4, 33, True), (1, 5, 3, 7, False),
                                              (4, 12, 24, 78, False), (2, 24, 47, 156, True), (1, 3, 1, 1, True)])
def test_agg_tail_backward_in_one_launch(B, D, H, W, with_gin):
  """as_agg_tail_bwd (csrc/agg_tail_bwd.hip: the logits gradient formed in LDS, data + weight gradient of conv3d_alone from one
  read of the activation) against as_softargmax_bwd -> as_conv3d_out_bwd: g_a bit for bit (same taps, same order), g_w / g_bias
  within the rounding of a different summation order (both against an fp64 sum), overwrite and accumulate."""
  lib = nat.load()
  g = Pcl(B, D, H, W, 1, 1, 1)
  assert lib.as_agg_tail_bwd_ok(g) == 1
  logits = (rnd(B, D, H, W, seed=1) * 6.0).to(DEV)
  gp = rnd(B, H, W, seed=2).to(DEV)
  gin = (rnd(B, D, H, W, seed=3, scale=0.1)).to(DEV) if with_gin else None
  a = ops.ncdhw_to_pcl(rnd(B, 32, D, H, W, seed=4).to(DEV), g)
  w = (rnd(1, 32, 3, 3, 3, seed=5, scale=0.05)).to(DEV).contiguous()
  # reference: three launches
  g_logits = torch.empty(B, D, H, W, device=DEV)
  nat.call("as_softargmax_bwd", nat.ptr(logits), nat.ptr(gp), nat.ptr(gin), B, D, H, W, nat.ptr(g_logits), nat.stream())
  ga_ref = ops.pcl_zeros(g, DEV); gw_ref = torch.empty_like(w); gb_ref = torch.empty(1, device=DEV)
  ws = torch.empty(lib.as_conv3d_out_bwd_workspace(g), device=DEV)
  nat.call("as_conv3d_out_bwd", nat.ptr(g_logits), nat.ptr(a), g, nat.ptr(w), nat.ptr(ga_ref), nat.ptr(gw_ref), nat.ptr(gb_ref), 0,
           nat.ptr(ws), nat.stream())
  # one launch
  ga = ops.pcl_zeros(g, DEV); gw = torch.full_like(w, float("nan")); gb = torch.full((1,), float("nan"), device=DEV)
  ws2 = torch.empty(lib.as_agg_tail_bwd_workspace(g), device=DEV)
  nat.call("as_agg_tail_bwd", nat.ptr(logits), nat.ptr(gp), nat.ptr(gin), nat.ptr(a), g, nat.ptr(w), nat.ptr(ga), nat.ptr(gw),
           nat.ptr(gb), 0, nat.ptr(ws2), nat.stream())
  assert torch.equal(ga, ga_ref), float((ga - ga_ref).abs().max())      # (halo included: nothing written there)
  # fp64 weight gradient from the same logits gradient
  a64 = ops.pcl_to_ncdhw(a, g).double().cpu()
  gl64 = g_logits.double().cpu().unsqueeze(1)
  gw64 = torch.nn.grad.conv3d_weight(a64, (1, 32, 3, 3, 3), gl64, padding=1).to(DEV)
  gb64 = gl64.sum().to(DEV)
  scale = float(gw64.abs().max())
  e_new, e_old = float((gw.double() - gw64).abs().max()), float((gw_ref.double() - gw64).abs().max())
  assert e_new <= max(2.0 * e_old, 2e-6 * scale), (e_new, e_old, scale)
  assert abs(float(gb) - float(gb64)) <= max(2.0 * abs(float(gb_ref) - float(gb64)), 2e-6 * float(gl64.abs().sum()) ** 0.5 + 1e-6)
  # accumulate into sinks
  gw2, gb2 = gw.clone(), gb.clone()
  nat.call("as_agg_tail_bwd", nat.ptr(logits), nat.ptr(gp), nat.ptr(gin), nat.ptr(a), g, nat.ptr(w), nat.ptr(ga), nat.ptr(gw2),
           nat.ptr(gb2), 1, nat.ptr(ws2), nat.stream())
  close(gw2, 2.0 * gw, 1e-7 * max(scale, 1e-30), 1e-6, "tail bwd accumulate g_w")
  close(gb2, 2.0 * gb, 1e-6, 1e-6, "tail bwd accumulate g_bias")
  from conftest import parity_note
  parity_note("agg_tail_bwd[B%d D%d %dx%d]" % (B, D, H, W), g_a_bit_identical=True, g_w_max_err_vs_fp64=e_new,
              three_launches_g_w_max_err_vs_fp64=e_old)


def test_softargmax_ties_and_extremes():
  """First maximum wins (torch.argmax semantics); large logits must not overflow."""
  l = torch.zeros(1, 6, 1, 4)
  l[0, :, 0, 1] = torch.tensor([1.0, 5.0, 5.0, 0.0, 5.0, -1.0])      # three-way tie -> index 1
  l[0, :, 0, 2] = torch.tensor([-300.0, 200.0, -50.0, 199.0, 0.0, 10.0])
  l[0, :, 0, 3] = torch.tensor([3e4, -3e4, 0.0, 0.0, 0.0, 0.0])
  ld = l.to(DEV)
  pred = torch.empty(1, 1, 4, device=DEV); am = torch.empty(1, 1, 4, dtype=torch.int32, device=DEV)
  fcs = torch.empty(1, 1, 4, device=DEV)
  nat.call("as_softargmax_fwd", nat.ptr(ld), 1, 6, 1, 4, nat.ptr(pred), nat.ptr(am), nat.ptr(fcs), nat.stream())
  assert am.cpu().flatten().tolist() == torch.argmax(l, dim=1).flatten().tolist()
  close(pred, orc.soft_argmax(l), 1e-5, 1e-5, "soft-argmax extremes")
  close(fcs, orc.feature_contrast_mean(l), 1e-2, 1e-6, "fcs extremes")
  assert bool(torch.isfinite(pred).all())


# ----------------------------------------------------------------------------- a6
@pytest.mark.parametrize("B,h,w,H,W", [(1, 3, 4, 24, 32), (2, 24, 78, 375, 1242), (1, 10, 17, 75, 131), (1, 1, 1, 5, 7)])
def test_upsample_bilinear(B, h, w, H, W):
  src = rnd(B, h, w, seed=1).requires_grad_(True)
  gain = W / w
  ref = F.interpolate(src.unsqueeze(1), size=(H, W), mode="bilinear", align_corners=False) * gain
  gd = rnd(B, 1, H, W, seed=2)
  ref.backward(gd)
  s = src.detach().to(DEV).requires_grad_(True)
  out = ops.UpsampleBilinearFn.apply(s, H, W, gain)
  close(out, ref, 2e-6 * gain, 1e-5, "upsample fwd")
  out.backward(gd.to(DEV))
  close(s.grad, src.grad, 3e-4 * gain, 1e-4, "upsample bwd")


# ----------------------------------------------------------------------------- a9
@pytest.mark.parametrize("r2l", [True, False])
@pytest.mark.parametrize("B,H,W", [(1, 9, 17), (2, 33, 70)])
def test_linear_warp(B, H, W, r2l):
  img = rnd(B, 3, H, W, seed=1) * 0.5 + 0.5
  disp = (rnd(B, 1, H, W, seed=2) * 6.0 + 4.0)          # includes negatives and out-of-image targets
  disp[:, :, 0, :3] = 0.0
  d = disp.clone().requires_grad_(True)
  warped_ref, mask_ref = orc.linear_warp(img, d, r2l)
  gw = rnd(B, 3, H, W, seed=3)
  warped_ref.backward(gw)
  dd = disp.to(DEV).requires_grad_(True)
  warped, mask = ops.LinearWarpFn.apply(img.to(DEV), dd, r2l)
  assert torch.equal(mask.cpu(), mask_ref), "validity mask must be exact"
  close(warped, warped_ref, 2e-6, 1e-5, "warp fwd")
  warped.backward(gw.to(DEV))
  close(dd.grad, d.grad, 2e-5, 1e-4, "warp bwd")


@pytest.mark.parametrize("r2l", [True, False])
def test_linear_warp_nearest_mode(r2l):
  """LinearWarping(..., mode="nearest") (the reference forwards `mode` to F.grid_sample, models/linear_warping.py:57).  The
  reference's grid puts EVERY sample exactly half-way between two rows (y - 0.5: it normalises with 2y/h - 1 and samples with
  align_corners=False), so which of the two rows "nearest" picks is decided by the last bit of the un-normalisation — in ATen
  as here.  The test therefore accepts, per pixel, the oracle's value for the sample position itself or nudged by 1e-3 of a
  pixel up or down, and excludes columns within 1e-3 of a horizontal tie; the validity mask is exact; the disparity receives a
  zero gradient; other modes raise."""
  from adaptive_stereo.models.linear_warping import LinearWarping
  import torch.nn.functional as F
  B, H, W = 2, 33, 70
  img = rnd(B, 3, H, W, seed=1) * 0.5 + 0.5
  disp = (rnd(B, 1, H, W, seed=2) * 6.0 + 4.0)
  _, mask_ref = orc.linear_warp(img, disp, r2l, mode="nearest")
  ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
  gx = xs.float().unsqueeze(0) - disp[:, 0] if r2l else xs.float().unsqueeze(0) + disp[:, 0]
  gy = ys.float().unsqueeze(0).expand(B, -1, -1)
  refs = []
  for nudge in (0.0, 1e-3, -1e-3):
    grid = torch.stack([2 * gx / W - 1.0, 2 * (gy + nudge) / H - 1.0], dim=-1)
    refs.append(F.grid_sample(img, grid, mode="nearest", padding_mode="border", align_corners=False))
  dd = disp.to(DEV).requires_grad_(True)
  warped, mask = LinearWarping(H, W)(img.to(DEV), dd, mode="nearest", right_to_left=r2l)
  assert torch.equal(mask.cpu(), mask_ref)
  got = warped.detach().cpu()
  ok = torch.zeros(B, 1, H, W, dtype=torch.bool)
  for r in refs:
    ok |= (got == r).all(dim=1, keepdim=True)
  pos = gx.unsqueeze(1) - 0.5
  near_tie = ((pos - pos.floor()) - 0.5).abs() < 1e-3
  assert not bool((~ok & ~near_tie).any()), int((~ok & ~near_tie).sum())
  assert int((~ok).sum()) <= 0.01 * ok.numel()
  warped.sum().backward()
  assert float(dd.grad.abs().max()) == 0.0
  with pytest.raises(NotImplementedError):
    LinearWarping(H, W)(img.to(DEV), dd, mode="bicubic")


# ----------------------------------------------------------------------------- a10
@pytest.mark.parametrize("B,H,W", [(1, 8, 11), (2, 37, 53)])
def test_monodepth_loss_fwd_bwd(B, H, W):
  img = rnd(B, 3, H, W, seed=1) * 0.5 + 0.5
  warped = (img + rnd(B, 3, H, W, seed=2) * 0.2).clamp(0, 1)
  pred = rnd(B, 1, H, W, seed=3).abs() * 20 + 1
  p = pred.clone().requires_grad_(True); wv = warped.clone().requires_grad_(True)
  ref = orc.monodepth_loss(p, img, wv, 1e-3)
  gs = [rnd(B, 1, H, W, seed=10 + i) for i in range(4)]
  sum((r * g).sum() for r, g in zip(ref, gs)).backward()

  pd = pred.to(DEV).requires_grad_(True); wd = warped.to(DEV).requires_grad_(True)
  out = ops.MonodepthLossFn.apply(pd, img.to(DEV), wd, 1e-3)
  for name, o, r in zip(("total", "l1", "ssim", "smooth"), out, ref):
    close(o, r, 3e-6, 2e-5, "monodepth " + name)
  sum((o * g.to(DEV)).sum() for o, g in zip(out, gs)).backward()
  close(wd.grad, wv.grad, 1e-5 + 1e-4 * float(wv.grad.abs().max()), 1e-3, "monodepth g_warped")
  close(pd.grad, p.grad, 1e-6 + 1e-4 * float(p.grad.abs().max()), 1e-3, "monodepth g_pred")


def test_masked_mean_matches_boolean_index_mean():
  v = rnd(2, 1, 37, 53, seed=1).requires_grad_(True)
  m = rnd(2, 1, 37, 53, seed=2) > 0.3
  ref = v[m].mean(); ref.backward()
  vd = v.detach().to(DEV).requires_grad_(True)
  out = ops.masked_mean(vd, m.to(DEV))
  close(out, ref, 1e-6, 1e-5, "masked mean")
  out.backward()
  close(vd.grad, v.grad, 1e-9, 1e-5, "masked mean grad")


# ----------------------------------------------------------------------------- a12
def test_sumsq_and_adam_step():
  n = 313698
  p, g = rnd(n, seed=1), rnd(n, seed=2, scale=1e-2)
  lib = nat.load()
  gd = g.to(DEV)
  out = torch.empty(1, device=DEV); ws = torch.empty(lib.as_sumsq_workspace(n), device=DEV)
  nat.call("as_sumsq", nat.ptr(gd), n, nat.ptr(out), nat.ptr(ws), nat.stream())
  close(out, (g.double() ** 2).sum().float().view(1), 0, 1e-6, "sumsq")

  pr, state = p.clone(), {}
  pd = p.to(DEV); m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
  coef = torch.tensor([0.37], device=DEV)
  for step in (1, 2, 3):
    orc.adam_step(pr, g * 0.37, state, 5e-5)
    nat.call("as_adam_step", nat.ptr(pd), nat.ptr(gd), nat.ptr(m), nat.ptr(v), n, nat.ptr(coef), 5e-5, 0.9, 0.999,
             1e-8, step, None, nat.stream())
  close(pd, pr, 1e-7, 1e-6, "adam params")
  close(m, state["exp_avg"], 1e-9, 1e-5, "adam m"); close(v, state["exp_avg_sq"], 1e-12, 1e-5, "adam v")


# ----------------------------------------------------------------------------- thin-input conv (Cin <= 4)
@pytest.mark.parametrize("B,H,W", [(8, 375, 1242), (2, 540, 960), (3, 301, 515)])
def test_first_head_layer_on_staged_rows_equals_the_one_tile_kernel(B, H, W):
  """downsample[0] = Conv2d(3, 32, 5, stride=2, padding=2) (stereo_net.py:61-69) on conv4_s2_fwd_kernel (persistent waves,
  rows staged through wave-private LDS) and its weight gradient on conv4_s2_wgrad_kernel (input rows staged in LDS) against
  conv4_fwd_kernel<25> / conv4_wgrad_kernel<4> (as_conv4_s2_enable(0)): the forward bit for bit — taps and channels in the same
  order —, nothing written into the output's halo; the weight gradient (the same products in differently cut chunks) to 2e-5 of
  its largest entry; everything against torch on the CPU.  The bench workload (8 images of 375 x 1242),
  SceneFlow size, odd extents with a ragged last segment (W_out = 258 = 8 x 32 + 2)."""
  shape = ConvShape(1, 5, 5, 0, 2, 2, 1, 2)
  Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
  g4, gout = Pcl(B, 1, H, W, 0, 2, 2), Pcl(B, 1, Ho, Wo, 0, 2, 2)
  lib = nat.load()
  x = rnd(B, 3, H, W, seed=1); w = rnd(32, 3, 5, 5, seed=2, scale=0.115); b = rnd(32, seed=3, scale=0.1)
  x4 = torch.zeros(lib.as_pcl4_numel(g4), device=DEV)
  nat.call("as_pack_in4", None, nat.ptr(x.to(DEV)), 3, nat.ptr(x4), g4, nat.stream())
  wd, bd = w.to(DEV), b.to(DEV)
  wp = torch.empty(25 * 128, device=DEV)
  nat.call("as_conv4_pack_weights", nat.ptr(wd), 3, nat.ptr(wp), shape, nat.stream())
  gz = rnd(B, 32, Ho, Wo, seed=4)
  gzb = ops.ncdhw_to_pcl(gz.unsqueeze(2).to(DEV), gout)
  res = {}
  prev = lib.as_conv4_s2_enable(2)
  try:
    for on in (0, 1):
      lib.as_conv4_s2_enable(on)
      z = ops.pcl_zeros(gout, DEV)
      nat.call("as_conv4_fwd", nat.ptr(x4), g4, nat.ptr(wp), nat.ptr(bd), nat.ptr(z), gout, shape, 0, None, None, 0.2,
               None, None, None, nat.stream())
      ws = torch.empty(lib.as_conv4_wgrad_workspace(gout, shape), device=DEV)
      dW, db = torch.empty(32, 3, 5, 5, device=DEV), torch.empty(32, device=DEV)
      nat.call("as_conv4_wgrad", nat.ptr(x4), g4, nat.ptr(gzb), gout, shape, 3, nat.ptr(dW), nat.ptr(db), 0, nat.ptr(ws), nat.stream())
      torch.cuda.synchronize()
      res[on] = (z, dW, db)
  finally:
    lib.as_conv4_s2_enable(prev)
  assert bool(torch.equal(res[0][0], res[1][0])), "staged rows differ from the one-tile kernel in %d elements" % int((res[0][0] != res[1][0]).sum())
  full = ops.pcl_view(res[1][0], gout).clone(); ops.pcl_interior(full, gout).zero_()
  assert float(full.abs().max()) == 0.0, "the staged-row kernel wrote into the halo"
  xr = x.clone(); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
  z_ref = F.conv2d(xr, wr, br, stride=2, padding=2)
  z_ref.backward(gz)
  close(ops.pcl_to_ncdhw(res[1][0], gout)[:, :, 0], z_ref, 2e-5, 1e-5, "first head layer")
  scale = float(wr.grad.abs().max())
  # (the two weight-gradient kernels cut the launch into different chunks: the same products, summed in different groups)
  close(res[1][1] / scale, res[0][1] / scale, 2e-5, 0.0, "first head layer: weight gradient, staged rows against the gather kernel")
  close(res[1][2] / scale, res[0][2] / scale, 2e-5, 0.0, "first head layer: bias gradient, staged rows against the gather kernel")
  close(res[1][1] / scale, wr.grad / scale, 5e-5, 0.0, "first head layer: weight gradient")
  close(res[1][2] / scale, br.grad / scale, 5e-5, 0.0, "first head layer: bias gradient")


@pytest.mark.parametrize("B,H,W,Cin,k,stride,pad", [(1, 9, 13, 4, 3, 1, 1), (2, 37, 53, 4, 3, 1, 1),
                                                    (2, 20, 131, 4, 3, 1, 1), (1, 7, 32, 4, 3, 1, 1),
                                                    (1, 21, 30, 3, 5, 2, 2), (2, 75, 131, 3, 5, 2, 2)])
def test_conv4_fwd_and_wgrad(B, H, W, Cin, k, stride, pad):
  x = rnd(B, Cin, H, W, seed=1).requires_grad_(True)
  w = rnd(32, Cin, k, k, seed=2, scale=1.0 / (Cin * k * k) ** 0.5).requires_grad_(True)
  b = rnd(32, seed=3, scale=0.1).requires_grad_(True)
  z_ref = F.conv2d(x, w, b, stride=stride, padding=pad)
  gz = rnd(*z_ref.shape, seed=4)
  z_ref.backward(gz)
  Ho, Wo = z_ref.shape[-2:]
  shape = ConvShape(1, k, k, 0, pad, pad, 1, stride)
  g4 = Pcl(B, 1, H, W, 0, pad, pad)
  gout = Pcl(B, 1, Ho, Wo, 0, 2, 2)
  lib = nat.load()
  x4 = torch.zeros(lib.as_pcl4_numel(g4), device=DEV)
  xd = x.detach().to(DEV)
  if Cin == 4:
    ch0, img = xd[:, :1].contiguous(), xd[:, 1:].contiguous()
    nat.call("as_pack_in4", nat.ptr(ch0), nat.ptr(img), 3, nat.ptr(x4), g4, nat.stream())
  else:
    nat.call("as_pack_in4", None, nat.ptr(xd), 3, nat.ptr(x4), g4, nat.stream())
  wd, bd = w.detach().to(DEV), b.detach().to(DEV)
  wp = torch.empty(k * k * 128, device=DEV)
  nat.call("as_conv4_pack_weights", nat.ptr(wd), Cin, nat.ptr(wp), shape, nat.stream())
  z = ops.pcl_zeros(gout, DEV)
  nat.call("as_conv4_fwd", nat.ptr(x4), g4, nat.ptr(wp), nat.ptr(bd), nat.ptr(z), gout, shape, 0, None, None, 0.2,
           None, None, None, nat.stream())
  close(ops.pcl_to_ncdhw(z, gout)[:, :, 0], z_ref, 1e-5, 1e-5, "conv4 fwd")
  # BatchNorm moments written by the epilogue: exact merge of the partials must give the layer's mean / M2
  nparts = lib.as_conv4_stat_parts(g4, gout, shape)
  sm = torch.zeros(nparts * 32, device=DEV); s2 = torch.zeros(nparts * 32, device=DEV); sc = torch.zeros(nparts, device=DEV)
  z2 = ops.pcl_zeros(gout, DEV)
  nat.call("as_conv4_fwd", nat.ptr(x4), g4, nat.ptr(wp), nat.ptr(bd), nat.ptr(z2), gout, shape, 0, None, None, 0.2,
           nat.ptr(sm), nat.ptr(s2), nat.ptr(sc), nat.stream())
  assert torch.equal(z2, z), "moments must not change the output"
  cnt = sc.double().cpu(); mean_i = sm.view(nparts, 32).double().cpu(); m2_i = s2.view(nparts, 32).double().cpu()
  n_tot = cnt.sum()
  assert int(n_tot) == B * Ho * Wo
  mean = (cnt[:, None] * mean_i).sum(0) / n_tot
  m2 = (m2_i + cnt[:, None] * (mean_i - mean) ** 2).sum(0)
  zr = z_ref.detach().double().permute(1, 0, 2, 3).reshape(32, -1)
  close(mean.float(), zr.mean(1).float(), 1e-5, 1e-5, "conv4 moments: mean")
  close((m2 / n_tot).float(), zr.var(1, unbiased=False).float(), 1e-5, 1e-4, "conv4 moments: variance")
  # fused eval epilogue: lrelu(z*scale + shift)
  scl = rnd(32, seed=7).abs().to(DEV) + 0.5; shf = rnd(32, seed=8).to(DEV)
  z3 = ops.pcl_zeros(gout, DEV)
  nat.call("as_conv4_fwd", nat.ptr(x4), g4, nat.ptr(wp), nat.ptr(bd), nat.ptr(z3), gout, shape, 1, nat.ptr(scl), nat.ptr(shf),
           0.2, None, None, None, nat.stream())
  ref3 = F.leaky_relu(z_ref.detach() * scl.cpu().view(1, 32, 1, 1) + shf.cpu().view(1, 32, 1, 1), 0.2)
  close(ops.pcl_to_ncdhw(z3, gout)[:, :, 0], ref3, 1e-5, 1e-5, "conv4 fused epilogue")
  gzb = ops.ncdhw_to_pcl(gz.unsqueeze(2).to(DEV), gout)
  dW = torch.empty(32, Cin, k, k, device=DEV); db = torch.empty(32, device=DEV)
  ws = torch.empty(lib.as_conv4_wgrad_workspace(gout, shape), device=DEV)
  nat.call("as_conv4_wgrad", nat.ptr(x4), g4, nat.ptr(gzb), gout, shape, Cin, nat.ptr(dW), nat.ptr(db), 0, nat.ptr(ws),
           nat.stream())
  n = B * Ho * Wo
  close(dW, w.grad, 2e-6 * n ** 0.5 + 1e-5, 2e-5, "conv4 wgrad")
  close(db, b.grad, 2e-6 * n ** 0.5 + 1e-5, 2e-5, "conv4 bias grad")


def test_conv32_stride2_fwd_wgrad_and_residual_epilogue():
  B, H, W = 2, 21, 30
  shape = ConvShape(1, 5, 5, 0, 2, 2, 1, 2)
  x = rnd(B, 32, H, W, seed=1).requires_grad_(True)
  w = rnd(32, 32, 5, 5, seed=2, scale=0.035).requires_grad_(True)
  b = rnd(32, seed=3, scale=0.1).requires_grad_(True)
  z_ref = F.conv2d(x, w, b, stride=2, padding=2)
  gz = rnd(*z_ref.shape, seed=4)
  z_ref.backward(gz)
  Ho, Wo = z_ref.shape[-2:]
  gin, gout = Pcl(B, 1, H, W, 0, 2, 2), Pcl(B, 1, Ho, Wo, 0, 1, 1)
  xb = ops.ncdhw_to_pcl(x.detach().unsqueeze(2).to(DEV), gin)
  wd, bd = w.detach().to(DEV), b.detach().to(DEV)
  res = rnd(B, 32, 1, Ho, Wo, seed=5)
  resb = ops.ncdhw_to_pcl(res.to(DEV), gout)
  zb = ops.conv32(xb, gin, ops.pack_weights(wd, shape, False), bd, gout, shape, residual=resb)
  close(ops.pcl_to_ncdhw(zb, gout), z_ref.unsqueeze(2) + res, 2e-5, 1e-5, "stride-2 conv + residual epilogue")
  gzb = ops.ncdhw_to_pcl(gz.unsqueeze(2).to(DEV), gout)
  dW, db = ops.conv32_wgrad(xb, gin, gzb, gout, shape)
  close(dW, w.grad, 1e-4, 2e-5, "stride-2 wgrad"); close(db, b.grad, 1e-4, 2e-5, "stride-2 bias grad")


@pytest.mark.parametrize("B,H,W", [(1, 9, 13), (2, 37, 53)])
def test_conv32to1_2d_fused_tail(B, H, W):
  """EdgeAwareRefinement tail: relu(up + Conv2d(32,1,3,p=1)(a)), forward and backward."""
  g = Pcl(B, 1, H, W, 0, 8, 8)
  shape = ops.conv_shape_2d(1)
  a = rnd(B, 32, H, W, seed=1).requires_grad_(True)
  w = rnd(1, 32, 3, 3, seed=2, scale=0.2).requires_grad_(True)
  b = rnd(1, seed=3, scale=0.1).requires_grad_(True)
  up = rnd(B, 1, H, W, seed=4)
  ref = F.relu(up + F.conv2d(a, w, b, padding=1))
  go = rnd(B, 1, H, W, seed=5)
  ref.backward(go)
  ab = ops.ncdhw_to_pcl(a.detach().unsqueeze(2).to(DEV), g)
  wd, bd, upd = w.detach().to(DEV).contiguous(), b.detach().to(DEV), up.to(DEV)
  out = torch.empty(B, 1, H, W, device=DEV)
  nat.call("as_conv32to1_fwd", nat.ptr(ab), g, shape, nat.ptr(wd), nat.ptr(bd), nat.ptr(upd), 1, nat.ptr(out), nat.stream())
  close(out, ref, 5e-6, 1e-5, "refinement tail fwd")
  gpre = (go.to(DEV) * (out > 0)).contiguous()
  lib = nat.load()
  g_a = ops.pcl_zeros(g, DEV); g_w = torch.empty_like(wd); g_b = torch.empty(1, device=DEV)
  ws = torch.empty(lib.as_conv32to1_bwd_workspace(g, shape), device=DEV)
  nat.call("as_conv32to1_bwd", nat.ptr(gpre), nat.ptr(ab), g, shape, nat.ptr(wd), nat.ptr(g_a), nat.ptr(g_w),
           nat.ptr(g_b), 0, nat.ptr(ws), nat.stream())
  # accumulate=1 adds on top of what is already there
  g_w2 = g_w.clone(); g_b2 = g_b.clone()
  nat.call("as_conv32to1_bwd", nat.ptr(gpre), nat.ptr(ab), g, shape, nat.ptr(wd), None, nat.ptr(g_w2),
           nat.ptr(g_b2), 1, nat.ptr(ws), nat.stream())
  close(g_w2, 2 * g_w, 1e-6, 1e-6, "tail wgrad accumulate"); close(g_b2, 2 * g_b, 1e-6, 1e-6, "tail bias accumulate")
  close(ops.pcl_to_ncdhw(g_a, g)[:, :, 0], a.grad, 5e-6, 1e-5, "tail dgrad")
  n = B * H * W
  close(g_w, w.grad, 2e-6 * n ** 0.5, 1e-5, "tail wgrad"); close(g_b, b.grad, 2e-6 * n ** 0.5, 1e-5, "tail bias grad")


@pytest.mark.parametrize("B,H,W", [(1, 9, 13), (2, 37, 53), (3, 33, 126), (2, 65, 127), (1, 96, 253), (2, 375, 1242)])
def test_refinement_output_layer_with_the_last_activation_on_the_way_in(B, H, W):
  """as_refine_out_fwd (csrc/refine_out.hip): relu(up + Conv2d(32,1,3,p=1)(a)) with a = lrelu(z * scale + shift) + skip formed
  while the rows are staged — against torch on the CPU, and against the two launches it replaces (as_bn_act_fwd, then
  as_conv32to1_fwd): the by-product a bit for bit (the backward pass reads it), nothing written into its halo; and the plain
  form (a given, inference) against as_conv32to1_fwd.  Widths of one strip, one strip + 1, two strips + 1, the KITTI size;
  heights that do not divide by the 32-row strips."""
  g = Pcl(B, 1, H, W, 0, 8, 8)
  lib = nat.load()
  assert lib.as_refine_out_ok(g) == 1
  shape = ops.conv_shape_2d(1)
  z = rnd(B, 32, H, W, seed=1); skip = rnd(B, 32, H, W, seed=6)
  w = rnd(1, 32, 3, 3, seed=2, scale=0.2); b = rnd(1, seed=3, scale=0.1); up = rnd(B, 1, H, W, seed=4)
  st = ops.BnState(DEV)
  st.scale.copy_(rnd(32, seed=7).abs().to(DEV) + 0.5); st.shift.copy_(rnd(32, seed=8).to(DEV) * 0.3)
  sc, sh = st.scale.cpu().view(1, 32, 1, 1), st.shift.cpu().view(1, 32, 1, 1)
  a_ref = F.leaky_relu(z * sc + sh, 0.2) + skip
  ref = F.relu(up + F.conv2d(a_ref.double(), w.double(), b.double(), padding=1).float())
  zb, sb = ops.ncdhw_to_pcl(z.unsqueeze(2).to(DEV), g), ops.ncdhw_to_pcl(skip.unsqueeze(2).to(DEV), g)
  wd, bd, upd = w.to(DEV).contiguous(), b.to(DEV), up.to(DEV)
  # the two launches
  a_two = ops.bn_act(zb, st, g, residual=sb, out=ops.pcl_zeros(g, DEV))
  out_two = torch.empty(B, 1, H, W, device=DEV)
  nat.call("as_conv32to1_fwd", nat.ptr(a_two), g, shape, nat.ptr(wd), nat.ptr(bd), nat.ptr(upd), 1, nat.ptr(out_two), nat.stream())
  # one launch
  a_one, out_one = ops.pcl_zeros(g, DEV), torch.empty(B, 1, H, W, device=DEV)
  nat.call("as_refine_out_fwd", nat.ptr(zb), nat.ptr(sb), nat.ptr(st.scale), nat.ptr(st.shift), 0.2, nat.ptr(a_one), g, nat.ptr(wd),
           nat.ptr(bd), nat.ptr(upd), 1, nat.ptr(out_one), nat.stream())
  assert bool(torch.equal(a_one, a_two)), "by-product differs from as_bn_act_fwd (or the halo was written)"
  scale = float(ref.abs().max())
  e_one, e_two = float((out_one.cpu() - ref).abs().max()), float((out_two.cpu() - ref).abs().max())
  assert e_one <= max(2.0 * e_two, 2e-6 * scale), (e_one, e_two, scale)
  # no skip connection; no add_src, no ReLU
  a_ns, out_ns = ops.pcl_zeros(g, DEV), torch.empty(B, 1, H, W, device=DEV)
  nat.call("as_refine_out_fwd", nat.ptr(zb), None, nat.ptr(st.scale), nat.ptr(st.shift), 0.2, nat.ptr(a_ns), g, nat.ptr(wd),
           None, None, 0, nat.ptr(out_ns), nat.stream())
  a_ns_ref = F.leaky_relu(z * sc + sh, 0.2)
  close(ops.pcl_to_ncdhw(a_ns, g)[:, :, 0], a_ns_ref, 2e-6, 1e-6, "activation without skip")
  close(out_ns, F.conv2d(a_ns_ref.double(), w.double(), None, padding=1).float(), 4e-6 * scale, 1e-5, "plain output")
  # inference form: a given
  out_inf = torch.empty(B, 1, H, W, device=DEV)
  nat.call("as_refine_out_fwd", nat.ptr(a_two), None, None, None, 0.2, None, g, nat.ptr(wd), nat.ptr(bd), nat.ptr(upd), 1,
           nat.ptr(out_inf), nat.stream())
  assert bool(torch.equal(out_inf, out_one)), "the inference form must give the fused form's bits (same staged values)"
  from conftest import parity_note
  parity_note("refine_out[B%d %dx%d]" % (B, H, W), max_err_vs_fp64=e_one, two_launch_max_err_vs_fp64=e_two, by_product_bit_identical=True)


@pytest.mark.parametrize("B,H,W", [(1, 9, 12), (2, 21, 30), (1, 47, 156), (1, 188, 621)])
def test_conv32_dgrad_stride2(B, H, W):
  """Data gradient of Conv2d(32,32,5,stride=2,padding=2): four parity phases of a transposed conv."""
  x = rnd(B, 32, H, W, seed=1).requires_grad_(True)
  w = rnd(32, 32, 5, 5, seed=2, scale=0.035)
  z = F.conv2d(x, w, None, stride=2, padding=2)
  gz = rnd(*z.shape, seed=3)
  z.backward(gz)
  Ho, Wo = z.shape[-2:]
  ggz, ggx = Pcl(B, 1, Ho, Wo, 0, 1, 1), Pcl(B, 1, H, W, 0, 2, 2)
  gzb = ops.ncdhw_to_pcl(gz.unsqueeze(2).to(DEV), ggz)
  gxb = ops.pcl_zeros(ggx, DEV)
  lib = nat.load()
  wd = w.to(DEV)
  ws = torch.empty(lib.as_conv32_dgrad_s2_workspace(), device=DEV)
  nat.call("as_conv32_dgrad_s2", nat.ptr(gzb), ggz, nat.ptr(wd), nat.ptr(gxb), ggx, nat.ptr(ws), nat.stream())
  close(ops.pcl_to_ncdhw(gxb, ggx)[:, :, 0], x.grad, 3e-5, 1e-5, "stride-2 dgrad")
  full = ops.pcl_view(gxb, ggx).clone(); ops.pcl_interior(full, ggx).zero_()
  assert float(full.abs().max()) == 0.0, "dgrad wrote into the halo"


@pytest.mark.parametrize("B,H,W", [(8, 188, 621), (2, 375, 1242), (8, 94, 311), (40, 21, 33), (6, 130, 350)])
def test_strided_head_on_staged_rows_equals_the_generic_kernels(B, H, W):
  """Conv2d(32,32,5,stride=2,padding=2) forward and data gradient on csrc/conv32_s2.hip (coalesced row segments through
  wave-private LDS) against the generic direct-load kernels of conv32_mfma.hip (as_conv32_s2_enable(0)): bit for bit — the taps
  and the channels are summed in the same order —, nothing written into a halo; forward, data gradient and the weight gradient
  (generic kernel in both settings: a staged-row variant was measured and not kept, profiles/r04_l_*) against torch on the
  CPU.  The bench workload's two levels (8 images of 188 x 621 and 94 x 311), the full-resolution level of
  a k = 3 tower, odd extents; the output halo is 1 where the tower's last level has 1."""
  shape = ConvShape(1, 5, 5, 0, 2, 2, 1, 2)
  Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
  gin, gout = Pcl(B, 1, H, W, 0, 2, 2), Pcl(B, 1, Ho, Wo, 0, 1 + (H & 1), 1 + (H & 1))
  lib = nat.load()
  x = rnd(B, 32, H, W, seed=1); w = rnd(32, 32, 5, 5, seed=2, scale=0.035); b = rnd(32, seed=3, scale=0.1)
  gz = rnd(B, 32, Ho, Wo, seed=4)
  xb = ops.ncdhw_to_pcl(x.unsqueeze(2).to(DEV), gin); gzb = ops.ncdhw_to_pcl(gz.unsqueeze(2).to(DEV), gout)
  wd, bd = w.to(DEV), b.to(DEV)
  wp = ops.pack_weights(wd, shape, False)
  ws = torch.empty(lib.as_conv32_dgrad_s2_workspace(), device=DEV)
  res = {}
  prev = lib.as_conv32_s2_enable(2)
  try:
    for on in (0, 1):
      lib.as_conv32_s2_enable(on)
      z = ops.conv32(xb, gin, wp, bd, gout, shape, out=ops.pcl_zeros(gout, DEV))
      gx = ops.pcl_zeros(gin, DEV)
      nat.call("as_conv32_dgrad_s2", nat.ptr(gzb), gout, nat.ptr(wd), nat.ptr(gx), gin, nat.ptr(ws), nat.stream())
      dW, db = ops.conv32_wgrad(xb, gin, gzb, gout, shape)
      torch.cuda.synchronize()
      res[on] = (z, gx, dW, db)
  finally:
    lib.as_conv32_s2_enable(prev)
  for name, a, c in (("forward", res[0][0], res[1][0]), ("data gradient", res[0][1], res[1][1])):
    assert bool(torch.equal(a, c)), "%s: staged rows differ from the generic kernel in %d elements" % (name, int((a != c).sum()))
  xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
  z_ref = F.conv2d(xr, wr, b, stride=2, padding=2)
  z_ref.backward(gz)
  close(ops.pcl_to_ncdhw(res[1][0], gout)[:, :, 0], z_ref, 3e-5, 1e-5, "stride-2 forward")
  close(ops.pcl_to_ncdhw(res[1][1], gin)[:, :, 0], xr.grad, 3e-5, 1e-5, "stride-2 data gradient")
  scale = float(wr.grad.abs().max())
  for on in (0, 1):
    close(res[on][2] / scale, wr.grad / scale, 5e-5, 0.0, "stride-2 weight gradient (s2 switch: %d)" % on)
    close(res[on][3] / scale, gz.double().sum((0, 2, 3)) / scale, 5e-5, 0.0, "stride-2 bias gradient (%d)" % on)
  assert bool(torch.equal(res[1][2], res[0][2])), "the s2 switch must not change the weight gradient"


@pytest.mark.parametrize("B,H,W,dil", [(2, 9, 131, 1), (1, 20, 300, 2)])
def test_conv32_dgrad_fused_with_bn_backward_sums(B, H, W, dil):
  """as_conv32_fwd_bnbwd (data gradient + stage 1 of the next BatchNorm backward) followed by as_bn_act_bwd_given must
  give what as_conv32_fwd followed by as_bn_act_bwd gives: the same g_x bit for bit (same kernel arithmetic), g_z and the
  gamma/beta gradients up to the different summation order of the per-channel sums."""
  g = Pcl(B, 1, H, W, 0, 8, 8)
  shape = ops.conv_shape_2d(dil)
  gz = ops.ncdhw_to_pcl(rnd(B, 32, 1, H, W, seed=1).to(DEV), g)
  res = ops.ncdhw_to_pcl(rnd(B, 32, 1, H, W, seed=2).to(DEV), g)
  zprev = ops.ncdhw_to_pcl(rnd(B, 32, 1, H, W, seed=3).to(DEV), g)
  w = (rnd(32, 32, 3, 3, seed=4) * 0.06).to(DEV)
  wp_t = ops.pack_weights(w, shape, True)
  st = ops.BnState(DEV)
  st.mean.copy_(rnd(32, seed=5).to(DEV) * 0.1); st.invstd.copy_(rnd(32, seed=6).abs().to(DEV) + 0.5)
  gamma = (rnd(32, seed=7).abs() + 0.5).to(DEV)
  st.scale.copy_(st.invstd * gamma); st.shift.copy_(rnd(32, seed=8).to(DEV) * 0.1 - st.mean * st.scale)
  # separate path
  gx_ref = ops.conv32(gz, g, wp_t, None, g, shape, residual=res)
  gzp_ref, gg_ref, gb_ref = ops.bn_act_bwd(gx_ref, zprev, st, gamma, g, True)
  # fused path
  fused = ops.conv32_dgrad_bnbwd(gz, g, wp_t, shape, res, zprev, st)
  assert fused is not None
  gx, sums = fused
  assert torch.equal(gx, gx_ref)
  gzp, gg, gb = ops.bn_act_bwd(gx, zprev, st, gamma, g, True, sums=sums)
  n = B * H * W
  close(gg, gg_ref, 2e-6 * n ** 0.5 * float(gg_ref.abs().max()) + 1e-5, 1e-5, "fused BN backward: g_gamma")
  close(gb, gb_ref, 2e-6 * n ** 0.5 * float(gb_ref.abs().max()) + 1e-5, 1e-5, "fused BN backward: g_beta")
  close(ops.pcl_interior(ops.pcl_view(gzp, g), g), ops.pcl_interior(ops.pcl_view(gzp_ref, g), g), 1e-5, 1e-4, "fused BN backward: g_z")


def test_conv32_wgrad_fused_with_bn_backward_apply():
  """as_bn_act_bwd(g_z=NULL) + as_conv32_wgrad_bnapply (stage 3 of the BatchNorm backward applied to the staged gradient
  row inside the weight-gradient kernel, g_z written as a by-product) against the separate passes, at a size that takes the
  double-buffered kernel (>= 4096 row segments) with a ragged last segment."""
  B, H, W, dil = 1, 421, 1250, 2
  g = Pcl(B, 1, H, W, 0, 8, 8)
  shape = ops.conv_shape_2d(dil)
  lib = nat.load()
  assert lib.as_conv32_wgrad_bnapply_ok(g, g, shape) == 1
  x = ops.ncdhw_to_pcl(rnd(B, 32, 1, H, W, seed=1).to(DEV), g)
  g_a = ops.ncdhw_to_pcl(rnd(B, 32, 1, H, W, seed=2).to(DEV), g)
  z = ops.ncdhw_to_pcl(rnd(B, 32, 1, H, W, seed=3).to(DEV), g)
  st = ops.BnState(DEV)
  st.mean.copy_(rnd(32, seed=5).to(DEV) * 0.1); st.invstd.copy_(rnd(32, seed=6).abs().to(DEV) + 0.5)
  gamma = (rnd(32, seed=7).abs() + 0.5).to(DEV)
  st.scale.copy_(st.invstd * gamma); st.shift.copy_(rnd(32, seed=8).to(DEV) * 0.1 - st.mean * st.scale)
  # separate passes
  gz_ref, gg_ref, gb_ref = ops.bn_act_bwd(g_a, z, st, gamma, g, True)
  dW_ref, db_ref = ops.conv32_wgrad(x, g, gz_ref, g, shape)
  # fused
  ws = torch.empty(lib.as_bn_bwd_workspace(g), device=DEV)
  gg, gb = torch.zeros(32, device=DEV), torch.zeros(32, device=DEV)
  nat.call("as_bn_act_bwd", nat.ptr(g_a), nat.ptr(z), nat.ptr(st.scale), nat.ptr(st.shift), nat.ptr(st.mean),
           nat.ptr(st.invstd), nat.ptr(gamma), 0.2, 1, None, nat.ptr(gg), nat.ptr(gb), 0, nat.ptr(ws), g, nat.stream())
  assert torch.equal(gg, gg_ref) and torch.equal(gb, gb_ref)
  coef = ws[lib.as_bn_bwd_coef_offset():]
  gz = ops.pcl_zeros(g, DEV)
  dW = torch.zeros(32, 32, 3, 3, device=DEV); db = torch.zeros(32, device=DEV)
  wws = torch.empty(lib.as_conv32_wgrad_workspace(g, g, shape), device=DEV)
  nat.call("as_conv32_wgrad_bnapply", nat.ptr(x), g, nat.ptr(g_a), nat.ptr(z), g, shape, nat.ptr(st.scale), nat.ptr(st.shift),
           nat.ptr(st.mean), nat.ptr(coef), 0.2, nat.ptr(gz), nat.ptr(dW), nat.ptr(db), 0, nat.ptr(wws), nat.stream())
  gzv, gzr = ops.pcl_view(gz, g), ops.pcl_view(gz_ref, g)
  assert torch.equal(ops.pcl_interior(gzv, g), ops.pcl_interior(gzr, g)), "stage 3 in LDS must round like the element-wise pass"
  halo = gzv.clone(); ops.pcl_interior(halo, g).zero_()
  assert float(halo.abs().max()) == 0.0, "the halo of g_z must stay zero"
  assert torch.equal(dW, dW_ref) and torch.equal(db, db_ref)


def test_conv4_wgrad_fused_with_bn_backward_apply():
  """as_conv4_wgrad_bnapply against as_bn_act_bwd + as_conv4_wgrad: g_z bit for bit (same arithmetic per element),
  dW / db bit for bit (same products, same order)."""
  B, H, W = 2, 37, 150
  g4, g = Pcl(B, 1, H, W, 0, 8, 8), Pcl(B, 1, H, W, 0, 8, 8)
  shape = ops.conv_shape_2d(1)
  lib = nat.load()
  assert lib.as_conv4_wgrad_bnapply_ok(g4, g, shape) == 1
  x4 = torch.zeros(lib.as_pcl4_numel(g4), device=DEV)
  ch0, img = rnd(B, 1, H, W, seed=1).to(DEV), rnd(B, 3, H, W, seed=2).to(DEV)
  nat.call("as_pack_in4", nat.ptr(ch0), nat.ptr(img), 3, nat.ptr(x4), g4, nat.stream())
  g_a = ops.ncdhw_to_pcl(rnd(B, 32, 1, H, W, seed=3).to(DEV), g)
  z = ops.ncdhw_to_pcl(rnd(B, 32, 1, H, W, seed=4).to(DEV), g)
  st = ops.BnState(DEV)
  st.mean.copy_(rnd(32, seed=5).to(DEV) * 0.1); st.invstd.copy_(rnd(32, seed=6).abs().to(DEV) + 0.5)
  gamma = (rnd(32, seed=7).abs() + 0.5).to(DEV)
  st.scale.copy_(st.invstd * gamma); st.shift.copy_(rnd(32, seed=8).to(DEV) * 0.1 - st.mean * st.scale)
  gz_ref, gg_ref, gb_ref = ops.bn_act_bwd(g_a, z, st, gamma, g, True)
  dW_ref = torch.zeros(32, 4, 3, 3, device=DEV); db_ref = torch.zeros(32, device=DEV)
  ws = torch.empty(lib.as_conv4_wgrad_workspace(g, shape), device=DEV)
  nat.call("as_conv4_wgrad", nat.ptr(x4), g4, nat.ptr(gz_ref), g, shape, 4, nat.ptr(dW_ref), nat.ptr(db_ref), 0, nat.ptr(ws),
           nat.stream())
  bws = torch.empty(lib.as_bn_bwd_workspace(g), device=DEV)
  gg, gb = torch.zeros(32, device=DEV), torch.zeros(32, device=DEV)
  nat.call("as_bn_act_bwd", nat.ptr(g_a), nat.ptr(z), nat.ptr(st.scale), nat.ptr(st.shift), nat.ptr(st.mean),
           nat.ptr(st.invstd), nat.ptr(gamma), 0.2, 1, None, nat.ptr(gg), nat.ptr(gb), 0, nat.ptr(bws), g, nat.stream())
  coef = bws[lib.as_bn_bwd_coef_offset():]
  gz = ops.pcl_zeros(g, DEV)
  dW = torch.zeros(32, 4, 3, 3, device=DEV); db = torch.zeros(32, device=DEV)
  nat.call("as_conv4_wgrad_bnapply", nat.ptr(x4), g4, nat.ptr(g_a), nat.ptr(z), g, shape, 4, nat.ptr(st.scale),
           nat.ptr(st.shift), nat.ptr(st.mean), nat.ptr(coef), 0.2, nat.ptr(gz), nat.ptr(dW), nat.ptr(db), 0, nat.ptr(ws), nat.stream())
  assert torch.equal(gg, gg_ref) and torch.equal(gb, gb_ref)
  assert torch.equal(ops.pcl_interior(ops.pcl_view(gz, g), g), ops.pcl_interior(ops.pcl_view(gz_ref, g), g))
  halo = ops.pcl_view(gz, g).clone(); ops.pcl_interior(halo, g).zero_()
  assert float(halo.abs().max()) == 0.0
  assert torch.equal(dW, dW_ref) and torch.equal(db, db_ref)


@pytest.mark.parametrize("B,H,W", [(2, 37, 150), (1, 61, 1242), (3, 5, 129)])
def test_conv4_wgrad_with_per_tap_projections_instead_of_g_z(B, H, W):
  """as_conv4_wgrad_bnapply_proj + as_tap_gather (conv2d_feature's backward without a g_z tensor: nine per-tap projections
  of g_z go out, a gather sums the shifted planes; LDS-staged tile, products on the matrix cores) against
  as_conv4_wgrad_bnapply + the 32->1 convolution of g_z: dW / db and the disparity-channel data gradient to fp32 summation
  order (the data gradient: one chain of 288 products per pixel there, 9 chains of 32 here; both equally close to fp64)."""
  g4, g = Pcl(B, 1, H, W, 0, 8, 8), Pcl(B, 1, H, W, 0, 8, 8)
  shape = ops.conv_shape_2d(1)
  lib = nat.load()
  x4 = torch.zeros(lib.as_pcl4_numel(g4), device=DEV)
  ch0, img = rnd(B, 1, H, W, seed=1).to(DEV), rnd(B, 3, H, W, seed=2).to(DEV)
  nat.call("as_pack_in4", nat.ptr(ch0), nat.ptr(img), 3, nat.ptr(x4), g4, nat.stream())
  g_a = ops.ncdhw_to_pcl(rnd(B, 32, 1, H, W, seed=3).to(DEV), g)
  z = ops.ncdhw_to_pcl(rnd(B, 32, 1, H, W, seed=4).to(DEV), g)
  g_pre = rnd(B, 1, H, W, seed=11).to(DEV).contiguous()
  w0 = (rnd(32, 4, 3, 3, seed=12) * 0.2).to(DEV)
  st = ops.BnState(DEV)
  st.mean.copy_(rnd(32, seed=5).to(DEV) * 0.1); st.invstd.copy_(rnd(32, seed=6).abs().to(DEV) + 0.5)
  gamma = (rnd(32, seed=7).abs() + 0.5).to(DEV)
  st.scale.copy_(st.invstd * gamma); st.shift.copy_(rnd(32, seed=8).to(DEV) * 0.1 - st.mean * st.scale)
  ws = torch.empty(lib.as_conv4_wgrad_workspace(g, shape), device=DEV)
  bws = torch.empty(lib.as_bn_bwd_workspace(g), device=DEV)
  gg, gb = torch.zeros(32, device=DEV), torch.zeros(32, device=DEV)
  nat.call("as_bn_act_bwd", nat.ptr(g_a), nat.ptr(z), nat.ptr(st.scale), nat.ptr(st.shift), nat.ptr(st.mean),
           nat.ptr(st.invstd), nat.ptr(gamma), 0.2, 1, None, nat.ptr(gg), nat.ptr(gb), 0, nat.ptr(bws), g, nat.stream())
  coef = bws[lib.as_bn_bwd_coef_offset():]
  # with g_z
  gz = ops.pcl_zeros(g, DEV)
  dW_ref = torch.zeros(32, 4, 3, 3, device=DEV); db_ref = torch.zeros(32, device=DEV)
  nat.call("as_conv4_wgrad_bnapply", nat.ptr(x4), g4, nat.ptr(g_a), nat.ptr(z), g, shape, 4, nat.ptr(st.scale), nat.ptr(st.shift),
           nat.ptr(st.mean), nat.ptr(coef), 0.2, nat.ptr(gz), nat.ptr(dW_ref), nat.ptr(db_ref), 0, nat.ptr(ws), nat.stream())
  w_ch0 = w0[:, 0].flip(-1, -2).reshape(32, 9).contiguous()
  g_up_ref = torch.empty(B, 1, H, W, device=DEV)
  nat.call("as_conv32to1_fwd", nat.ptr(gz), g, shape, nat.ptr(w_ch0), None, nat.ptr(g_pre), 0, nat.ptr(g_up_ref), nat.stream())
  # without
  w_proj = w_ch0.t().contiguous()
  h = torch.full((B * 9 * H * W,), float("nan"), device=DEV)
  dW = torch.zeros(32, 4, 3, 3, device=DEV); db = torch.zeros(32, device=DEV)
  nat.call("as_conv4_wgrad_bnapply_proj", nat.ptr(x4), g4, nat.ptr(g_a), nat.ptr(z), g, shape, 4, nat.ptr(st.scale),
           nat.ptr(st.shift), nat.ptr(st.mean), nat.ptr(coef), 0.2, nat.ptr(w_proj), nat.ptr(h), nat.ptr(dW), nat.ptr(db), 0,
           nat.ptr(ws), nat.stream())
  g_up = torch.empty(B, 1, H, W, device=DEV)
  nat.call("as_tap_gather", nat.ptr(h), nat.ptr(g_pre), nat.ptr(g_up), B, H, W, nat.stream())
  assert bool(torch.isfinite(h).all()), "a projection was not written"
  for name, got, exp in (("dW", dW, dW_ref), ("db", db, db_ref)):      # matrix cores, pixel pairs as K: another summation order
    rel = float((got.double() - exp.double()).norm() / exp.double().norm())
    assert rel < 2e-5, "%s: relative L2 error %.2e" % (name, rel)
  # fp64 truth from the g_z tensor: both paths must be equally close to it
  gz_i = ops.pcl_to_ncdhw(gz, g)[:, :, 0].double().cpu()
  truth = torch.nn.functional.conv2d(gz_i, w0[:, 0].flip(-1, -2).double().cpu().reshape(1, 32, 3, 3), padding=1) + g_pre.double().cpu()
  err_new = float((g_up.double().cpu() - truth).abs().max()); err_ref = float((g_up_ref.double().cpu() - truth).abs().max())
  scale = float(truth.abs().max())
  assert err_new <= 2e-6 * scale and err_ref <= 2e-6 * scale, (err_new, err_ref, scale)
  from conftest import parity_note
  parity_note("head_proj[B%d H%d W%d]" % (B, H, W), max_err_vs_fp64=err_new, two_launch_path_err=err_ref, scale=scale)


@pytest.mark.parametrize("B,H,W", [(2, 37, 150), (1, 61, 1242), (3, 5, 129)])
def test_output_layer_data_gradient_with_batchnorm_backward_sums(B, H, W):
  """as_conv32to1_dgrad_bnsums — conv2d_out's data gradient that also leaves stage 1 of the consuming BatchNorm backward
  behind — against as_conv32to1_bwd followed by as_bn_act_bwd's own stage 1: g_a bit for bit, the per-channel parameter
  gradients that come out of the sums to fp32 summation order."""
  g = Pcl(B, 1, H, W, 0, 8, 8)
  s33 = ops.conv_shape_2d(1)
  lib = nat.load()
  assert lib.as_conv32to1_bnsums_ok(g, s33) == 1
  g_pre = rnd(B, 1, H, W, seed=1).to(DEV).contiguous()
  a = ops.ncdhw_to_pcl(rnd(B, 32, 1, H, W, seed=2).to(DEV), g)
  z = ops.ncdhw_to_pcl(rnd(B, 32, 1, H, W, seed=3).to(DEV), g)
  w_out = (rnd(1, 32, 3, 3, seed=4) * 0.2).to(DEV)
  st = ops.BnState(DEV)
  st.mean.copy_(rnd(32, seed=5).to(DEV) * 0.1); st.invstd.copy_(rnd(32, seed=6).abs().to(DEV) + 0.5)
  gamma = (rnd(32, seed=7).abs() + 0.5).to(DEV)
  st.scale.copy_(st.invstd * gamma); st.shift.copy_(rnd(32, seed=8).to(DEV) * 0.1 - st.mean * st.scale)
  # two passes
  g_a_ref = ops.pcl_zeros(g, DEV)
  ws = torch.empty(lib.as_conv32to1_bwd_workspace(g, s33), device=DEV)
  gw, gb = torch.zeros_like(w_out), torch.zeros(1, device=DEV)
  nat.call("as_conv32to1_bwd", nat.ptr(g_pre), nat.ptr(a), g, s33, nat.ptr(w_out), nat.ptr(g_a_ref), nat.ptr(gw), nat.ptr(gb), 0,
           nat.ptr(ws), nat.stream())
  gz_ref, gg_ref, gb_ref = ops.bn_act_bwd(g_a_ref, z, st, gamma, g, True)
  # one pass + given sums
  g_a = ops.pcl_zeros(g, DEV)
  nws = torch.empty(lib.as_bn_bwd_workspace(g), device=DEV)
  nat.call("as_conv32to1_dgrad_bnsums", nat.ptr(g_pre), g, s33, nat.ptr(w_out), nat.ptr(g_a), nat.ptr(z), nat.ptr(st.scale),
           nat.ptr(st.shift), nat.ptr(st.mean), 0.2, nat.ptr(nws), nat.stream())
  assert torch.equal(ops.pcl_view(g_a, g), ops.pcl_view(g_a_ref, g))
  gz, gg, gbeta = ops.bn_act_bwd(g_a, z, st, gamma, g, True, sums=ops.BnBwdSums(nws, lib.as_conv32to1_bnsums_parts(g)))
  n = B * H * W
  close(gg, gg_ref, 2e-6 * n ** 0.5 * float(gg_ref.abs().max()) + 1e-5, 1e-5, "g_gamma")
  close(gbeta, gb_ref, 2e-6 * n ** 0.5 * float(gb_ref.abs().max()) + 1e-5, 1e-5, "g_beta")
  close(ops.pcl_interior(ops.pcl_view(gz, g), g), ops.pcl_interior(ops.pcl_view(gz_ref, g), g), 2e-5, 1e-4, "g_z")


def test_conv3d_lds_random_geometries():
  """The flattened-plane 3-D kernels (forward, data gradient, weight gradient, moments) over a sweep of geometries around
  their applicability edges: the narrowest row (W + 2 = 34), planes of just over one tile, every remainder class of the
  shifted last tile, one and several disparity planes, batch > 1 — against torch conv3d and its autograd."""
  shape = ConvShape(3, 3, 3, 1, 1, 1, 1, 1)
  rng = torch.Generator().manual_seed(2024)
  geoms = [(1, 1, 4, 32), (1, 2, 5, 32), (2, 3, 4, 33), (1, 1, 3, 62), (1, 4, 7, 45), (3, 2, 6, 41), (1, 5, 9, 64),
           (1, 1, 2, 126), (2, 2, 3, 100), (1, 3, 13, 37), (1, 2, 24, 78), (1, 1, 5, 188)]
  lib = nat.load()
  for (B, D, H, W) in geoms:
    g = Pcl(B, D, H, W, 1, 1, 1)
    assert lib.as_conv32_stat_parts(g, g, shape) == B * D * (((H - 1) * (W + 2) + W + 127) // 128), (B, D, H, W)
    x = torch.randn(B, 32, D, H, W, generator=rng).requires_grad_(True)
    w = (torch.randn(32, 32, 3, 3, 3, generator=rng) / (32 * 27) ** 0.5).requires_grad_(True)
    b = (torch.randn(32, generator=rng) * 0.1).requires_grad_(True)
    z_ref = F.conv3d(x, w, b, padding=1)
    gz = torch.randn(z_ref.shape, generator=rng)
    z_ref.backward(gz)
    xb = ops.ncdhw_to_pcl(x.detach().to(DEV), g)
    wd = w.detach().to(DEV)
    stats = ops.conv32_stat_parts(g, g, shape, DEV)
    zb = ops.conv32(xb, g, ops.pack_weights(wd, shape, False), b.detach().to(DEV), g, shape, stats=stats)
    tag = "B%d D%d H%d W%d" % (B, D, H, W)
    close(ops.pcl_to_ncdhw(zb, g), z_ref, 2e-5, 1e-5, tag + " fwd")
    full = ops.pcl_view(zb, g).clone(); ops.pcl_interior(full, g).zero_()
    assert float(full.abs().max()) == 0.0, tag + ": wrote into the halo"
    st = ops.bn_train_stats(stats, torch.ones(32, device=DEV), torch.zeros(32, device=DEV), torch.zeros(32, device=DEV),
                            torch.ones(32, device=DEV))
    zr = z_ref.detach()
    close(st.mean, zr.mean(dim=(0, 2, 3, 4)), 3e-6, 1e-5, tag + " mean")
    close(st.invstd, 1.0 / torch.sqrt(zr.var(dim=(0, 2, 3, 4), unbiased=False) + 1e-5), 0, 3e-5, tag + " invstd")
    gzb = ops.ncdhw_to_pcl(gz.to(DEV), g)
    gxb = ops.conv32(gzb, g, ops.pack_weights(wd, shape, True), None, g, shape)
    close(ops.pcl_to_ncdhw(gxb, g), x.grad, 3e-5, 1e-5, tag + " dgrad")
    dW, db = ops.conv32_wgrad(xb, g, gzb, g, shape)
    for name, got, exp in ((" wgrad", dW, w.grad), (" bias grad", db, b.grad)):
      rel = float((got.cpu().double() - exp.double()).norm() / exp.double().norm())
      assert rel < 1e-5, "%s%s: relative L2 error %.2e" % (tag, name, rel)


@pytest.mark.parametrize("B,H,W,dil", [(2, 19, 300, 1), (1, 23, 257, 4), (1, 33, 140, 8), (1, 5, 128, 2)])
def test_conv32_lds_skip_from_staged_tile(B, H, W, dil):
  """Eval forward of a BasicBlock on the LDS-staged kernel: when the residual IS the input buffer the skip connection
  is read from the staged tile (conv32_lds_skip_kernel).  Must equal torch, and — bit for bit — the same call with the
  residual handed over as a separate copy (the three-stream flavour)."""
  shape, halo = ConvShape(1, 3, 3, 0, dil if dil > 1 else 1, dil if dil > 1 else 1, dil, 1), (0, 8, 8)
  x, w, b = rnd(B, 32, 1, H, W, seed=1), rnd(32, 32, 3, 3, seed=2, scale=0.06), rnd(32, seed=3, scale=0.1)
  sc, sh = rnd(32, seed=4) * 0.5 + 1.0, rnd(32, seed=5) * 0.3
  ref = x + F.leaky_relu(_conv_ref(x, w, b, shape) * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1), 0.2)
  g = Pcl(B, 1, H, W, *halo)
  xb = ops.ncdhw_to_pcl(x.to(DEV), g)
  wp = ops.pack_weights(w.to(DEV), shape, False)
  args = dict(epilogue=1, scale=sc.to(DEV), shift=sh.to(DEV))
  out_self = ops.conv32(xb, g, wp, b.to(DEV), g, shape, residual=xb, **args)
  out_copy = ops.conv32(xb, g, wp, b.to(DEV), g, shape, residual=xb.clone(), **args)
  close(ops.pcl_to_ncdhw(out_self, g), ref, 3e-5, 1e-5, "skip from the staged tile")
  assert torch.equal(out_self, out_copy)
  full = ops.pcl_view(out_self, g).clone(); ops.pcl_interior(full, g).zero_()
  assert float(full.abs().max()) == 0.0


# ----------------------------------------------------------------------------- a13
@pytest.mark.parametrize("shape,invalid", [((1, 1, 17, 33), "some"), ((2, 1, 96, 256), "some"), ((3, 1, 5, 7), "all"),
                                           ((1, 1, 375, 1242), "none")])
def test_khamis_robust_loss_fwd_bwd(shape, invalid):
  """as_khamis_fwd / as_khamis_bwd (reference utils/loss_functions.py:6-15) against the oracle and its autograd:
  ragged sizes, no valid pixel at all (the reference divides by max(n, 1)) and a full KITTI-size map."""
  from adaptive_stereo.utils.loss_functions import khamis_robust_loss
  pred = (rnd(*shape, seed=1) * 30 + 40).requires_grad_(True)
  gt = rnd(*shape, seed=2) * 30 + 41
  if invalid == "some":
    gt[..., ::3, ::5] = 0.0
    gt[..., 1::4, 2::7] = -3.0             # negative ground truth is invalid too (gt > 0)
  elif invalid == "all":
    gt = -gt.abs()
  ref = orc.khamis_robust_loss(pred, gt)
  pd = pred.detach().to(DEV).requires_grad_(True)
  got = khamis_robust_loss(pd, gt.to(DEV))
  assert got.dim() == 0
  assert abs(float(got) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref))), (float(got), float(ref))
  (got * 0.7).backward()
  if invalid == "all":
    assert float(got) == 0.0 and float(pd.grad.abs().max()) == 0.0
    return
  (ref * 0.7).backward()
  close(pd.grad, pred.grad, 1e-9, 2e-6, "khamis g_pred")
  assert float(pd.grad[gt.to(DEV) <= 0].abs().max() if bool((gt <= 0).any()) else 0.0) == 0.0


def test_khamis_robust_loss_matches_reference_fixture(golden_loader):
  """The reference's own value on the fixture's construction (tests/golden/make_golden.py: gt = pred + 0.5 with a grid
  of invalid pixels; the value does not depend on pred beyond rounding)."""
  from adaptive_stereo.utils.loss_functions import khamis_robust_loss
  gold = golden_loader("crop_96x256_k4_b1")
  pred = torch.from_numpy(gold.z["full__train/pred_refined"]).to(DEV)
  gt = (pred + 0.5).clone()
  gt[:, :, ::3, ::5] = 0.0
  assert abs(float(khamis_robust_loss(pred, gt)) - gold.scalar("train/khamis")) < 1e-6


# ----------------------------------------------------------------------------- a3, second generation (csrc/agg3d.hip)
AGG_CASES = [  # B, D, H, W
  (1, 12, 24, 78),      # KITTI k=4, one pair: segments of one plane
  (4, 12, 24, 78),      # the benchmark's launch: 240 units of three planes
  (2, 12, 34, 60),      # SceneFlow k=4
  (3, 5, 9, 40),        # ragged: D not a multiple of the segment length, last tile shifted back
  (1, 1, 4, 33),        # a single plane, the smallest supported width (Wp = 35 >= 34)
  (2, 7, 6, 81),        # the widest plane whose four runs fit in LDS as they are
  (1, 6, 11, 156),      # KITTI k=3 width: two column strips of 78 (+2) columns
  (2, 5, 9, 120),       # SceneFlow k=3 width: two strips of 62 (+2), the second right-aligned and overlapping the first
  (1, 4, 7, 200),       # three strips
  (1, 3, 47, 156),      # KITTI k=3 plane, full height
  (6, 7, 24, 78),       # segments of four planes, the last one ragged (planes 4..6)
]


@pytest.mark.parametrize("B,D,H,W", AGG_CASES)
def test_agg3d_layer_against_torch_and_first_generation(B, D, H, W):
  """as_agg3d_fwd, every flavour: raw output + BatchNorm partials (training forward of layer 1), the previous layer's
  BatchNorm merged from its partials by the consumer and applied with the LeakyReLU to the operand in LDS, the activated
  tensor written back (layers 2-4), folded BatchNorm epilogue (eval) and raw data gradient — against torch's conv3d / batch_norm on the CPU and, bit for
  bit, against the first-generation kernels (same fp32 accumulation order by construction)."""
  g = Pcl(B, D, H, W, 1, 1, 1)
  lib = nat.load()
  assert lib.as_agg3d_ok(g) == 1
  shape = ops.CONV3D_333
  x = rnd(B, 32, D, H, W, seed=1)
  w = rnd(32, 32, 3, 3, 3, seed=2, scale=1.0 / 864 ** 0.5)
  b = rnd(32, seed=3, scale=0.1)
  gamma, beta = rnd(32, seed=5) * 0.5 + 1.0, rnd(32, seed=6) * 0.2
  rm, rv = rnd(32, seed=7) * 0.1, rnd(32, seed=8).abs() + 0.5
  z_ref = F.conv3d(x, w, b, padding=1)
  xb = ops.ncdhw_to_pcl(x.to(DEV), g)
  wd, bd = w.to(DEV), b.to(DEV)
  wp, wpt = ops.pack_weights(wd, shape, False), ops.pack_weights(wd, shape, True)
  gd, bed = gamma.to(DEV), beta.to(DEV)
  tag = "B%d D%d H%d W%d" % (B, D, H, W)

  def halo_is_zero(buf, what):
    full = ops.pcl_view(buf, g).clone(); ops.pcl_interior(full, g).zero_()
    assert float(full.abs().max()) == 0.0, "%s: %s wrote into the halo" % (tag, what)

  prev = ops.set_agg3d(False)
  try:
    old_stats = ops.conv32_stat_parts(g, g, shape, DEV)
    z_old = ops.conv32(xb, g, wp, bd, g, shape, stats=old_stats)
    rm_o, rv_o = rm.to(DEV).clone(), rv.to(DEV).clone()
    st_old = ops.bn_train_stats(old_stats, gd, bed, rm_o, rv_o)
  finally:
    ops.set_agg3d(prev)

  # -- training forward of a first layer: raw output + BatchNorm partials ------------------------------------------
  nparts = lib.as_agg3d_parts(g)
  z1 = ops.pcl_zeros(g, DEV)
  rm_d, rv_d = rm.to(DEV).clone(), rv.to(DEV).clone()
  pend = ops.PendingBn(ops.StatParts(nparts, DEV), gd, bed, rm_d, rv_d)
  ops.agg3d(xb, g, wp, bd, z=z1, stats=pend.stats)
  close(ops.pcl_to_ncdhw(z1, g), z_ref, 2e-5, 1e-5, tag + " fwd")
  assert torch.equal(z1, z_old), tag + ": not bit-identical to conv3d_lds_kernel"
  halo_is_zero(z1, "forward")
  assert abs(float(pend.stats.cnt.sum()) - B * D * H * W) < 0.5, "the partials must count every voxel exactly once"

  # -- layers 2-4: the operand is the previous layer's raw output; its BatchNorm is merged from the partials by every
  #    workgroup (workgroup 0 publishes the state and the running statistics), applied in LDS; a_out is a by-product --
  a_new, z2 = ops.pcl_zeros(g, DEV), ops.pcl_zeros(g, DEV)
  ops.agg3d(z1, g, wp, bd, z=z2, in_bn=pend, a_out=a_new, stats=ops.StatParts(nparts, DEV))
  st = pend.state
  mean_ref = z_ref.mean(dim=(0, 2, 3, 4)); var_ref = z_ref.var(dim=(0, 2, 3, 4), unbiased=False)
  close(st.mean, mean_ref, 2e-6, 1e-5, tag + " bn mean")
  close(st.invstd, 1.0 / torch.sqrt(var_ref + 1e-5), 0, 2e-5, tag + " bn invstd")
  close(st.scale, gamma / torch.sqrt(var_ref + 1e-5), 0, 2e-5, tag + " bn scale")
  close(st.shift, beta - mean_ref * gamma / torch.sqrt(var_ref + 1e-5), 3e-6, 3e-5, tag + " bn shift")
  rm_ref, rv_ref = rm.clone(), rv.clone()
  F.batch_norm(z_ref, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)
  close(rm_d, rm_ref, 2e-6, 1e-5, tag + " running_mean"); close(rv_d, rv_ref, 2e-6, 2e-5, tag + " running_var")
  close(st.mean, st_old.mean, 1e-6, 1e-6, tag + " mean vs first generation")
  close(st.invstd, st_old.invstd, 0, 2e-6, tag + " invstd vs first generation")
  # the stand-alone finalize launch on the same partials gives the same state (merge order differs: fp64, then rounded)
  pend2 = ops.PendingBn(pend.stats, gd, bed, rm.to(DEV).clone(), rv.to(DEV).clone())
  st2 = pend2.finalize()
  close(st2.scale, st.scale, 0, 2e-7, tag + " scale vs as_bn_finalize"); close(st2.shift, st.shift, 1e-7, 2e-7, tag + " shift vs as_bn_finalize")
  close(pend2.rm, rm_d, 1e-8, 2e-7, tag + " running_mean vs as_bn_finalize")
  # first generation on the state the kernel published: element-wise pass, then the convolution
  a_old = ops.bn_act(z1, st, g)
  prev = ops.set_agg3d(False)
  try:
    z2_old = ops.conv32(a_old, g, wp, bd, g, shape)
  finally:
    ops.set_agg3d(prev)
  assert torch.equal(a_new, a_old), tag + ": activated by-product differs from as_bn_act_fwd"
  assert torch.equal(z2, z2_old), tag + ": fused-operand layer differs from the two-pass result"
  halo_is_zero(a_new, "by-product"); halo_is_zero(z2, "fused layer")
  a_ref = F.leaky_relu(z_ref * st.scale.cpu().view(1, -1, 1, 1, 1) + st.shift.cpu().view(1, -1, 1, 1, 1), 0.2)
  close(ops.pcl_to_ncdhw(z2, g), F.conv3d(a_ref, w, b, padding=1), 5e-5, 1e-5, tag + " fused layer vs torch")
  # the same layer with the affine handed over finalized, without by-product and moments; and a second merge of the same
  # partials (fixed order: same bits)
  z2n = ops.pcl_zeros(g, DEV)
  ops.agg3d(z1, g, wp, bd, z=z2n, in_state=st)
  assert torch.equal(z2n, z2)
  pend3 = ops.PendingBn(pend.stats, gd, bed, rm.to(DEV).clone(), rv.to(DEV).clone())
  z2m = ops.pcl_zeros(g, DEV)
  ops.agg3d(z1, g, wp, bd, z=z2m, in_bn=pend3)
  assert torch.equal(z2m, z2) and torch.equal(pend3.state.scale, st.scale) and torch.equal(pend3.rm, rm_d)

  # -- eval forward (BatchNorm folded into the epilogue) and the data gradient --------------------------------------
  prev = ops.set_agg3d(False)
  try:
    y_old = ops.conv32(xb, g, wp, bd, g, shape, epilogue=1, scale=st.scale, shift=st.shift)
    gx_old = ops.conv32(xb, g, wpt, None, g, shape)
  finally:
    ops.set_agg3d(prev)
  y_new = ops.agg3d(xb, g, wp, bd, z=ops.pcl_zeros(g, DEV), epilogue=1, ep_state=st)
  assert torch.equal(y_new, y_old), tag + ": eval epilogue"
  halo_is_zero(y_new, "eval epilogue")
  gx_new = ops.agg3d(xb, g, wpt, None, z=ops.pcl_zeros(g, DEV), epilogue=2)
  assert torch.equal(gx_new, gx_old), tag + ": data gradient"
  halo_is_zero(gx_new, "data gradient")


def test_agg3d_refuses_unsupported_geometry():
  lib = nat.load()
  assert lib.as_agg3d_ok(Pcl(1, 12, 24, 31, 1, 1, 1)) == 0      # Wp = 33 < 34
  assert lib.as_agg3d_ok(Pcl(1, 12, 24, 90, 1, 1, 1)) == 1      # wider than four LDS runs: two column strips (round 3)
  assert lib.as_agg3d_ok(Pcl(1, 24, 47, 156, 1, 1, 1)) == 1     # KITTI k=3: two strips of 78 columns
  assert lib.as_agg3d_ok(Pcl(1, 24, 68, 120, 1, 1, 1)) == 1     # SceneFlow k=3
  assert lib.as_agg3d_ok(Pcl(1, 12, 24, 78, 1, 2, 2)) == 0      # halo must be exactly 1
  assert lib.as_agg3d_ok(Pcl(1, 12, 900, 78, 1, 1, 1)) == 0     # positions of a (sub-)plane must stay below 2^16
  g = Pcl(1, 12, 24, 31, 1, 1, 1)
  x = ops.pcl_zeros(g, DEV)
  with pytest.raises(RuntimeError, match="not supported"):
    ops.agg3d(x, g, torch.zeros(27 * 1024, device=DEV), None, z=ops.pcl_zeros(g, DEV), epilogue=2)


# ----------------------------------------------------------------------------- a4 + a5 + a8 fused (csrc/agg_tail.hip)
def _tail(x_pcl, g, w, b, in_state=None, a_out=None, in_bn=None):
  B, D, H, W = g.B, g.D, g.H, g.W
  logits = torch.full((B, D, H, W), float("nan"), device=DEV)
  pred = torch.full((B, H, W), float("nan"), device=DEV)
  am = torch.full((B, H, W), -1, dtype=torch.int32, device=DEV)
  fcs = torch.full((B, H, W), float("nan"), device=DEV)
  nat.call("as_agg_tail_fwd", nat.ptr(x_pcl), g, nat.ptr(in_state.scale) if in_state is not None else None,
           nat.ptr(in_state.shift) if in_state is not None else None, in_bn.block if in_bn is not None else None,
           nat.ptr(a_out), nat.ptr(w), nat.ptr(b), 0.2,
           nat.ptr(logits), nat.ptr(pred), nat.ptr(am), nat.ptr(fcs), nat.stream())
  return logits, pred, am, fcs


@pytest.mark.parametrize("B,D,H,W,gain", [(1, 8, 5, 9, 1.0), (2, 12, 6, 19, 50.0), (1, 24, 4, 33, 300.0), (4, 12, 24, 78, 20.0),
                                          (1, 24, 47, 156, 20.0), (2, 1, 3, 40, 1.0), (1, 2, 2, 31, 5.0), (1, 32, 3, 17, 100.0)])
def test_agg_tail_logits_softargmax_argmax_fcs(B, D, H, W, gain):
  """conv3d_alone -> logits -> soft-argmax / arg-max / FCS in one launch (as_agg_tail_fwd) against torch on the CPU and
  against the two-launch path (as_conv3d_out_fwd + as_softargmax_fwd); with the last layer's BatchNorm + LeakyReLU applied
  on the way in (training forward) against as_bn_act_fwd followed by the plain flavour, bit for bit.  Arg-max indices:
  equal to torch.argmax of the kernel's OWN logits everywhere (the index logic is exact) and equal to the reference's
  wherever its top-2 gap exceeds twice the logit deviation (all pixels, in practice)."""
  g = Pcl(B, D, H, W, 1, 1, 1)
  assert nat.load().as_agg_tail_ok(g) == 1
  a = rnd(B, 32, D, H, W, seed=1)
  w = rnd(1, 32, 3, 3, 3, seed=2, scale=0.03) * gain
  b = rnd(1, seed=3, scale=0.1) * gain
  logits_ref = F.conv3d(a, w, b, padding=1).squeeze(1)
  ab = ops.ncdhw_to_pcl(a.to(DEV), g)
  wd, bd = w.to(DEV).contiguous(), b.to(DEV)
  logits, pred, am, fcs = _tail(ab, g, wd, bd)
  assert not bool(torch.isnan(logits).any() | torch.isnan(pred).any() | torch.isnan(fcs).any()) and int(am.min()) >= 0, \
      "every interior pixel must be written"
  close(logits, logits_ref, 3e-6 * max(1.0, gain), 1e-5, "logits")
  own = logits.cpu()
  assert torch.equal(am.cpu().long(), torch.argmax(own, dim=1)), "arg-max of the kernel's own logits"
  srt = torch.sort(logits_ref, dim=1, descending=True)[0]
  dev_l = float((own - logits_ref).abs().max())
  decided = (srt[:, 0] - srt[:, 1]) > 2 * dev_l if D > 1 else torch.ones(B, H, W, dtype=torch.bool)
  assert bool((am.cpu().long()[decided] == torch.argmax(logits_ref, dim=1)[decided]).all())
  close(pred, orc.soft_argmax(own), 2e-5, 1e-5, "soft-argmax of the kernel's own logits")
  close(pred, orc.soft_argmax(logits_ref), 2e-5 + 40.0 * dev_l, 1e-5, "soft-argmax")
  if D > 2:
    close(fcs, orc.feature_contrast_mean(own), 1e-5 * max(1.0, gain), 1e-5, "fcs")
  else:
    assert float(fcs.abs().max()) == 0.0
  # the two-launch path on the same input
  l2 = torch.empty(B, D, H, W, device=DEV)
  nat.call("as_conv3d_out_fwd", nat.ptr(ab), g, nat.ptr(wd), nat.ptr(bd), nat.ptr(l2), nat.stream())
  close(logits, l2, 3e-6 * max(1.0, gain), 1e-5, "logits vs conv32to1_fwd_kernel")
  # training forward: raw input + BatchNorm affine + LeakyReLU on the way in, activated tensor as a by-product
  st = ops.BnState(DEV)
  st.scale.copy_((rnd(32, seed=11) * 0.5 + 1.0).to(DEV)); st.shift.copy_((rnd(32, seed=12) * 0.3).to(DEV))
  act = ops.bn_act(ab, st, g)
  l_two, p_two, am_two, f_two = _tail(act, g, wd, bd)
  a_out = ops.pcl_zeros(g, DEV)
  l_one, p_one, am_one, f_one = _tail(ab, g, wd, bd, in_state=st, a_out=a_out)
  assert torch.equal(a_out, act), "activated by-product differs from as_bn_act_fwd"
  assert torch.equal(l_one, l_two) and torch.equal(p_one, p_two) and torch.equal(am_one, am_two) and torch.equal(f_one, f_two)
  l_nobp = _tail(ab, g, wd, bd, in_state=st)[0]
  assert torch.equal(l_nobp, l_one)
  # ... and with the BatchNorm still in partials (merged by every workgroup): against as_bn_finalize + the affine flavour
  parts = ops.StatParts(37, DEV)
  parts.mean.copy_((rnd(37 * 32, seed=21) * 0.3).to(DEV)); parts.m2.copy_((rnd(37 * 32, seed=22).abs() * 50 + 5).to(DEV))
  parts.cnt.copy_((rnd(37, seed=23).abs() * 90 + 10).floor().to(DEV)); parts.cnt[5] = 0.0; parts.m2[5 * 32:6 * 32] = 0.0
  gam, bet = (rnd(32, seed=24) * 0.5 + 1.0).to(DEV), (rnd(32, seed=25) * 0.2).to(DEV)
  pa = ops.PendingBn(parts, gam, bet, torch.zeros(32, device=DEV), torch.ones(32, device=DEV))
  pb = ops.PendingBn(parts, gam, bet, torch.zeros(32, device=DEV), torch.ones(32, device=DEV))
  a_m = ops.pcl_zeros(g, DEV)
  l_m, p_m, am_m, f_m = _tail(ab, g, wd, bd, in_bn=pa, a_out=a_m)
  st_f = pb.finalize()
  close(pa.state.scale, st_f.scale, 0, 2e-7, "merged scale"); close(pa.state.shift, st_f.shift, 1e-7, 2e-7, "merged shift")
  close(pa.rm, pb.rm, 1e-8, 2e-7, "merged running_mean"); close(pa.rv, pb.rv, 1e-8, 2e-7, "merged running_var")
  l_a, p_a, am_a, f_a = _tail(ab, g, wd, bd, in_state=pa.state)
  assert torch.equal(l_m, l_a) and torch.equal(p_m, p_a) and torch.equal(am_m, am_a) and torch.equal(f_m, f_a)
  assert torch.equal(a_m, ops.bn_act(ab, pa.state, g))


def test_agg_tail_ties_and_extremes():
  """First maximum wins (torch.argmax semantics), duplicates of the maximum count twice in the FCS (as in a sort), large
  logits do not overflow: designed logits are driven through the convolution exactly (centre tap, one channel)."""
  D, H, W = 6, 2, 40
  l = torch.zeros(1, D, H, W)
  l[0, :, 0, 1] = torch.tensor([1.0, 5.0, 5.0, 0.0, 5.0, -1.0])      # three-way tie -> index 1
  l[0, :, 0, 2] = torch.tensor([-300.0, 200.0, -50.0, 199.0, 0.0, 10.0])
  l[0, :, 0, 3] = torch.tensor([3e4, -3e4, 0.0, 0.0, 0.0, 0.0])
  l[0, :, 1, 39] = torch.tensor([0.0, 0.0, 0.0, 0.0, 0.0, 7.0])     # last pixel of the plane, maximum at the last index
  l[0, :, 1, 0] = torch.tensor([2.0, 2.0, 2.0, 2.0, 2.0, 2.0])       # all equal -> index 0
  a = torch.zeros(1, 32, D, H, W); a[0, 0] = l[0]
  w = torch.zeros(1, 32, 3, 3, 3); w[0, 0, 1, 1, 1] = 1.0
  g = Pcl(1, D, H, W, 1, 1, 1)
  logits, pred, am, fcs = _tail(ops.ncdhw_to_pcl(a.to(DEV), g), g, w.to(DEV), None)
  assert torch.equal(logits.cpu(), l)
  assert torch.equal(am.cpu().long(), torch.argmax(l, dim=1))
  assert int(am[0, 0, 1]) == 1 and int(am[0, 1, 39]) == 5 and int(am[0, 1, 0]) == 0
  assert bool(torch.isfinite(pred).all())
  close(pred, orc.soft_argmax(l), 2e-5, 1e-5, "soft-argmax")
  close(fcs, orc.feature_contrast_mean(l), 1e-5, 1e-5, "fcs")


# ----------------------------------------------------------------------------- a7 backward in one launch (csrc/conv32_bwd.hip)
@pytest.mark.parametrize("B,H,W,dil", [(2, 160, 1242, 1), (2, 161, 1242, 2), (1, 375, 1030, 4), (2, 163, 1237, 8)])
def test_conv32_backward_fused_in_one_launch(B, H, W, dil):
  """as_conv32_bwd_fused — stage 3 of the layer's BatchNorm backward, data gradient + skip connection, weight / bias
  gradient and stage 1 of the next BatchNorm backward from ONE staged copy — against the two-launch path
  (as_conv32_wgrad_bnapply, then as_conv32_fwd_bnbwd): g_x bit for bit where the arithmetic is the same chain, the weight
  gradient and the per-channel sums to summation order.  Ragged last segment, every dilation, comb residues that do not
  divide H."""
  g = Pcl(B, 1, H, W, 0, 8, 8)
  shape = ops.conv_shape_2d(dil)
  lib = nat.load()
  assert lib.as_conv32_bwd_fused_ok(g, g, shape) == 1
  x = ops.ncdhw_to_pcl(rnd(B, 32, 1, H, W, seed=1).to(DEV), g)
  g_a = ops.ncdhw_to_pcl(rnd(B, 32, 1, H, W, seed=2).to(DEV), g)
  z = ops.ncdhw_to_pcl(rnd(B, 32, 1, H, W, seed=3).to(DEV), g)
  zn = ops.ncdhw_to_pcl(rnd(B, 32, 1, H, W, seed=4).to(DEV), g)
  w = (rnd(32, 32, 3, 3, seed=9) * 0.06).to(DEV)
  wp_t = ops.pack_weights(w, shape, True)

  def state(seed):
    st = ops.BnState(DEV)
    st.mean.copy_(rnd(32, seed=seed).to(DEV) * 0.1); st.invstd.copy_(rnd(32, seed=seed + 1).abs().to(DEV) + 0.5)
    gamma = (rnd(32, seed=seed + 2).abs() + 0.5).to(DEV)
    st.scale.copy_(st.invstd * gamma); st.shift.copy_(rnd(32, seed=seed + 3).to(DEV) * 0.1 - st.mean * st.scale)
    return st, gamma
  st, gamma = state(5)
  stn, gamman = state(15)
  # stages 1-2 of this layer's BatchNorm backward: coefficients for stage 3
  ws = torch.empty(lib.as_bn_bwd_workspace(g), device=DEV)
  gg, gb = torch.zeros(32, device=DEV), torch.zeros(32, device=DEV)
  nat.call("as_bn_act_bwd", nat.ptr(g_a), nat.ptr(z), nat.ptr(st.scale), nat.ptr(st.shift), nat.ptr(st.mean),
           nat.ptr(st.invstd), nat.ptr(gamma), 0.2, 1, None, nat.ptr(gg), nat.ptr(gb), 0, nat.ptr(ws), g, nat.stream())
  coef = ws[lib.as_bn_bwd_coef_offset():]
  # two launches
  gz = ops.pcl_zeros(g, DEV)
  dW_ref = torch.zeros(32, 32, 3, 3, device=DEV); db_ref = torch.zeros(32, device=DEV)
  wws = torch.empty(lib.as_conv32_wgrad_workspace(g, g, shape), device=DEV)
  nat.call("as_conv32_wgrad_bnapply", nat.ptr(x), g, nat.ptr(g_a), nat.ptr(z), g, shape, nat.ptr(st.scale), nat.ptr(st.shift),
           nat.ptr(st.mean), nat.ptr(coef), 0.2, nat.ptr(gz), nat.ptr(dW_ref), nat.ptr(db_ref), 0, nat.ptr(wws), nat.stream())
  gx_ref, sums_ref = ops.conv32_dgrad_bnbwd(gz, g, wp_t, shape, g_a, zn, stn)
  _, ggn_ref, gbn_ref = ops.bn_act_bwd(gx_ref, zn, stn, gamman, g, True, sums=sums_ref)
  # one launch
  gx = ops.pcl_zeros(g, DEV)
  dW = torch.zeros(32, 32, 3, 3, device=DEV); db = torch.zeros(32, device=DEV)
  nws = torch.empty(lib.as_bn_bwd_workspace(g), device=DEV)
  fws = torch.empty(lib.as_conv32_bwd_fused_workspace(), device=DEV)
  nat.call("as_conv32_bwd_fused", nat.ptr(x), g, nat.ptr(g_a), nat.ptr(z), g, shape, nat.ptr(wp_t), nat.ptr(st.scale),
           nat.ptr(st.shift), nat.ptr(st.mean), nat.ptr(coef), 0.2, nat.ptr(zn), nat.ptr(stn.scale), nat.ptr(stn.shift),
           nat.ptr(stn.mean), nat.ptr(gx), nat.ptr(dW), nat.ptr(db), 0, nat.ptr(nws), nat.ptr(fws), nat.stream())
  tag = "B%d H%d W%d d%d" % (B, H, W, dil)
  full = ops.pcl_view(gx, g).clone(); ops.pcl_interior(full, g).zero_()
  assert float(full.abs().max()) == 0.0, tag + ": wrote into the halo"
  gi, gi_ref = ops.pcl_interior(ops.pcl_view(gx, g), g), ops.pcl_interior(ops.pcl_view(gx_ref, g), g)
  close(gi, gi_ref, 2e-6, 1e-6, tag + " g_x")
  exact = bool(torch.equal(gi, gi_ref))
  n = B * H * W
  for name, got, exp in ((" dW", dW, dW_ref), (" db", db, db_ref)):
    rel = float((got.double() - exp.double()).norm() / exp.double().norm())
    assert rel < 2e-5, "%s%s: relative L2 error %.2e" % (tag, name, rel)
  _, ggn, gbn = ops.bn_act_bwd(gx, zn, stn, gamman, g, True, sums=ops.BnBwdSums(nws, lib.as_conv32_bwd_fused_parts()))
  close(ggn, ggn_ref, 2e-6 * n ** 0.5 * float(ggn_ref.abs().max()) + 1e-5, 1e-5, tag + " next g_gamma")
  close(gbn, gbn_ref, 2e-6 * n ** 0.5 * float(gbn_ref.abs().max()) + 1e-5, 1e-5, tag + " next g_beta")
  # accumulate flavour: a second launch adds the same gradient again
  nat.call("as_conv32_bwd_fused", nat.ptr(x), g, nat.ptr(g_a), nat.ptr(z), g, shape, nat.ptr(wp_t), nat.ptr(st.scale),
           nat.ptr(st.shift), nat.ptr(st.mean), nat.ptr(coef), 0.2, nat.ptr(zn), nat.ptr(stn.scale), nat.ptr(stn.shift),
           nat.ptr(stn.mean), nat.ptr(gx), nat.ptr(dW), nat.ptr(db), 1, nat.ptr(nws), nat.ptr(fws), nat.stream())
  close(dW, 2 * dW_ref, 1e-4 * float(dW_ref.abs().max()), 1e-4, tag + " accumulated dW")
  from conftest import parity_note
  parity_note("bwd_fused[%s]" % tag, g_x_bit_identical=exact)

def _wino_edge_cases():
  """Geometries around every edge of the minimal-filtering kernels' tiling: W = 64 k + {0, 1, 63} (no shifted segment; a
  neighbour that keeps ONE column; one that keeps 63), H = 2 d and 2 d + 1 (combs of two rows and of one pair + a lone row),
  H odd / even, every dilation — with the batch chosen so that as_conv32_wino_ok accepts the launch."""
  cases = []
  for dil in (1, 2, 4, 8):
    for H, W in ((2 * dil, 129), (2 * dil + 1, 64), (37, 65), (50, 127), (51, 192), (23 + dil, 191)):
      if H < 2 * dil:
        continue
      pairs = sum(((H - r + dil - 1) // dil + 1) // 2 for r in range(dil))
      B = -(-2048 // (((W + 63) // 64) * pairs))
      if B * H * W * 32 * 4 <= 600e6:
        cases.append((B, H, W, dil))
  return cases


@pytest.mark.parametrize("B,H,W,dil", _wino_edge_cases())
def test_minimal_filtering_kernels_on_the_edges_of_their_tiling(B, H, W, dil):
  """Forward (with skip input), inference block, data gradient and weight gradient by minimal filtering against the direct
  kernels on small odd geometries: every output within 3e-6 of the direct result's scale (both are fp32 roundings of the same sums),
  nothing written into the halo, element counts exact."""
  g = Pcl(B, 1, H, W, 0, 8, 8)
  shape = ops.conv_shape_2d(dil)
  lib = nat.load()
  assert lib.as_conv32_wino_ok(g, g, shape) == 1
  T = lambda seed: ops.ncdhw_to_pcl(rnd(B, 32, 1, H, W, seed=seed).to(DEV), g)
  z_prev, a_pp, x, g_a, zz, zn = T(1), T(2), T(3), T(4), T(5), T(6)
  w = (rnd(32, 32, 3, 3, seed=9) * 0.06).to(DEV)
  b = (rnd(32, seed=10) * 0.1).to(DEV)
  wp, wp_t = ops.pack_weights(w, shape, False), ops.pack_weights(w, shape, True)
  ww, ww_t = torch.empty(16 * 1024, device=DEV), torch.empty(16 * 1024, device=DEV)
  nat.call("as_conv32_wino_pack_weights", nat.ptr(w), nat.ptr(ww), 0, nat.stream())
  nat.call("as_conv32_wino_pack_weights", nat.ptr(w), nat.ptr(ww_t), 1, nat.stream())
  st = ops.BnState(DEV)
  st.mean.copy_(rnd(32, seed=5).to(DEV) * 0.1); st.invstd.copy_(rnd(32, seed=6).abs().to(DEV) + 0.5)
  st.scale.copy_(st.invstd * 1.1); st.shift.copy_(rnd(32, seed=7).to(DEV) * 0.1 - st.mean * st.scale)
  tag = "edge B%d H%d W%d d%d" % (B, H, W, dil)
  def same(got, exp, what):
    for buf in (got,):
      full = ops.pcl_view(buf, g).clone(); ops.pcl_interior(full, g).zero_()
      assert float(full.abs().max()) == 0.0, "%s: %s written into the halo" % (tag, what)
    gi, ei = ops.pcl_interior(ops.pcl_view(got, g), g), ops.pcl_interior(ops.pcl_view(exp, g), g)
    err, scale = float((gi - ei).abs().max()), float(ei.abs().max())
    assert err <= 3e-6 * scale, "%s: %s differs by %.3e (scale %.3e)" % (tag, what, err, scale)
  # forward (training, with skip input)
  a_ref = ops.bn_act(z_prev, st, g, residual=a_pp, out=ops.pcl_zeros(g, DEV))
  z_ref = ops.conv32(a_ref, g, wp, b, g, shape, out=ops.pcl_zeros(g, DEV))
  a_out, z = ops.pcl_zeros(g, DEV), ops.pcl_zeros(g, DEV)
  stats = ops.StatParts(lib.as_conv32_wino_parts(), DEV)
  nat.call("as_conv32_wino_fwd", nat.ptr(z_prev), nat.ptr(a_pp), nat.ptr(st.scale), nat.ptr(st.shift), nat.ptr(a_out), g,
           nat.ptr(ww), nat.ptr(b), 0.2, nat.ptr(z), g, shape, nat.ptr(stats.mean), nat.ptr(stats.m2), nat.ptr(stats.cnt),
           nat.stream())
  assert bool(torch.equal(a_out, a_ref)), tag + ": by-product"
  same(z, z_ref, "forward z")
  assert float(stats.cnt.sum()) == float(B * H * W), tag + ": moments count"
  # inference block
  e_ref = ops.conv32(x, g, wp, b, g, shape, out=ops.pcl_zeros(g, DEV), epilogue=1, scale=st.scale, shift=st.shift, residual=x)
  e_out = ops.pcl_zeros(g, DEV)
  nat.call("as_conv32_wino_eval", nat.ptr(x), g, shape, nat.ptr(ww), nat.ptr(b), nat.ptr(st.scale), nat.ptr(st.shift), 0.2, 1,
           nat.ptr(e_out), nat.stream())
  same(e_out, e_ref, "inference block")
  # backward
  ws = torch.empty(lib.as_bn_bwd_workspace(g), device=DEV)
  gg, gb = torch.zeros(32, device=DEV), torch.zeros(32, device=DEV)
  gamma = torch.full((32,), 1.1, device=DEV)
  nat.call("as_bn_act_bwd", nat.ptr(g_a), nat.ptr(zz), nat.ptr(st.scale), nat.ptr(st.shift), nat.ptr(st.mean),
           nat.ptr(st.invstd), nat.ptr(gamma), 0.2, 1, None, nat.ptr(gg), nat.ptr(gb), 0, nat.ptr(ws), g, nat.stream())
  coef = ws[lib.as_bn_bwd_coef_offset():]
  gz, gx = ops.pcl_zeros(g, DEV), ops.pcl_zeros(g, DEV)
  dW, db = torch.zeros(32, 32, 3, 3, device=DEV), torch.zeros(32, device=DEV)
  nws = torch.empty(lib.as_bn_bwd_workspace(g), device=DEV)
  fws = torch.empty(lib.as_conv32_wino_bwd_workspace(), device=DEV)
  nat.call("as_conv32_wino_bwd", nat.ptr(x), g, nat.ptr(g_a), nat.ptr(zz), g, shape, nat.ptr(ww_t), nat.ptr(st.scale),
           nat.ptr(st.shift), nat.ptr(st.mean), nat.ptr(coef), 0.2, nat.ptr(zn), nat.ptr(st.scale), nat.ptr(st.shift),
           nat.ptr(st.mean), nat.ptr(gz), nat.ptr(gx), nat.ptr(dW), nat.ptr(db), 0, nat.ptr(nws), nat.ptr(fws), nat.stream())
  # references from g_z itself: direct data gradient + skip, autograd-free weight gradient on the CPU in fp64
  gx_ref = ops.conv32(gz, g, wp_t, None, g, shape, out=ops.pcl_zeros(g, DEV), residual=g_a)
  same(gx, gx_ref, "data gradient")
  gz64 = ops.pcl_to_ncdhw(gz, g)[:, :, 0].double().cpu(); x64 = ops.pcl_to_ncdhw(x, g)[:, :, 0].double().cpu()
  dW64 = torch.nn.grad.conv2d_weight(x64, tuple(w.shape), gz64, padding=dil, dilation=dil)
  assert float((dW.double().cpu() - dW64).norm() / dW64.norm()) < 3e-6, tag + ": weight gradient"
  # (a sum of ~1e5 signed terms: its fp32 rounding scales with the sum of magnitudes, not with the result)
  assert float((db.double().cpu() - gz64.sum(dim=(0, 2, 3))).abs().max()) <= 5e-7 * float(gz64.abs().sum(dim=(0, 2, 3)).max()), \
      tag + ": bias gradient"
  # g_z itself: stage 3 of the BatchNorm backward, element-wise in fp32
  k1, k2, k3 = coef[:32].view(1, 32, 1, 1), coef[32:64].view(1, 32, 1, 1), coef[64:96].view(1, 32, 1, 1)
  zz_n = ops.pcl_to_ncdhw(zz, g)[:, :, 0]; ga_n = ops.pcl_to_ncdhw(g_a, g)[:, :, 0]
  yy = zz_n * st.scale.view(1, 32, 1, 1) + st.shift.view(1, 32, 1, 1)
  gy = torch.where(yy > 0, ga_n, ga_n * 0.2)
  gz_exp = (gy - k1 - (zz_n - st.mean.view(1, 32, 1, 1)) * k2) * k3
  assert float((ops.pcl_to_ncdhw(gz, g)[:, :, 0] - gz_exp).abs().max()) <= 2e-6 * float(gz_exp.abs().max()), tag + ": g_z"


@pytest.mark.parametrize("B,H,W,dil,skip", [(2, 160, 1242, 1, True), (1, 375, 1242, 2, True), (2, 161, 1030, 4, True),
                                            (2, 163, 1237, 8, True), (4, 97, 700, 1, False)])
def test_conv32_eval_block_by_minimal_filtering(B, H, W, dil, skip):
  """as_conv32_wino_eval — an eval-mode BasicBlock in one launch, out = lrelu((conv(x) + bias) * scale + shift) (+ x) with the
  skip connection read from the staged rows — against as_conv32_fwd's fused epilogue (direct form): both held against the fp64
  result of the same fp32 operands, minimal filtering not further away than 2x the direct kernel."""
  g = Pcl(B, 1, H, W, 0, 8, 8)
  shape = ops.conv_shape_2d(dil)
  lib = nat.load()
  assert lib.as_conv32_wino_ok(g, g, shape) == 1
  x = ops.ncdhw_to_pcl(rnd(B, 32, 1, H, W, seed=1).to(DEV), g)
  w = (rnd(32, 32, 3, 3, seed=9) * 0.06).to(DEV)
  b = (rnd(32, seed=10) * 0.1).to(DEV)
  st = ops.BnState(DEV)
  st.scale.copy_(rnd(32, seed=5).abs().to(DEV) + 0.5); st.shift.copy_(rnd(32, seed=6).to(DEV) * 0.3)
  wp = ops.pack_weights(w, shape, False)
  ww = torch.empty(16 * 1024, device=DEV)
  nat.call("as_conv32_wino_pack_weights", nat.ptr(w), nat.ptr(ww), 0, nat.stream())
  ref = ops.conv32(x, g, wp, b, g, shape, out=ops.pcl_zeros(g, DEV), epilogue=1, scale=st.scale, shift=st.shift,
                   residual=x if skip else None)
  out = ops.pcl_zeros(g, DEV)
  nat.call("as_conv32_wino_eval", nat.ptr(x), g, shape, nat.ptr(ww), nat.ptr(b), nat.ptr(st.scale), nat.ptr(st.shift), 0.2,
           1 if skip else 0, nat.ptr(out), nat.stream())
  tag = "wino eval B%d H%d W%d d%d %s" % (B, H, W, dil, "skip" if skip else "plain")
  full = ops.pcl_view(out, g).clone(); ops.pcl_interior(full, g).zero_()
  assert float(full.abs().max()) == 0.0, tag + ": written into the halo"
  x64 = ops.pcl_to_ncdhw(x, g)[:, :, 0].double().cpu()
  y64 = torch.nn.functional.conv2d(x64, w.double().cpu(), b.double().cpu(), padding=dil, dilation=dil)
  y64 = y64 * st.scale.double().cpu().view(1, 32, 1, 1) + st.shift.double().cpu().view(1, 32, 1, 1)
  y64 = torch.where(y64 > 0, y64, 0.2 * y64) + (x64 if skip else 0.0)
  got = ops.pcl_to_ncdhw(out, g)[:, :, 0].double().cpu(); dire = ops.pcl_to_ncdhw(ref, g)[:, :, 0].double().cpu()
  e_w, e_d = float((got - y64).abs().max()), float((dire - y64).abs().max())
  r_w, r_d = float((got - y64).pow(2).mean().sqrt()), float((dire - y64).pow(2).mean().sqrt())
  scale = float(y64.abs().max())
  assert e_d <= 2e-6 * scale and e_w <= max(2.0 * e_d, 1e-6 * scale) and r_w <= 2.0 * r_d, (tag, e_w, e_d, r_w, r_d, scale)
  from conftest import parity_note
  parity_note("conv32_wino_eval[%s]" % tag, max_err_vs_fp64=e_w, direct_max_err_vs_fp64=e_d, rms_err_vs_fp64=r_w,
              direct_rms_err_vs_fp64=r_d)


def _dgrad_generation_cases():
  return _wino_edge_cases() + [(2, 160, 1242, 1), (2, 161, 1242, 2), (1, 375, 1030, 4), (2, 163, 1237, 8), (4, 375, 1242, 1),
                               (3, 375, 1242, 8), (12, 375, 64, 2), (5, 97, 700, 4)]


@pytest.mark.parametrize("B,H,W,dil", _dgrad_generation_cases())
def test_data_gradient_generations_write_the_same_bits(B, H, W, dil):
  """as_conv32_wino_bwd_data on csrc/conv32_wino_dgrad.hip (generation 2: the waves of a workgroup have roles, the skip
  connection's g_a rows wait in LDS between conversion and output, rows staged 64 + 2d voxels wide) against generation 1
  (csrc/conv32_wino.hip MODE 2, itself held against the direct kernels and fp64 below): g_z, g_x and the next BatchNorm's
  per-workgroup sums BIT FOR BIT — the same element-wise chains, the same order in every sum, the same owner for the columns
  the shifted last segment shares —, nothing written into a halo, two launches identical.  Every dilation, the small odd
  geometries of the tiling's edges, the KITTI size (4 pairs, the bench workload), single-segment images."""
  g = Pcl(B, 1, H, W, 0, 8, 8)
  shape = ops.conv_shape_2d(dil)
  lib = nat.load()
  assert lib.as_conv32_wino_ok(g, g, shape) == 1
  T = lambda seed: ops.ncdhw_to_pcl(rnd(B, 32, 1, H, W, seed=seed).to(DEV), g)
  g_a, zz, zn = T(4), T(5), T(6)
  w = (rnd(32, 32, 3, 3, seed=9) * 0.06).to(DEV)
  ww_t = torch.empty(16 * 1024, device=DEV)
  nat.call("as_conv32_wino_pack_weights", nat.ptr(w), nat.ptr(ww_t), 1, nat.stream())
  st, stn = ops.BnState(DEV), ops.BnState(DEV)
  for i, s_ in enumerate((st, stn)):
    s_.mean.copy_(rnd(32, seed=5 + 10 * i).to(DEV) * 0.1); s_.invstd.copy_(rnd(32, seed=6 + 10 * i).abs().to(DEV) + 0.5)
    s_.scale.copy_(s_.invstd * 1.1); s_.shift.copy_(rnd(32, seed=7 + 10 * i).to(DEV) * 0.1 - s_.mean * s_.scale)
  coef = (rnd(96, seed=21) * 0.05).to(DEV); coef[64:] = coef[64:].abs() + 0.7
  nparts = lib.as_conv32_wino_bwd_parts()
  out = {}
  prev = lib.as_conv32_wino_bwd_generation(0)
  try:
    for gen in (1, 2, 2):
      lib.as_conv32_wino_bwd_generation(3 if gen == 2 else 1)      # (3: the second generation for EVERY dilation, 8 included)
      gz, gx = ops.pcl_zeros(g, DEV), ops.pcl_zeros(g, DEV)
      nws = torch.zeros(lib.as_bn_bwd_workspace(g), device=DEV)
      nat.call("as_conv32_wino_bwd_data", nat.ptr(g_a), nat.ptr(zz), g, shape, nat.ptr(ww_t), nat.ptr(st.scale), nat.ptr(st.shift),
               nat.ptr(st.mean), nat.ptr(coef), 0.2, nat.ptr(zn), nat.ptr(stn.scale), nat.ptr(stn.shift), nat.ptr(stn.mean),
               nat.ptr(gz), nat.ptr(gx), nat.ptr(nws), nat.stream())
      torch.cuda.synchronize()
      out.setdefault(gen, []).append((gz, gx, nws[:nparts * 128].view(torch.int32).clone()))
  finally:
    lib.as_conv32_wino_bwd_generation(prev)
  tag = "dgrad generations B%d H%d W%d d%d" % (B, H, W, dil)
  (gz1, gx1, s1), (gz2, gx2, s2), (gz3, gx3, s3) = out[1][0], out[2][0], out[2][1]
  for name, a, b_ in (("g_z", gz1, gz2), ("g_x", gx1, gx2)):
    if not bool(torch.equal(a, b_)):
      va, vb = ops.pcl_view(a, g), ops.pcl_view(b_, g)
      bad = (va != vb).nonzero()
      raise AssertionError("%s: %s differs at %d elements; first [b, d, y, x, c] (padded) = %s: %r against %r" % (
          tag, name, bad.shape[0], bad[0].tolist(), float(va[tuple(bad[0])]), float(vb[tuple(bad[0])])))
  assert bool(torch.equal(s1, s2)), tag + ": next-BatchNorm partial sums differ"
  assert bool(torch.equal(gz2, gz3)) and bool(torch.equal(gx2, gx3)) and bool(torch.equal(s2, s3)), tag + ": two launches differ"
  assert float(ops.pcl_interior(ops.pcl_view(gx2, g), g).abs().max()) > 0.0


@pytest.mark.parametrize("B,H,W,dil", [(2, 160, 1242, 1), (2, 161, 1242, 2), (1, 375, 1030, 4), (2, 163, 1237, 8)])
def test_conv32_backward_by_minimal_filtering(B, H, W, dil):
  """as_conv32_wino_bwd — the backward of as_conv32_bwd_fused as a data gradient F(2x2, 3x3) and a weight gradient F(3x3, 2x2)
  — against the two-launch direct path (as_conv32_wgrad_bnapply, then as_conv32_fwd_bnbwd): g_z (stage 3 of the BatchNorm
  backward, the by-product) bit for bit; g_x and dW are different associations of the same sums, so they are held against the
  fp64 results computed from the SAME fp32 g_z, where they must not be further away than 2x the direct kernels are; db and
  the next BatchNorm's sums to summation order.  Ragged last segment, every dilation, odd comb lengths."""
  g = Pcl(B, 1, H, W, 0, 8, 8)
  shape = ops.conv_shape_2d(dil)
  lib = nat.load()
  assert lib.as_conv32_wino_ok(g, g, shape) == 1
  x = ops.ncdhw_to_pcl(rnd(B, 32, 1, H, W, seed=1).to(DEV), g)
  g_a = ops.ncdhw_to_pcl(rnd(B, 32, 1, H, W, seed=2).to(DEV), g)
  z = ops.ncdhw_to_pcl(rnd(B, 32, 1, H, W, seed=3).to(DEV), g)
  zn = ops.ncdhw_to_pcl(rnd(B, 32, 1, H, W, seed=4).to(DEV), g)
  w = (rnd(32, 32, 3, 3, seed=9) * 0.06).to(DEV)
  wp_t = ops.pack_weights(w, shape, True)
  ww_t = torch.empty(16 * 1024, device=DEV)
  nat.call("as_conv32_wino_pack_weights", nat.ptr(w), nat.ptr(ww_t), 1, nat.stream())

  def state(seed):
    st = ops.BnState(DEV)
    st.mean.copy_(rnd(32, seed=seed).to(DEV) * 0.1); st.invstd.copy_(rnd(32, seed=seed + 1).abs().to(DEV) + 0.5)
    gamma = (rnd(32, seed=seed + 2).abs() + 0.5).to(DEV)
    st.scale.copy_(st.invstd * gamma); st.shift.copy_(rnd(32, seed=seed + 3).to(DEV) * 0.1 - st.mean * st.scale)
    return st, gamma
  st, gamma = state(5)
  stn, gamman = state(15)
  ws = torch.empty(lib.as_bn_bwd_workspace(g), device=DEV)
  gg, gb = torch.zeros(32, device=DEV), torch.zeros(32, device=DEV)
  nat.call("as_bn_act_bwd", nat.ptr(g_a), nat.ptr(z), nat.ptr(st.scale), nat.ptr(st.shift), nat.ptr(st.mean),
           nat.ptr(st.invstd), nat.ptr(gamma), 0.2, 1, None, nat.ptr(gg), nat.ptr(gb), 0, nat.ptr(ws), g, nat.stream())
  coef = ws[lib.as_bn_bwd_coef_offset():]
  # the direct path
  gz_ref = ops.pcl_zeros(g, DEV)
  dW_ref = torch.zeros(32, 32, 3, 3, device=DEV); db_ref = torch.zeros(32, device=DEV)
  wws = torch.empty(lib.as_conv32_wgrad_workspace(g, g, shape), device=DEV)
  nat.call("as_conv32_wgrad_bnapply", nat.ptr(x), g, nat.ptr(g_a), nat.ptr(z), g, shape, nat.ptr(st.scale), nat.ptr(st.shift),
           nat.ptr(st.mean), nat.ptr(coef), 0.2, nat.ptr(gz_ref), nat.ptr(dW_ref), nat.ptr(db_ref), 0, nat.ptr(wws), nat.stream())
  gx_ref, sums_ref = ops.conv32_dgrad_bnbwd(gz_ref, g, wp_t, shape, g_a, zn, stn)
  _, ggn_ref, gbn_ref = ops.bn_act_bwd(gx_ref, zn, stn, gamman, g, True, sums=sums_ref)
  # minimal filtering
  gz, gx = ops.pcl_zeros(g, DEV), ops.pcl_zeros(g, DEV)
  dW = torch.zeros(32, 32, 3, 3, device=DEV); db = torch.zeros(32, device=DEV)
  nws = torch.empty(lib.as_bn_bwd_workspace(g), device=DEV)
  fws = torch.empty(lib.as_conv32_wino_bwd_workspace(), device=DEV)
  def run(acc):
    nat.call("as_conv32_wino_bwd", nat.ptr(x), g, nat.ptr(g_a), nat.ptr(z), g, shape, nat.ptr(ww_t), nat.ptr(st.scale),
             nat.ptr(st.shift), nat.ptr(st.mean), nat.ptr(coef), 0.2, nat.ptr(zn), nat.ptr(stn.scale), nat.ptr(stn.shift),
             nat.ptr(stn.mean), nat.ptr(gz), nat.ptr(gx), nat.ptr(dW), nat.ptr(db), acc, nat.ptr(nws), nat.ptr(fws), nat.stream())
  run(0)
  tag = "wino bwd B%d H%d W%d d%d" % (B, H, W, dil)
  for name, got in (("g_z", gz), ("g_x", gx)):
    full = ops.pcl_view(got, g).clone(); ops.pcl_interior(full, g).zero_()
    assert float(full.abs().max()) == 0.0, tag + ": %s written into the halo" % name
  assert bool(torch.equal(ops.pcl_interior(ops.pcl_view(gz, g), g), ops.pcl_interior(ops.pcl_view(gz_ref, g), g))), tag + " g_z"
  # fp64 from the same fp32 g_z (CPU)
  gz64 = ops.pcl_to_ncdhw(gz_ref, g)[:, :, 0].double().cpu()
  x64 = ops.pcl_to_ncdhw(x, g)[:, :, 0].double().cpu()
  ga64 = ops.pcl_to_ncdhw(g_a, g)[:, :, 0].double().cpu()
  w64 = w.double().cpu()
  gx64 = torch.nn.functional.conv_transpose2d(gz64, w64, padding=dil, dilation=dil) + ga64
  dW64 = torch.nn.grad.conv2d_weight(x64, w64.shape, gz64, padding=dil, dilation=dil)
  gx_got = ops.pcl_to_ncdhw(gx, g)[:, :, 0].double().cpu(); gx_dir = ops.pcl_to_ncdhw(gx_ref, g)[:, :, 0].double().cpu()
  e_w, e_d = float((gx_got - gx64).abs().max()), float((gx_dir - gx64).abs().max())
  r_w, r_d = float((gx_got - gx64).pow(2).mean().sqrt()), float((gx_dir - gx64).pow(2).mean().sqrt())
  scale = float(gx64.abs().max())
  assert e_d <= 2e-6 * scale and e_w <= max(2.0 * e_d, 1e-6 * scale) and r_w <= 2.0 * r_d, (tag, e_w, e_d, r_w, r_d, scale)
  dw_w = float((dW.double().cpu() - dW64).norm() / dW64.norm()); dw_d = float((dW_ref.double().cpu() - dW64).norm() / dW64.norm())
  assert dw_d < 2e-5 and dw_w <= max(2.0 * dw_d, 2e-6), (tag, dw_w, dw_d)
  db64 = gz64.sum(dim=(0, 2, 3))
  assert float((db.double().cpu() - db64).norm() / db64.norm()) < 2e-5
  n = B * H * W
  _, ggn, gbn = ops.bn_act_bwd(gx, zn, stn, gamman, g, True, sums=ops.BnBwdSums(nws, lib.as_conv32_wino_bwd_parts()))
  close(ggn, ggn_ref, 2e-6 * n ** 0.5 * float(ggn_ref.abs().max()) + 1e-5, 1e-5, tag + " next g_gamma")
  close(gbn, gbn_ref, 2e-6 * n ** 0.5 * float(gbn_ref.abs().max()) + 1e-5, 1e-5, tag + " next g_beta")
  # determinism: a second launch into fresh buffers gives the same bits (the shifted last segment must not race its neighbour)
  gx1, dW1, db1, nws1 = gx.clone(), dW.clone(), db.clone(), nws.clone()
  gx.zero_(); run(0)
  assert bool(torch.equal(gx, gx1)), tag + ": g_x differs between two launches"
  assert bool(torch.equal(dW, dW1)) and bool(torch.equal(db, db1)), tag + ": dW / db differ between two launches"
  nsum = lib.as_conv32_wino_bwd_parts() * 128             # [parts][64] doubles, compared as bit patterns
  assert bool(torch.equal(nws[:nsum].view(torch.int32), nws1[:nsum].view(torch.int32))), tag + ": sums differ"
  run(1)                                                   # accumulate flavour: a second launch adds the same gradient again
  close(dW, 2 * dW_ref, 1e-4 * float(dW_ref.abs().max()), 1e-4, tag + " accumulated dW")
  from conftest import parity_note
  parity_note("conv32_wino_bwd[%s]" % tag, g_z_bit_identical=True, g_x_max_err_vs_fp64=e_w, direct_g_x_max_err_vs_fp64=e_d,
              g_x_rms_err_vs_fp64=r_w, direct_g_x_rms_err_vs_fp64=r_d, dW_rel_l2_vs_fp64=dw_w, direct_dW_rel_l2_vs_fp64=dw_d)


@pytest.mark.parametrize("B,H,W,dil", _dgrad_generation_cases())
def test_backward_in_one_launch_by_minimal_filtering(B, H, W, dil):
  """as_conv32_wino_bwd_fused (csrc/conv32_wino_bwd.hip: data gradient and weight gradient side by side in one 8-wave workgroup
  per CU, g_z never written) against the two launches of as_conv32_wino_bwd: g_x BIT FOR BIT (the data-gradient waves run the
  same role code on the same tiles); dW is another order of the same sum over tiles, so both are held against the fp64 weight
  gradient from the two-launch path's fp32 g_z and the fused one may be at most 2x further away (or within 2e-6); db to summation
  order; the next BatchNorm's sums through the finalize they feed (256 partials here, 512 there); two launches bit-identical;
  nothing written into the halo.  Every dilation, the small odd geometries of the tiling's edges (shared columns of the shifted
  last segment: the weight gradient must count them once), the KITTI size at 4 pairs, single-segment images."""
  g = Pcl(B, 1, H, W, 0, 8, 8)
  shape = ops.conv_shape_2d(dil)
  lib = nat.load()
  assert lib.as_conv32_wino_ok(g, g, shape) == 1
  T = lambda seed: ops.ncdhw_to_pcl(rnd(B, 32, 1, H, W, seed=seed).to(DEV), g)
  x, g_a, zz, zn = T(3), T(4), T(5), T(6)
  w = (rnd(32, 32, 3, 3, seed=9) * 0.06).to(DEV)
  ww_t = torch.empty(16 * 1024, device=DEV)
  nat.call("as_conv32_wino_pack_weights", nat.ptr(w), nat.ptr(ww_t), 1, nat.stream())
  st, stn = ops.BnState(DEV), ops.BnState(DEV)
  for i, s_ in enumerate((st, stn)):
    s_.mean.copy_(rnd(32, seed=5 + 10 * i).to(DEV) * 0.1); s_.invstd.copy_(rnd(32, seed=6 + 10 * i).abs().to(DEV) + 0.5)
    s_.scale.copy_(s_.invstd * 1.1); s_.shift.copy_(rnd(32, seed=7 + 10 * i).to(DEV) * 0.1 - s_.mean * s_.scale)
  gamman = torch.full((32,), 1.1, device=DEV)
  coef = (rnd(96, seed=21) * 0.05).to(DEV); coef[64:] = coef[64:].abs() + 0.7
  tag = "fused wino bwd B%d H%d W%d d%d" % (B, H, W, dil)
  # the two launches
  gz, gx2 = ops.pcl_zeros(g, DEV), ops.pcl_zeros(g, DEV)
  dW2, db2 = torch.zeros(32, 32, 3, 3, device=DEV), torch.zeros(32, device=DEV)
  nws2 = torch.zeros(lib.as_bn_bwd_workspace(g), device=DEV)
  fws2 = torch.empty(lib.as_conv32_wino_bwd_workspace(), device=DEV)
  nat.call("as_conv32_wino_bwd", nat.ptr(x), g, nat.ptr(g_a), nat.ptr(zz), g, shape, nat.ptr(ww_t), nat.ptr(st.scale),
           nat.ptr(st.shift), nat.ptr(st.mean), nat.ptr(coef), 0.2, nat.ptr(zn), nat.ptr(stn.scale), nat.ptr(stn.shift),
           nat.ptr(stn.mean), nat.ptr(gz), nat.ptr(gx2), nat.ptr(dW2), nat.ptr(db2), 0, nat.ptr(nws2), nat.ptr(fws2), nat.stream())
  # one launch
  nparts = lib.as_conv32_wino_bwd_fused_parts()
  assert nparts * 128 <= lib.as_bn_bwd_workspace(g)
  def run(acc, gx, dW, db, nws):
    fws = torch.empty(lib.as_conv32_wino_bwd_fused_workspace(), device=DEV)
    nat.call("as_conv32_wino_bwd_fused", nat.ptr(x), g, nat.ptr(g_a), nat.ptr(zz), g, shape, nat.ptr(ww_t), nat.ptr(st.scale),
             nat.ptr(st.shift), nat.ptr(st.mean), nat.ptr(coef), 0.2, nat.ptr(zn), nat.ptr(stn.scale), nat.ptr(stn.shift),
             nat.ptr(stn.mean), nat.ptr(gx), nat.ptr(dW), nat.ptr(db), acc, nat.ptr(nws), nat.ptr(fws), nat.stream())
    torch.cuda.synchronize()
  gx, dW, db = ops.pcl_zeros(g, DEV), torch.zeros(32, 32, 3, 3, device=DEV), torch.zeros(32, device=DEV)
  nws = torch.zeros(lib.as_bn_bwd_workspace(g), device=DEV)
  run(0, gx, dW, db, nws)
  full = ops.pcl_view(gx, g).clone(); ops.pcl_interior(full, g).zero_()
  assert float(full.abs().max()) == 0.0, tag + ": g_x written into the halo"
  if not bool(torch.equal(gx, gx2)):
    va, vb = ops.pcl_view(gx, g), ops.pcl_view(gx2, g)
    bad = (va != vb).nonzero()
    raise AssertionError("%s: g_x differs from the two-launch form at %d elements; first [b, d, y, x, c] (padded) = %s: %r against %r" % (
        tag, bad.shape[0], bad[0].tolist(), float(va[tuple(bad[0])]), float(vb[tuple(bad[0])])))
  assert float(ops.pcl_interior(ops.pcl_view(gx, g), g).abs().max()) > 0.0
  # weight / bias gradient: fp64 from the two-launch path's fp32 g_z
  gz64 = ops.pcl_to_ncdhw(gz, g)[:, :, 0].double().cpu(); x64 = ops.pcl_to_ncdhw(x, g)[:, :, 0].double().cpu()
  dW64 = torch.nn.grad.conv2d_weight(x64, tuple(w.shape), gz64, padding=dil, dilation=dil)
  e1 = float((dW.double().cpu() - dW64).norm() / dW64.norm()); e2 = float((dW2.double().cpu() - dW64).norm() / dW64.norm())
  assert e2 < 3e-6 and e1 <= max(2.0 * e2, 2e-6), (tag, "dW", e1, e2)
  db64 = gz64.sum(dim=(0, 2, 3)); mag = float(gz64.abs().sum(dim=(0, 2, 3)).max())
  assert float((db.double().cpu() - db64).abs().max()) <= 5e-7 * mag, tag + ": bias gradient"
  # the next BatchNorm's sums through its finalize
  _, gg1, gb1 = ops.bn_act_bwd(gx, zn, stn, gamman, g, True, sums=ops.BnBwdSums(nws, nparts))
  _, gg2, gb2 = ops.bn_act_bwd(gx2, zn, stn, gamman, g, True, sums=ops.BnBwdSums(nws2, lib.as_conv32_wino_bwd_parts()))
  n = B * H * W
  close(gg1, gg2, 2e-6 * n ** 0.5 * float(gg2.abs().max()) + 1e-5, 1e-5, tag + " next g_gamma")
  close(gb1, gb2, 2e-6 * n ** 0.5 * float(gb2.abs().max()) + 1e-5, 1e-5, tag + " next g_beta")
  # determinism, and the accumulate flavour
  gx_b, dW_b, db_b = ops.pcl_zeros(g, DEV), torch.zeros(32, 32, 3, 3, device=DEV), torch.zeros(32, device=DEV)
  nws_b = torch.zeros(lib.as_bn_bwd_workspace(g), device=DEV)
  run(0, gx_b, dW_b, db_b, nws_b)
  assert bool(torch.equal(gx_b, gx)) and bool(torch.equal(dW_b, dW)) and bool(torch.equal(db_b, db)), tag + ": two launches differ"
  assert bool(torch.equal(nws_b[:nparts * 128].view(torch.int32), nws[:nparts * 128].view(torch.int32))), tag + ": sums differ"
  run(1, gx_b, dW_b, db_b, nws_b)
  close(dW_b, 2 * dW, 1e-4 * float(dW.abs().max()), 1e-4, tag + " accumulated dW")
  from conftest import parity_note
  parity_note("conv32_wino_bwd_fused[%s]" % tag, g_x_bit_identical_to_two_launches=True, dW_rel_l2_vs_fp64=e1,
              two_launch_dW_rel_l2_vs_fp64=e2)


# ----------------------------------------------------------------------------- a7 forward with the previous BN + LReLU on the way in
@pytest.mark.parametrize("B,H,W,dil,skip", [(2, 160, 1242, 1, True), (2, 161, 1242, 2, True), (1, 375, 1030, 4, True),
                                            (2, 163, 1237, 8, True), (4, 97, 700, 1, False)])
def test_conv32_forward_with_previous_activation_on_the_way_in(B, H, W, dil, skip):
  """as_conv32_act_fwd — a_prev = lrelu(z_prev*scale + shift) (+ a_prevprev) formed while the operand is staged, written
  back once, then the layer's convolution and BatchNorm moments — against as_bn_act_fwd followed by as_conv32_fwd: the
  by-product and the convolution output bit for bit (same arithmetic chain, same K order), the moments after the merge to
  rounding.  Ragged last segment, every dilation, comb residues that do not divide H, with and without skip."""
  g = Pcl(B, 1, H, W, 0, 8, 8)
  shape = ops.conv_shape_2d(dil)
  lib = nat.load()
  assert lib.as_conv32_act_ok(g, g, shape) == 1
  z_prev = ops.ncdhw_to_pcl(rnd(B, 32, 1, H, W, seed=1).to(DEV), g)
  a_pp = ops.ncdhw_to_pcl(rnd(B, 32, 1, H, W, seed=2).to(DEV), g) if skip else None
  w = (rnd(32, 32, 3, 3, seed=9) * 0.06).to(DEV)
  b = (rnd(32, seed=10) * 0.1).to(DEV)
  wp = ops.pack_weights(w, shape, False)
  st = ops.BnState(DEV)
  st.scale.copy_(rnd(32, seed=5).abs().to(DEV) + 0.5); st.shift.copy_(rnd(32, seed=6).to(DEV) * 0.3)
  # two launches
  a_ref = ops.bn_act(z_prev, st, g, residual=a_pp, out=ops.pcl_zeros(g, DEV))
  stats_ref = ops.conv32_stat_parts(g, g, shape, DEV)
  z_ref = ops.conv32(a_ref, g, wp, b, g, shape, out=ops.pcl_zeros(g, DEV), stats=stats_ref)
  # one launch
  a_out, z = ops.pcl_zeros(g, DEV), ops.pcl_zeros(g, DEV)
  stats = ops.StatParts(lib.as_conv32_act_parts(), DEV)
  nat.call("as_conv32_act_fwd", nat.ptr(z_prev), nat.ptr(a_pp), nat.ptr(st.scale), nat.ptr(st.shift), nat.ptr(a_out), g,
           nat.ptr(wp), nat.ptr(b), 0.2, nat.ptr(z), g, shape, nat.ptr(stats.mean), nat.ptr(stats.m2), nat.ptr(stats.cnt),
           nat.stream())
  tag = "B%d H%d W%d d%d %s" % (B, H, W, dil, "skip" if skip else "plain")
  for name, got in (("a_out", a_out), ("z", z)):
    full = ops.pcl_view(got, g).clone(); ops.pcl_interior(full, g).zero_()
    assert float(full.abs().max()) == 0.0, tag + ": %s written into the halo" % name
  ai, ai_ref = ops.pcl_interior(ops.pcl_view(a_out, g), g), ops.pcl_interior(ops.pcl_view(a_ref, g), g)
  zi, zi_ref = ops.pcl_interior(ops.pcl_view(z, g), g), ops.pcl_interior(ops.pcl_view(z_ref, g), g)
  close(ai, ai_ref, 1e-6, 1e-6, tag + " by-product")
  close(zi, zi_ref, 2e-6, 1e-6, tag + " z")
  a_exact, z_exact = bool(torch.equal(ai, ai_ref)), bool(torch.equal(zi, zi_ref))
  assert a_exact, tag + ": by-product differs from as_bn_act_fwd"
  # moments: finalize both sets of partials
  gam, bet = torch.ones(32, device=DEV), torch.zeros(32, device=DEV)
  fin = [ops.bn_train_stats(sp, gam, bet, torch.zeros(32, device=DEV), torch.ones(32, device=DEV)) for sp in (stats, stats_ref)]
  close(fin[0].mean, fin[1].mean, 2e-6, 1e-5, tag + " batch mean")
  close(fin[0].invstd, fin[1].invstd, 0, 2e-6, tag + " batch invstd")
  assert float(stats.cnt.sum()) == float(B * H * W)
  from conftest import parity_note
  parity_note("conv32_act[%s]" % tag, by_product_bit_identical=a_exact, z_bit_identical=z_exact)


@pytest.mark.parametrize("B,H,W,dil,skip", [(2, 160, 1242, 1, True), (2, 161, 1242, 2, True), (1, 375, 1030, 4, True),
                                            (2, 163, 1237, 8, True), (4, 97, 700, 1, False), (1, 375, 1242, 8, False),
                                            (12, 375, 64, 2, True)])
def test_conv32_forward_by_minimal_filtering(B, H, W, dil, skip):
  """as_conv32_wino_fwd — the layer of as_conv32_act_fwd by the minimal-filtering algorithm F(2x2, 3x3) — against
  as_conv32_act_fwd's two-launch equivalent (as_bn_act_fwd, as_conv32_fwd): the by-product bit for bit (the same element-wise
  chain); the convolution output is a different association of the same sum, so it is compared with the fp64 convolution of the
  SAME fp32 operand, where its error must stay within 2x the direct kernel's own (and the direct kernel within 2e-6 of it).
  Odd comb lengths (a pair with one row), ragged last segment, every dilation, with and without skip, W == one segment."""
  g = Pcl(B, 1, H, W, 0, 8, 8)
  shape = ops.conv_shape_2d(dil)
  lib = nat.load()
  assert lib.as_conv32_wino_ok(g, g, shape) == 1
  z_prev = ops.ncdhw_to_pcl(rnd(B, 32, 1, H, W, seed=1).to(DEV), g)
  a_pp = ops.ncdhw_to_pcl(rnd(B, 32, 1, H, W, seed=2).to(DEV), g) if skip else None
  w = (rnd(32, 32, 3, 3, seed=9) * 0.06).to(DEV)
  b = (rnd(32, seed=10) * 0.1).to(DEV)
  wp = ops.pack_weights(w, shape, False)
  ww = torch.empty(16 * 1024, device=DEV)
  nat.call("as_conv32_wino_pack_weights", nat.ptr(w), nat.ptr(ww), 0, nat.stream())
  st = ops.BnState(DEV)
  st.scale.copy_(rnd(32, seed=5).abs().to(DEV) + 0.5); st.shift.copy_(rnd(32, seed=6).to(DEV) * 0.3)
  a_ref = ops.bn_act(z_prev, st, g, residual=a_pp, out=ops.pcl_zeros(g, DEV))
  stats_ref = ops.conv32_stat_parts(g, g, shape, DEV)
  z_ref = ops.conv32(a_ref, g, wp, b, g, shape, out=ops.pcl_zeros(g, DEV), stats=stats_ref)
  a_out, z = ops.pcl_zeros(g, DEV), ops.pcl_zeros(g, DEV)
  stats = ops.StatParts(lib.as_conv32_wino_parts(), DEV)
  nat.call("as_conv32_wino_fwd", nat.ptr(z_prev), nat.ptr(a_pp), nat.ptr(st.scale), nat.ptr(st.shift), nat.ptr(a_out), g,
           nat.ptr(ww), nat.ptr(b), 0.2, nat.ptr(z), g, shape, nat.ptr(stats.mean), nat.ptr(stats.m2), nat.ptr(stats.cnt),
           nat.stream())
  tag = "wino B%d H%d W%d d%d %s" % (B, H, W, dil, "skip" if skip else "plain")
  # determinism: the columns the shifted last segment shares with its neighbour fall into different tiles there (different
  # rounding) — they must be written by one of the two only
  z2, a2 = ops.pcl_zeros(g, DEV), ops.pcl_zeros(g, DEV)
  for _ in range(2):
    nat.call("as_conv32_wino_fwd", nat.ptr(z_prev), nat.ptr(a_pp), nat.ptr(st.scale), nat.ptr(st.shift), nat.ptr(a2), g,
             nat.ptr(ww), nat.ptr(b), 0.2, nat.ptr(z2), g, shape, nat.ptr(stats.mean), nat.ptr(stats.m2), nat.ptr(stats.cnt),
             nat.stream())
    assert bool(torch.equal(z2, z)), tag + ": two launches differ"
  for name, got in (("a_out", a_out), ("z", z)):
    full = ops.pcl_view(got, g).clone(); ops.pcl_interior(full, g).zero_()
    assert float(full.abs().max()) == 0.0, tag + ": %s written into the halo" % name
  ai, ai_ref = ops.pcl_interior(ops.pcl_view(a_out, g), g), ops.pcl_interior(ops.pcl_view(a_ref, g), g)
  zi, zi_ref = ops.pcl_interior(ops.pcl_view(z, g), g), ops.pcl_interior(ops.pcl_view(z_ref, g), g)
  assert bool(torch.equal(ai, ai_ref)), tag + ": by-product differs from as_bn_act_fwd"
  # the fp64 convolution of the same fp32 operand
  a_nchw = ops.pcl_to_ncdhw(a_ref, g)[:, :, 0].double().cpu()
  z64 = torch.nn.functional.conv2d(a_nchw, w.double().cpu(), b.double().cpu(), padding=dil, dilation=dil)
  z64 = z64.permute(0, 2, 3, 1).reshape(zi.shape).to(DEV)
  err_w = float((zi.double() - z64).abs().max()); err_d = float((zi_ref.double() - z64).abs().max())
  scale = float(z64.abs().max())
  assert err_d <= 2e-6 * scale, (tag, err_d, scale)
  assert err_w <= max(2.0 * err_d, 1e-6 * scale), (tag, err_w, err_d, scale)
  rms_w = float((zi.double() - z64).pow(2).mean().sqrt()); rms_d = float((zi_ref.double() - z64).pow(2).mean().sqrt())
  assert rms_w <= 2.0 * rms_d, (tag, rms_w, rms_d)
  gam, bet = torch.ones(32, device=DEV), torch.zeros(32, device=DEV)
  fin = [ops.bn_train_stats(sp, gam, bet, torch.zeros(32, device=DEV), torch.ones(32, device=DEV)) for sp in (stats, stats_ref)]
  close(fin[0].mean, fin[1].mean, 2e-6, 1e-5, tag + " batch mean")
  close(fin[0].invstd, fin[1].invstd, 0, 2e-6, tag + " batch invstd")
  assert float(stats.cnt.sum()) == float(B * H * W)
  from conftest import parity_note
  parity_note("conv32_wino[%s]" % tag, by_product_bit_identical=True, max_err_vs_fp64=err_w, direct_max_err_vs_fp64=err_d,
              rms_err_vs_fp64=rms_w, direct_rms_err_vs_fp64=rms_d)


def test_fused_full_resolution_kernels_refuse_unsupported_geometry():
  """as_conv32_act_fwd / as_conv32_bwd_fused are built for the refinement's geometry (2-D 3x3, dilation <= 8 inside an 8-voxel
  halo, rows of >= 128 / 64 pixels, enough tiles for the fixed grid): everything else is declined by the *_ok query and
  refused with an error by the entry point — never launched."""
  lib = nat.load()
  s1 = ops.conv_shape_2d(1)
  big = Pcl(2, 1, 160, 1242, 0, 8, 8)
  assert lib.as_conv32_act_ok(big, big, s1) == 1 and lib.as_conv32_bwd_fused_ok(big, big, s1) == 1
  cases = [(Pcl(2, 1, 96, 256, 0, 8, 8), s1, "too few tiles for 512 workgroups"),
           (Pcl(2, 1, 400, 100, 0, 8, 8), s1, "rows narrower than a segment (act) / too few tiles"),
           (Pcl(2, 1, 160, 1242, 0, 4, 4), ops.conv_shape_2d(8), "dilation reaches beyond the halo"),
           (Pcl(2, 1, 160, 1242, 0, 8, 4), s1, "halo columns < 8")]
  for g, shape, why in cases:
    assert lib.as_conv32_act_ok(g, g, shape) == 0, why
  for g, shape, why in cases[:1] + cases[2:]:
    assert lib.as_conv32_bwd_fused_ok(g, g, shape) == 0, why
  g = cases[0][0]
  buf = [ops.pcl_zeros(g, DEV) for _ in range(4)]
  st = ops.BnState(DEV)
  wp = torch.zeros(9 * 1024, device=DEV)
  stats = ops.StatParts(lib.as_conv32_act_parts(), DEV)
  with pytest.raises(RuntimeError, match="not supported"):
    nat.call("as_conv32_act_fwd", nat.ptr(buf[0]), nat.ptr(buf[1]), nat.ptr(st.scale), nat.ptr(st.shift), nat.ptr(buf[2]), g,
             nat.ptr(wp), None, 0.2, nat.ptr(buf[3]), g, s1, nat.ptr(stats.mean), nat.ptr(stats.m2), nat.ptr(stats.cnt), nat.stream())
  with pytest.raises(RuntimeError, match="slope"):
    nat.call("as_conv32_act_fwd", nat.ptr(ops.pcl_zeros(big, DEV)), None, nat.ptr(st.scale), nat.ptr(st.shift),
             nat.ptr(ops.pcl_zeros(big, DEV)), big, nat.ptr(wp), None, 1.5, nat.ptr(ops.pcl_zeros(big, DEV)), big, s1, None, None,
             None, nat.stream())
  ws = torch.empty(lib.as_bn_bwd_workspace(g), device=DEV)
  fws = torch.empty(lib.as_conv32_bwd_fused_workspace(), device=DEV)
  dW, db = torch.zeros(32, 32, 3, 3, device=DEV), torch.zeros(32, device=DEV)
  coef = torch.zeros(96, device=DEV)
  with pytest.raises(RuntimeError, match="not supported"):
    nat.call("as_conv32_bwd_fused", nat.ptr(buf[0]), g, nat.ptr(buf[1]), nat.ptr(buf[2]), g, s1, nat.ptr(wp), nat.ptr(st.scale),
             nat.ptr(st.shift), nat.ptr(st.mean), nat.ptr(coef), 0.2, nat.ptr(buf[2]), nat.ptr(st.scale), nat.ptr(st.shift),
             nat.ptr(st.mean), nat.ptr(buf[3]), nat.ptr(dW), nat.ptr(db), 0, nat.ptr(ws), nat.ptr(fws), nat.stream())


@pytest.mark.parametrize("B,H,W", [(1, 2, 2), (3, 8, 11), (1, 41, 67), (2, 33, 59), (1, 17, 60), (2, 9, 61), (1, 19, 62), (1, 12, 63),
                                   (3, 25, 121), (1, 37, 125), (2, 96, 256), (3, 130, 701), (1, 375, 1242)])
def test_monodepth_loss_row_strips_equal_the_pixel_kernel(B, H, W):
  """as_monodepth_loss_rows_fwd (csrc/photometric_rows.hip: a wave walks a strip of rows, 3x3 windows from neighbouring lanes
  by DPP and from registers) against as_monodepth_loss_fwd (one thread per pixel, every tap loaded): the four loss maps bit for
  bit — widths around the 62-column blocks, heights below / across the strip height, image planes that are not 16-byte multiples."""
  lib = nat.load()
  g = torch.Generator().manual_seed(7 + H + W)
  img = torch.rand(B, 3, H, W, generator=g).to(DEV)
  warped = (img.cpu() + (torch.rand(B, 3, H, W, generator=g) - 0.5) * 0.3).clamp(0, 1).to(DEV)
  pred = (torch.rand(B, 1, H, W, generator=g) * 20 + 0.5).to(DEV)
  ws_a = torch.empty(lib.as_monodepth_workspace(B, H, W), device=DEV)
  ws_b = torch.empty(lib.as_photometric_chain_workspace(B, H, W), device=DEV)
  ref = [torch.full((B, 1, H, W), float("nan"), device=DEV) for _ in range(4)]
  got = [torch.full((B, 1, H, W), float("nan"), device=DEV) for _ in range(4)]
  nat.call("as_monodepth_loss_fwd", nat.ptr(pred), nat.ptr(img), nat.ptr(warped), B, H, W, 1e-3, *[nat.ptr(t) for t in ref],
           nat.ptr(ws_a), nat.stream())
  nat.call("as_monodepth_loss_rows_fwd", nat.ptr(pred), nat.ptr(img), nat.ptr(warped), B, H, W, 1e-3, *[nat.ptr(t) for t in got],
           nat.ptr(ws_b), nat.stream())
  for name, a, b in zip(("total", "l1", "ssim", "smooth"), got, ref):
    assert bool(torch.isfinite(b).all()) and torch.equal(a, b), (name, float((a - b).abs().max()))


@pytest.mark.parametrize("B,H,W", [(1, 2, 2), (3, 8, 11), (1, 41, 67), (2, 33, 59), (1, 17, 60), (2, 9, 61), (1, 19, 62), (3, 25, 121),
                                   (2, 96, 256), (3, 130, 701), (1, 375, 1242), (4, 375, 1242)])
def test_masked_photometric_loss_in_one_node_equals_the_separate_functions(B, H, W):
  """hip_ops.MaskedPhotometricFn (warp + monodepth loss + masked mean as one autograd node, the gradient map mask * g / N
  never built, the disparity's two gradients added by the warp's backward kernel) against LinearWarpFn -> MonodepthLossFn ->
  masked_mean stitched by autograd: loss, sum, count, warped image, mask and d loss / d pred bit for bit — for the mean
  (one-GPU step) and for the sum (what a data-parallel rank back-propagates)."""
  from adaptive_stereo import hip_ops as ops
  g = torch.Generator().manual_seed(41)
  left = torch.rand(B, 3, H, W, generator=g).to(DEV)
  right = torch.rand(B, 3, H, W, generator=g).to(DEV)
  pred0 = (torch.rand(B, 1, H, W, generator=g) * min(12.0, W / 4.0)).to(DEV)      # (narrow images: keep some samples inside)
  for use_sum in (False, True):
    p1 = pred0.clone().requires_grad_(True)
    warped, mask = ops.LinearWarpFn.apply(right, p1, True)
    total = ops.MonodepthLossFn.apply(p1, left, warped, 1e-3)[0]
    if use_sum:
      m8 = mask.to(torch.uint8)
      ref = (total * m8).sum()          # (value only: fp32 torch sum; the gradient is what is compared)
      total.backward(m8.to(torch.float32))
    else:
      ref = ops.masked_mean(total, mask)
      ref.backward()
    p2 = pred0.clone().requires_grad_(True)
    mean, lsum, count, warped2, mask2 = ops.MaskedPhotometricFn.apply(p2, left, right, 1e-3)
    (lsum if use_sum else mean).backward()
    assert torch.equal(warped2, warped) and torch.equal(mask2.bool(), mask)
    assert int(count) == int(mask.sum())
    if not use_sum:
      # (fp64 partial sums in a different order, rounded to fp32 once: the same float unless the fp64 sum sits on a rounding
      # boundary)
      assert abs(float(mean) - float(ref)) <= 1.2e-7 * abs(float(ref))
    else:
      assert abs(float(lsum) - float(ref)) <= 1e-5 * abs(float(ref))
    assert torch.equal(p2.grad, p1.grad), float((p2.grad - p1.grad).abs().max())
    assert float(p2.grad.abs().max()) > 0.0
