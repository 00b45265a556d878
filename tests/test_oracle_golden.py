"""Pins the oracle (oracle/stereo_oracle.py) against golden vectors produced by the
reference's own modules (tests/golden/make_golden.py).  CPU only.

Tolerances: the oracle and the reference both run ATen CPU fp32 kernels; the only
differences in form are the vectorised cost volume (exact) and explicit state handling,
so agreement is expected at the few-ulp level.  Tolerances below are absolute + relative
and are stated per quantity.
"""
import pytest
import torch

from conftest import GOLDEN_CASES
from adaptive_stereo.models.stereo_net import StereoNet, FeatureExtractorNetwork
from adaptive_stereo.utils import synthetic as syn
from oracle import stereo_oracle as orc

FAST_CASES = [c for c in GOLDEN_CASES]


def build_states(meta):
  fnet = FeatureExtractorNetwork(meta["k"])
  snet = StereoNet(meta["k"], 1, meta["s"], maxdisp=meta["maxdisp"])
  fsd = syn.synthetic_state_dict(fnet.state_dict(), seed=123)
  ssd = syn.synthetic_state_dict(snet.state_dict(), seed=123, logit_gain=meta["gain"])
  return fsd, ssd


def inputs(meta, gold):
  left, right = syn.stereo_pair(meta["B"], meta["H"], meta["W"], seed=1)
  for name, t in (("left", left), ("right", right)):
    s, ss = syn.checksum(t)
    es, ess = gold.z["sum__" + name]
    assert abs(s - es) <= 1e-6 * abs(es) and abs(ss - ess) <= 1e-6 * abs(ess), \
        "synthetic %s image differs from the one the fixture was generated with" % name
  return left, right


def after_step_atol(gold, net, name, lr, clip_coef, scale=1.0):
  """Absolute tolerance for a tensor after one Adam step.

  Adam's first step moves a weight by lr*g/(|g|+1e-8): where the reference's own gradient is at
  rounding-noise level (|g| < 1e-6 — e.g. every conv bias in front of a train-mode BatchNorm, whose
  exact gradient is zero) the step is +-lr times noise, and no two correct fp32 implementations
  agree there.  Those elements get 2.1*lr; everything else 2e-6.  ``scale`` is the logit gain of
  the case: gradient magnitudes (and their rounding noise) grow with it."""
  gkey = "grad/%s.%s" % (net, name)
  if not gold.has(gkey):
    return 2e-6
  g = gold.expected(gkey)[0].abs().double()
  if net == "stereo":
    g = g * clip_coef
  return 2e-6 + 2.1 * lr * (g < 1e-6 * scale).double()


@pytest.mark.parametrize("case", FAST_CASES)
def test_oracle_eval_forward_matches_reference(case, golden_loader):
  gold = golden_loader(case)
  meta = gold.meta
  torch.set_num_threads(8)
  fsd, ssd = build_states(meta)
  left, right = inputs(meta, gold)
  out, fcs = orc.forward_only(fsd, ssd, left, right, meta["k"], meta["s"], meta["maxdisp"])
  k, s = meta["k"], meta["s"]
  scale = max(1.0, meta["gain"])
  gold.compare("eval/logits", out["cost_volume_l/%d" % (s + k)], atol=2e-6 * scale, rtol=1e-5)
  gold.compare("eval/pred_coarse_up", out["pred_disp_l/%d" % (s + k)], atol=2e-5, rtol=1e-5)
  gold.compare("eval/pred_refined", out["pred_disp_l/%d" % s], atol=2e-5, rtol=1e-5)
  gold.compare("eval/fcs", fcs, atol=2e-6 * scale, rtol=1e-5)


@pytest.mark.parametrize("case", FAST_CASES)
def test_oracle_adapt_step_matches_reference(case, golden_loader):
  gold = golden_loader(case)
  meta = gold.meta
  torch.set_num_threads(8)
  fsd, ssd = build_states(meta)
  left, right = inputs(meta, gold)
  fp, sp = orc.make_params(fsd, True), orc.make_params(ssd, True)
  taps = {}
  res = orc.adapt_step(fp, sp, {}, left, right, meta["k"], meta["s"], meta["maxdisp"], lr=meta["lr"], taps=taps)
  k, s = meta["k"], meta["s"]
  scale = max(1.0, meta["gain"])
  out = res["outputs"]

  gold.compare("train/fl", taps["fl"], atol=1e-5, rtol=1e-5)
  gold.compare("train/fr", taps["fr"], atol=1e-5, rtol=1e-5)
  for i in range(4):
    gold.compare("train/filter%d" % i, taps["filter%d" % i], atol=2e-5, rtol=1e-5)
  logits = out["cost_volume_l/%d" % (s + k)]
  gold.compare("train/logits", logits, atol=2e-6 * scale, rtol=1e-5)
  gold.compare("train/pred_coarse", taps["pred"], atol=2e-5, rtol=1e-5)
  gold.compare("train/pred_refined", out["pred_disp_l/%d" % s], atol=5e-5, rtol=1e-5)
  gold.compare("train/warped", res["warped"], atol=2e-5)
  gold.compare("train/mask", res["mask"], atol=0)

  # arg-max indices: bit-exact wherever the reference's own top-2 gap exceeds fp32 noise
  am = torch.argmax(logits.detach(), dim=1).to(torch.int32)
  ref_am = gold.full("train/argmax")
  gap = gold.full("train/top2gap")
  safe = gap > 1e-6 * scale
  assert bool((am[safe] == ref_am[safe]).all())
  assert float((am != ref_am).float().mean()) < 1e-3

  assert abs(float(res["loss"]) - gold.scalar("train/loss")) < 2e-6
  assert abs(float(res["fcs"]) - gold.scalar("train/fcs_mean")) < 1e-5 * max(1.0, abs(gold.scalar("train/fcs_mean")))

  # gradients w.r.t. the features and every parameter (pre-clip values are not kept by the
  # oracle: feature_net grads are never clipped, stereo_net grads are compared after un-clipping)
  gold.compare("train/grad_fl", taps["fl"].grad, atol=2e-7 * scale, rtol=2e-3)
  gold.compare("train/grad_fr", taps["fr"].grad, atol=2e-7 * scale, rtol=2e-3)
  norm = gold.scalar("train/stereo_grad_norm")
  coef = min(1.0, 1.0 / (norm + 1e-6))
  for name, p in sp.items():
    key = "grad/stereo." + name
    if not p.requires_grad:
      continue
    if "stereo." + name in gold.no_grad_keys:
      assert p.grad is None and orc.conv2_is_dead(name)
      continue
    gmax = float(p.grad.abs().max()) / coef
    gold.compare(key, p.grad / coef, atol=1e-6 * scale + 2e-4 * gmax, rtol=2e-3)
  for name, p in fp.items():
    if not p.requires_grad:
      continue
    if "feature." + name in gold.no_grad_keys:
      assert p.grad is None and orc.conv2_is_dead(name)
      continue
    gmax = float(p.grad.abs().max())
    gold.compare("grad/feature." + name, p.grad, atol=1e-6 * scale + 2e-4 * gmax, rtol=2e-3)

  # state after the step: BN running statistics and Adam-updated weights
  for net, group in (("stereo", sp), ("feature", fp)):
    for name, p in group.items():
      key = "after/%s.%s" % (net, name)
      if name.endswith("num_batches_tracked"):
        assert int(p) == int(gold.z[key]), key
      else:
        gold.compare(key, p, atol=after_step_atol(gold, net, name, meta["lr"], coef, scale), rtol=1e-5)


def test_khamis_matches_reference(golden_loader):
  gold = golden_loader("crop_96x256_k4_b1")
  meta = gold.meta
  fsd, ssd = build_states(meta)
  left, right = inputs(meta, gold)
  fp, sp = orc.make_params(fsd, True), orc.make_params(ssd, True)
  res = orc.adapt_step(fp, sp, {}, left, right, meta["k"], meta["s"], meta["maxdisp"], lr=meta["lr"])
  pred = res["outputs"]["pred_disp_l/0"].detach()
  gt = (pred + 0.5).clone()
  gt[:, :, ::3, ::5] = 0.0
  assert abs(float(orc.khamis_robust_loss(pred, gt)) - gold.scalar("train/khamis")) < 1e-6


def test_cost_volume_is_the_difference_volume():
  """stereo_net.py:173-184 — a difference (not a correlation), zero in the x<d wedge."""
  torch.manual_seed(0)
  fl, fr = torch.randn(2, 32, 5, 9), torch.randn(2, 32, 5, 9)
  vol = orc.cost_volume(fl, fr, 4)
  assert vol.shape == (2, 32, 4, 5, 9)
  for d in range(4):
    assert torch.equal(vol[:, :, d, :, :d], torch.zeros(2, 32, 5, d))
    assert torch.equal(vol[:, :, d, :, d:], fl[:, :, :, d:] - fr[:, :, :, :9 - d])
