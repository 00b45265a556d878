"""How far does the REFERENCE's own answer move when nothing but the fp32 summation order of its convolutions changes?

The GPU kernels sum the 864 (3-D) / 288 (2-D) products of an output value in a different order than oneDNN does on the
CPU (tap-major fp32 fma chains on the matrix cores).  Both are correct fp32 convolutions; this script measures what such a
reassociation alone does to the quantities the parity tests bound, so that the tests' tolerances are derived, not
guessed: the oracle (pinned against the reference's golden vectors) is run twice on the fixtures' inputs — once as is,
once with every convolution evaluated as a sum over kernel taps of per-tap matrix products, taps visited in a given
order — and the differences are written to tests/golden/reassociation_bound.json:

  logit_delta_over_gain      max |delta logits| / gain
  argmax_flips               pixels whose arg-max index changes, and the largest top-2 gap (over gain) among them
  epe_delta_refined          mean |delta pred_disp_l/0|  (the north-star's EPE, oracle against itself)
  loss_delta

and, for the BACKWARD pass (the same two runs, autograd through the tap sums):

  worst_tensor_rel_l2        max over parameter tensors of |g' - g|_2 / |g|_2   (tensors whose gradient is rounding noise, and
                             the two single-number output biases, excluded exactly as tests/test_gpu_end_to_end.py excludes them)
  whole_{stereo,feature}_rel_l2   the same over each network's whole gradient vector
  clip_norm_rel_delta        relative change of the stereo network's gradient norm (what clip_grad_norm_ scales by)
  sign_flips / elements      gradient elements above the noise floor whose sign differs (Adam's first step is +-lr by sign)

tests/test_gpu_end_to_end.py:grad_bounds() is derived from these rows.

Run in the build container (CPU):  python tests/tools/reassociation_bound.py [case ...]
"""
import json
import os
import sys

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import conftest                                          # noqa: E402,F401
from conftest import Golden                              # noqa: E402
from adaptive_stereo.models.stereo_net import StereoNet, FeatureExtractorNetwork   # noqa: E402
from adaptive_stereo.utils import synthetic as syn       # noqa: E402
from oracle import stereo_oracle as orc                  # noqa: E402

CASES = ["plumbing_240x320_k3_b1", "crop_96x256_k4_b1", "crop_96x256_k4_b2_trained", "odd_75x131_k3_b1",
         "kitti_375x1242_k4_b1", "sceneflow_540x960_k4_b1"]
BIG = ("kitti_375x1242_k4_b1", "sceneflow_540x960_k4_b1")      # full-size cases: the two all-convolution orders only


class TapSum(object):
  """torch.nn.functional with conv2d / conv3d evaluated tap by tap (each tap one [Cout,Cin] matrix product over the
  shifted input), taps accumulated in ``order`` (+1 first-to-last, -1 last-to-first)."""

  def __init__(self, order, dims):
    self.order, self.dims = order, dims

  def __getattr__(self, name):
    return getattr(F, name)

  def _conv(self, x, w, b, stride, padding, dilation, nd):
    st = (stride,) * nd if isinstance(stride, int) else tuple(stride)
    pd = (padding,) * nd if isinstance(padding, int) else tuple(padding)
    dl = (dilation,) * nd if isinstance(dilation, int) else tuple(dilation)
    pads = []
    for p in reversed(pd):
      pads += [p, p]
    xp = F.pad(x, pads)
    ks = w.shape[2:]
    out_sz = [(xp.shape[2 + i] - dl[i] * (ks[i] - 1) - 1) // st[i] + 1 for i in range(nd)]
    taps = [()]
    for i in range(nd):
      taps = [t + (j,) for t in taps for j in range(ks[i])]
    if self.order < 0:
      taps = taps[::-1]
    acc = None
    for t in taps:
      sl = [slice(None), slice(None)]
      for i in range(nd):
        s0 = t[i] * dl[i]
        sl.append(slice(s0, s0 + st[i] * (out_sz[i] - 1) + 1, st[i]))
      xs = xp[tuple(sl)]
      wt = w[(slice(None), slice(None)) + t]                  # [Cout, Cin]
      term = torch.einsum("oc,bc...->bo...", wt, xs)
      acc = term if acc is None else acc + term
    if b is not None:
      acc = acc + b.view((1, -1) + (1,) * nd)
    return acc

  def conv3d(self, x, w, b=None, stride=1, padding=0, dilation=1):
    if 3 not in self.dims:
      return F.conv3d(x, w, b, stride=stride, padding=padding, dilation=dilation)
    return self._conv(x, w, b, stride, padding, dilation, 3)

  def conv2d(self, x, w, b=None, stride=1, padding=0, dilation=1):
    if 2 not in self.dims:
      return F.conv2d(x, w, b, stride=stride, padding=padding, dilation=dilation)
    return self._conv(x, w, b, stride, padding, dilation, 2)


def run(meta, shim):
  fnet = FeatureExtractorNetwork(meta["k"])
  snet = StereoNet(meta["k"], 1, meta["s"], maxdisp=meta["maxdisp"])
  fsd = syn.synthetic_state_dict(fnet.state_dict(), seed=123)
  ssd = syn.synthetic_state_dict(snet.state_dict(), seed=123, logit_gain=meta["gain"])
  left, right = syn.stereo_pair(meta["B"], meta["H"], meta["W"], seed=1)
  fp, sp = orc.make_params(fsd, True), orc.make_params(ssd, True)
  saved = orc.F
  orc.F = shim if shim is not None else saved
  try:
    # no clip, lr 0: p.grad keeps the raw gradients of the step, the parameters do not move
    res = orc.adapt_step(fp, sp, {}, left, right, meta["k"], meta["s"], meta["maxdisp"], lr=0.0, clip=False)
  finally:
    orc.F = saved
  k, s = meta["k"], meta["s"]
  grads = {}
  for net, group in (("stereo", sp), ("feature", fp)):
    for name, p in group.items():
      if p.requires_grad and p.grad is not None:
        grads[(net, name)] = p.grad.detach().clone()
  return (res["outputs"]["cost_volume_l/%d" % (s + k)].detach(), res["outputs"]["pred_disp_l/%d" % s].detach(),
          float(res["loss"]), grads)


def grad_rows(base, other, scale):
  """The quantities tests/test_gpu_end_to_end.py bounds, for two gradient sets of the same step."""
  worst = (0.0, "")
  whole = {"stereo": [0.0, 0.0], "feature": [0.0, 0.0]}
  flips = elems = 0
  for (net, name), g in base.items():
    h = other[(net, name)]
    diff = (h.double() - g.double())
    whole[net][0] += float(diff.pow(2).sum()); whole[net][1] += float(g.double().pow(2).sum())
    if float(g.abs().max()) < 1e-6 * scale or name.endswith(("conv2d_out.bias", "conv3d_alone.bias")):
      continue
    rel = float(diff.norm() / g.double().norm())
    worst = max(worst, (rel, "%s.%s" % (net, name)))
    live = torch.minimum(g.abs(), h.abs()) >= 1e-6 * scale
    flips += int(((torch.sign(g) != torch.sign(h)) & live).sum()); elems += int(live.sum())
  norm = lambda gs: float(torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(v) for (n_, _), v in gs.items() if n_ == "stereo"])))
  nb, no = norm(base), norm(other)
  return {"worst_tensor_rel_l2": worst[0], "worst_tensor": worst[1],
          "whole_stereo_rel_l2": (whole["stereo"][0] / whole["stereo"][1]) ** 0.5,
          "whole_feature_rel_l2": (whole["feature"][0] / whole["feature"][1]) ** 0.5,
          "clip_norm_rel_delta": abs(no - nb) / nb, "sign_flips": flips, "elements": elems}


def main():
  torch.set_num_threads(8)
  path = os.path.join(HERE, "..", "golden", "reassociation_bound.json")
  report = {"torch": torch.__version__, "what": __doc__.split("\n")[0], "cases": {}}
  only = sys.argv[1:]
  if only and os.path.exists(path):
    report["cases"] = json.load(open(path))["cases"]          # re-measure the named cases, keep the others
  for case in CASES:
    if only and case not in only:
      continue
    meta = Golden(case).meta
    scale = max(1.0, meta["gain"])
    base_logits, base_pred, base_loss, base_grads = run(meta, None)
    srt = torch.sort(base_logits, dim=1, descending=True)[0]
    gap = srt[:, 0] - srt[:, 1]
    base_am = torch.argmax(base_logits, dim=1)
    rows = {}
    variants = (("conv3d_taps_fwd", TapSum(+1, (3,))), ("conv3d_taps_rev", TapSum(-1, (3,))),
                ("all_convs_taps_fwd", TapSum(+1, (2, 3))), ("all_convs_taps_rev", TapSum(-1, (2, 3))))
    for label, shim in (variants[2:] if case in BIG else variants):
      logits, pred, loss, grads = run(meta, shim)
      flips = torch.argmax(logits, dim=1) != base_am
      rows[label] = {
          "logit_delta_over_gain": float((logits - base_logits).abs().max()) / scale,
          "argmax_flips": int(flips.sum()), "pixels": int(flips.numel()),
          "max_gap_at_flip_over_gain": float(gap[flips].max()) / scale if bool(flips.any()) else 0.0,
          "epe_delta_refined": float((pred - base_pred).abs().mean()),
          "max_delta_refined": float((pred - base_pred).abs().max()),
          "loss_delta": abs(loss - base_loss), "backward": grad_rows(base_grads, grads, scale)}
      print(case, label, rows[label], flush=True)
    report["cases"][case] = {"gain": meta["gain"], "rows": rows}
  with open(path, "w") as f:
    json.dump(report, f, indent=1, sort_keys=True)


if __name__ == "__main__":
  main()
