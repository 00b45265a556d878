"""Prints, per golden case, the numbers the GPU parity tests bound: arg-max mismatches (count, pixels, the
reference's largest top-2 gap at a mismatch, in units of the logit gain), logit error, train-mode EPE, per-tensor
gradient errors, and the state after the Adam step against tests/test_oracle_golden.after_step_atol.
Run on the GPU box:  python tests/tools/parity_report.py > gpurun_out/parity.txt"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import conftest                                         # noqa: E402  (sets sys.path for the package and the oracle)
from conftest import GOLDEN_CASES, Golden               # noqa: E402
from test_oracle_golden import after_step_atol          # noqa: E402
from adaptive_stereo.adaptation import OnlineAdapter    # noqa: E402
from adaptive_stereo.models.stereo_net import StereoNet, FeatureExtractorNetwork   # noqa: E402
from adaptive_stereo.utils import synthetic as syn      # noqa: E402

DEV = "cuda:0"


def build(meta):
  fnet = FeatureExtractorNetwork(meta["k"])
  snet = StereoNet(meta["k"], 1, meta["s"], maxdisp=meta["maxdisp"])
  fnet.load_state_dict(syn.synthetic_state_dict(fnet.state_dict(), seed=123), strict=True)
  snet.load_state_dict(syn.synthetic_state_dict(snet.state_dict(), seed=123, logit_gain=meta["gain"]), strict=True)
  return fnet.to(DEV), snet.to(DEV)


def argmax_report(am, gold, scale):
  ref_am, gap = gold.full("train/argmax"), gold.full("train/top2gap")
  bad = am.cpu() != ref_am
  n = int(bad.sum())
  return {"pixels": int(bad.numel()), "mismatches": n,
          "max_gap_at_mismatch_over_gain": float(gap[bad].max()) / scale if n else 0.0,
          "ref_gap_min_over_gain": float(gap.min()) / scale,
          "pixels_with_gap_below_1e-6gain": int((gap <= 1e-6 * scale).sum()),
          "pixels_with_gap_below_2e-5gain": int((gap <= 2e-5 * scale).sum())}


def main():
  for case in GOLDEN_CASES:
    gold = Golden(case); meta = gold.meta
    k, s, scale = meta["k"], meta["s"], max(1.0, meta["gain"])
    rep = {"case": case, "gain": meta["gain"]}
    fnet, snet = build(meta)
    left, right = (t.to(DEV) for t in syn.stereo_pair(meta["B"], meta["H"], meta["W"], seed=1))
    # eval forward
    fnet.eval(); snet.eval()
    with torch.no_grad():
      out = snet(left, fnet(left), fnet(right), "l", output_cost_volume=True)
    exp, full = gold.expected("eval/logits")
    rep["eval_logit_err_over_gain"] = float((out["cost_volume_l/%d" % (s + k)].cpu() - exp).abs().max()) / scale
    exp, full = gold.expected("eval/pred_refined")
    got = out["pred_disp_l/%d" % s].cpu()
    got = got if full else syn.subsample(got, 4096)
    rep["eval_epe"] = float((got.reshape(exp.shape) - exp).abs().mean())
    rep["eval_max_err"] = float((got.reshape(exp.shape) - exp).abs().max())
    # train step
    adapter = OnlineAdapter(fnet, snet, meta["H"], meta["W"], lr=meta["lr"], clip_grad_norm=True)
    res = adapter.step(left, right)
    o = res["outputs"]
    logits = o["cost_volume_l/%d" % (s + k)]
    exp, _ = gold.expected("train/logits")
    rep["train_logit_err_over_gain"] = float((logits.detach().cpu() - exp).abs().max()) / scale
    rep["argmax"] = argmax_report(logits._as_argmax, gold, scale)
    exp, full = gold.expected("train/pred_refined")
    got = o["pred_disp_l/%d" % s].detach().cpu()
    got = got if full else syn.subsample(got, 4096)
    rep["train_epe"] = float((got.reshape(exp.shape) - exp).abs().mean())
    rep["train_max_err"] = float((got.reshape(exp.shape) - exp).abs().max())
    exp = gold.full("train/pred_coarse")
    rep["train_coarse_epe"] = float((adapter_pred_coarse(o, s, k, exp) - exp).abs().mean()) if False else None
    rep["loss_err"] = abs(float(res["loss"]) - gold.scalar("train/loss"))
    # gradients
    names = ("stereo", "feature")
    worst = []
    for mi, name, p, off, n in adapter.arena.entries:
      key = "grad/%s.%s" % (names[mi], name)
      if "%s.%s" % (names[mi], name) in gold.no_grad_keys or not gold.has(key):
        continue
      e, full = gold.expected(key)
      g = adapter.arena.grads[off:off + n].view(p.shape).detach().cpu()
      g = g if full else syn.subsample(g, 4096)
      den = float(e.double().norm())
      rel = float((g.reshape(e.shape).double() - e.double()).norm()) / max(den, 1e-30)
      worst.append((rel, key, float(e.abs().max())))
    worst.sort(reverse=True)
    rep["grad_rel_l2_worst5"] = [(round(r, 5), kk, "%.2e" % m) for r, kk, m in worst[:5]]
    real = [w for w in worst if w[2] >= 1e-6 * scale and not w[1].endswith(("conv2d_out.bias", "conv3d_alone.bias"))]
    rep["grad_rel_l2_worst5_nonnoise"] = [(round(r, 5), kk, "%.2e" % m) for r, kk, m in real[:5]]
    rep["grad_rel_l2_median"] = worst[len(worst) // 2][0]
    # whole-network gradient vectors (what the update direction depends on)
    for mi, nm in enumerate(names):
      num = den = 0.0
      for mj, name, p, off, n in adapter.arena.entries:
        key = "grad/%s.%s" % (names[mj], name)
        if mj != mi or "%s.%s" % (names[mj], name) in gold.no_grad_keys or not gold.has(key):
          continue
        e, full = gold.expected(key)
        g = adapter.arena.grads[off:off + n].view(p.shape).detach().cpu()
        g = g if full else syn.subsample(g, 4096)
        num += float((g.reshape(e.shape).double() - e.double()).pow(2).sum()); den += float(e.double().pow(2).sum())
      rep["grad_rel_l2_whole_%s" % nm] = (num / den) ** 0.5
    norm, ref_norm = float(adapter.optimizer.grad_norm()), gold.scalar("train/stereo_grad_norm")
    rep["grad_norm_rel_err"] = abs(norm - ref_norm) / ref_norm
    # state after the step
    coef = min(1.0, 1.0 / (ref_norm + 1e-6))
    over, moved, ref_moved, nchecked = [], 0.0, 0.0, 0
    bn_worst = bn_abs = 0.0
    init = {"stereo": syn.synthetic_state_dict(StereoNet(k, 1, s, maxdisp=meta["maxdisp"]).state_dict(), seed=123,
                                               logit_gain=meta["gain"]),
            "feature": syn.synthetic_state_dict(FeatureExtractorNetwork(k).state_dict(), seed=123)}
    for net_name, net in (("stereo", snet), ("feature", fnet)):
      for name, t in net.state_dict().items():
        key = "after/%s.%s" % (net_name, name)
        if name.endswith("num_batches_tracked") or not gold.has(key):
          continue
        e, full = gold.expected(key)
        got = t.detach().cpu()
        got = got if full else syn.subsample(got, 4096)
        got = got.reshape(e.shape).double()
        if name.endswith(("running_mean", "running_var")):
          d = (got - e.double()).abs()
          bn_abs = max(bn_abs, float(d.max()))
          bn_worst = max(bn_worst, float((d / (1e-4 + e.double().abs())).max()))
          continue
        atol = after_step_atol(gold, net_name, name, meta["lr"], coef, scale)
        err = (got - e.double()).abs()
        tol = atol + 1e-5 * e.double().abs()
        nchecked += err.numel()
        if bool((err > tol).any()):
          over.append((key, int((err > tol).sum()), err.numel(), float(err.max())))
        w0 = init[net_name][name]
        w0 = (w0 if full else syn.subsample(w0, 4096)).reshape(e.shape).double()
        moved += float((got - w0).pow(2).sum()); ref_moved += float((e.double() - w0).pow(2).sum())
    rep["after_weights_over_tol"] = over[:8]
    rep["after_weights_n_over"] = len(over)
    rep["after_weights_checked"] = nchecked
    rep["moved_norm"], rep["ref_moved_norm"] = moved ** 0.5, ref_moved ** 0.5
    rep["bn_running_worst_rel"] = bn_worst
    rep["bn_running_worst_abs"] = bn_abs
    print(json.dumps(rep), flush=True)
    del adapter, fnet, snet
    torch.cuda.empty_cache()


def adapter_pred_coarse(o, s, k, exp):
  return exp


if __name__ == "__main__":
  main()
