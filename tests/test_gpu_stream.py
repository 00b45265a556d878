"""BASELINE configs[4], the one-GPU leg: >= 1000 steps of continuous online adaptation on a synthetic stream through the whole
control plane (control.AdaptationLoop, mode VS+ER: FCS-EMA out-of-distribution gate, reservoir OVS with validation and
state transitions, experience replay with the Khamis loss) — no NaN / Inf anywhere, the counters consistent, and at steps
1, 10, 100 and 1000 the step's loss and FCS against the ORACLE loaded with the GPU's state of that moment (two fp32
implementations of this loss drift apart over hundreds of Adam steps; each checkpoint is compared from the same state)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import REPO, parity_note
from oracle import stereo_oracle as orc

sys.path.insert(0, os.path.join(REPO, "tests", "tools"))


def test_thousand_steps_of_continuous_adaptation():
  import adapt_stream
  from adaptive_stereo.control import State
  H, W, K, STEPS = 96, 256, 4, 1000
  checks = {}

  def on_step(i, loop, l, r):
    if i + 1 not in (1, 10, 100, 1000):
      return
    ad = loop.adapter
    training = loop.state_machine.state() == State.IN_PROGRESS
    fsd = {n: t.detach().cpu().clone() for n, t in ad.feature_net.state_dict().items()}
    ssd = {n: t.detach().cpu().clone() for n, t in ad.stereo_net.state_dict().items()}
    fp, sp = orc.make_params(fsd, False), orc.make_params(ssd, False)
    lc, rc = l.cpu(), r.cpu()
    with torch.no_grad():
      fl, fr = orc.feature_extractor(fp, lc, K, training), orc.feature_extractor(fp, rc, K, training)
      out = orc.stereo_forward(sp, lc, fl, fr, K, 0, 192, "l", training, True)
      loss, _, _ = orc.monodepth_single_loss(lc, rc, out["pred_disp_l/0"])
      fcs = orc.feature_contrast_mean(out["cost_volume_l/%d" % K]).mean()
    checks[i + 1] = (float(loss), float(fcs), training)

  loop, adapter, fnet, snet, trace, dt = adapt_stream.run(STEPS, H, W, mode="VS+ER", on_step=on_step)
  losses = torch.stack([t[0].float() for t in trace]).cpu()
  fcs = torch.stack([t[1].float() for t in trace]).cpu()
  ema = torch.stack([t[2].float() for t in trace]).cpu()
  assert bool(torch.isfinite(losses).all()) and bool(torch.isfinite(fcs).all()) and bool(torch.isfinite(ema).all())
  assert bool(torch.isfinite(adapter.arena.params).all()) and bool(torch.isfinite(adapter.optimizer.exp_avg_sq).all())
  for net in (fnet, snet):
    for name, b in net.named_buffers():
      assert bool(torch.isfinite(b.float()).all()), name
  updates = sum(1 for t in trace if t[4])
  added = sum(1 for t in trace if t[5])
  assert updates == loop.gradient_updates and updates > 100, updates
  assert 1 <= loop.state_machine.ovs_buffer_size() <= 8 and added >= loop.state_machine.ovs_buffer_size()
  assert adapter.optimizer.step_count == updates and float(adapter.optimizer.step_dev) == float(updates)
  # train-mode BatchNorm counters advance once per forward in IN_PROGRESS (adaptation + ER replay forwards)
  assert int(snet.filter[0][0].bn.num_batches_tracked) >= updates
  worst = (0.0, 0.0)
  for step, (ref_loss, ref_fcs, training) in checks.items():
    got_loss, got_fcs = float(losses[step - 1]), float(fcs[step - 1])
    worst = (max(worst[0], abs(got_loss - ref_loss)), max(worst[1], abs(got_fcs - ref_fcs) / max(1.0, abs(ref_fcs))))
    assert abs(got_loss - ref_loss) <= 2e-5 + 1e-4 * abs(ref_loss), (step, got_loss, ref_loss, training)
    assert abs(got_fcs - ref_fcs) <= 1e-4 * max(1.0, abs(ref_fcs)), (step, got_fcs, ref_fcs, training)
  assert set(checks) == {1, 10, 100, 1000}
  parity_note("stream_1000", steps=STEPS, gradient_updates=updates, added_to_ovs=added,
              steps_in_done_state=sum(1 for t in trace if t[3] == State.DONE), worst_loss_dev=worst[0], worst_fcs_rel_dev=worst[1],
              loss_first_50=float(losses[:50].mean()), loss_last_50=float(losses[-50:].mean()), pairs_per_s=STEPS / dt)
