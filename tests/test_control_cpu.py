"""Host-side control plane on CPU: the state machine's transitions follow the reference's
rules (adapt.py:89-172), and the option surface matches the reference's defaults (train.py:246-301)."""
import random

import torch

from adaptive_stereo.control import State, StateMachine, AdaptationLoop, MODES
import train as train_surface


def _pair(i):
  return torch.full((1, 3, 2, 2), float(i)), torch.full((1, 3, 2, 2), float(-i))


def test_state_machine_stops_after_retries_without_improvement_and_restarts_on_novelty():
  random.seed(0)
  sm = StateMachine(State.IN_PROGRESS, ovs_buffer_size=3)
  assert sm.state() == State.IN_PROGRESS and sm.ovs_buffer_size() == 0
  for i in range(3):
    l, r = _pair(i)
    assert sm.add_to_ovs(l, r, torch.tensor(1.0), i) is True
  assert sm.add_to_ovs(*_pair(1), torch.tensor(1.0), 1) is False          # duplicate index refused
  assert sm.ovs_buffer_size() == 3 and sm.ovs_did_change

  losses = iter([1.0, 1.0, 1.0])
  sm.validate(lambda l, r: next(losses))
  assert sm.transition(2) == State.IN_PROGRESS                 # buffer changed -> baseline recorded
  assert sm.prev_ovs_loss == 1.0 and not sm.ovs_did_change

  sm.validate(lambda l, r: 0.5)                                 # improved
  assert sm.transition(2) == State.IN_PROGRESS and sm.prev_ovs_loss == 0.5 and sm.ovs_iters_without_improvement == 0

  sm.validate(lambda l, r: 0.7)                                 # worse, buffer unchanged: strike 1
  assert sm.transition(2) == State.IN_PROGRESS and sm.ovs_iters_without_improvement == 1
  sm.validate(lambda l, r: 0.7)                                 # strike 2 -> DONE, baseline reset
  assert sm.transition(2) == State.DONE and sm.prev_ovs_loss == float("inf")

  # a novel pair offered while DONE restarts adaptation, whether or not the reservoir keeps it
  sm.add_to_ovs(*_pair(99), torch.tensor(2.0), 99)
  assert sm.state() == State.IN_PROGRESS


def test_validate_rescoring_uses_buffered_pairs():
  sm = StateMachine(State.IN_PROGRESS, ovs_buffer_size=4)
  for i in range(4):
    sm.add_to_ovs(*_pair(i), torch.tensor(0.0), i)
  seen = []
  sm.validate(lambda l, r: (seen.append(float(l.mean())), float(l.mean()) * 2)[1])
  assert seen == [0.0, 1.0, 2.0, 3.0]
  assert sm.ovs.average_value() == 3.0


class _FakeAdapter(object):
  """Records what AdaptationLoop asks for; no GPU."""

  def __init__(self, fcs_values):
    self.fcs = iter(fcs_values)
    self.calls = []

  def forward_loss(self, left, right, train=True, replay=None, er_loss_weight=0.05):
    self.calls.append(("fwd", train, replay is not None))
    return {"loss": torch.tensor(0.4), "fcs_smoothed": torch.tensor(next(self.fcs)), "backprop_loss": None, "outputs": {}}

  def backward_update(self, result):
    self.calls.append(("bwd",))

  def validation_loss(self, left, right):
    return 0.4


def test_adaptation_loop_modes():
  l, r = _pair(1)
  # NONSTOP: every step is a gradient step, nothing is gated
  ad = _FakeAdapter([1.0] * 5)
  loop = AdaptationLoop(ad, mode="NONSTOP")
  for i in range(5):
    res = loop.process(l, r, i)
    assert res["updated"] and not res["added_to_ovs"]
  assert loop.gradient_updates == 5

  # VS: a novel pair (smoothed FCS below threshold) goes to the OVS and is NOT trained on (adapt.py:367-394)
  ad = _FakeAdapter([20.0, 5.0, 5.0, 20.0])
  loop = AdaptationLoop(ad, mode="VS", ovs_buffer_size=2, ood_threshold=10.0, ovs_validate_hz=1000)
  flags = [loop.process(l, r, i) for i in range(4)]
  assert [f["added_to_ovs"] for f in flags] == [False, True, True, False]
  assert [f["updated"] for f in flags] == [True, False, False, True]
  assert loop.state_machine.ovs_buffer_size() == 2

  # NONE: never adapts, runs the networks in eval mode
  ad = _FakeAdapter([1.0] * 3)
  loop = AdaptationLoop(ad, mode="NONE")
  for i in range(3):
    assert not loop.process(l, r, i)["updated"]
  assert all(c == ("fwd", False, False) for c in ad.calls)

  # ER: the replay pair is forwarded too
  ad = _FakeAdapter([1.0] * 2)
  loop = AdaptationLoop(ad, mode="ER")
  loop.process(l, r, 0, replay=(l, r, torch.ones(1, 1, 2, 2)))
  assert ad.calls[0] == ("fwd", True, True)
  assert set(MODES) == {"NONSTOP", "VS", "ER", "VS+ER", "NONE"}


def test_train_options_match_reference_defaults():
  opt = train_surface.TrainOptions().parse([])
  assert (opt.height, opt.width, opt.stereonet_k, opt.stereonet_input_scale) == (320, 960, 3, 0)
  assert (opt.batch_size, opt.learning_rate, opt.clip_grad_norm) == (2, 1e-5, False)
  assert (opt.ovs_buffer_size, opt.ovs_validate_hz, opt.val_improve_retries, opt.eval_hz) == (10, 100, 1, 1000)
  assert (opt.er_loss_weight, opt.ood_threshold, opt.fcs_ema_weight, opt.smoothness_weight) == (0.05, 15.0, 0.999, 1e-3)
  # the paper's canonical run (experiments/adaptation/adapt_vs.sh:5-27)
  opt = train_surface.TrainOptions().parse("--height 320 --width 960 --batch_size 1 --learning_rate 5e-5 --stereonet_k 4 "
                                           "--clip_grad_norm --num_steps 4000 --ovs_buffer_size 16 --ovs_validate_hz 200 "
                                           "--val_improve_retries 2 --adapt_mode VS".split())
  assert opt.adapt_mode == "VS" and opt.clip_grad_norm and opt.stereonet_k == 4
