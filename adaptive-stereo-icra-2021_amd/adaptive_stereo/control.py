"""Adaptation control plane: states, online validation set (OVS), FCS-EMA out-of-distribution gate,
experience replay.  SURVEY.md §8f-2.

Reference: adapt.py — ``State`` (:34-37), ``StateMachine`` (:89-172) and the per-batch logic of
``adapt()`` (:290-396).  This is host-side control; every tensor op it triggers is one of the HIP
operators.  Differences in form: validation takes the loss function as a callable, the OOD decision is
the only place that reads a device scalar back (and only in the VS / VS+ER modes, which need the
decision on the host to route the pair into the reservoir), and the per-step work is delegated to
``OnlineAdapter``.
"""
from enum import Enum

import torch

from .utils.stereo_reservoir import StereoReservoir

MODES = ("NONSTOP", "VS", "ER", "VS+ER", "NONE")


class State(Enum):
  DONE = 0          # adaptation finished: no gradient updates
  IN_PROGRESS = 1   # adapting
  VALIDATION = 2    # scoring the OVS: gradients off


class StateMachine(object):
  def __init__(self, initial_state, ovs_buffer_size=8, verbose=False):
    self.initial_state = initial_state
    self.current_state = initial_state
    self.ovs = StereoReservoir(ovs_buffer_size)
    self.prev_ovs_loss = float("inf")
    self.ovs_did_change = True
    self.ovs_iters_without_improvement = 0
    self.verbose = verbose

  def state(self):
    return self.current_state

  def ovs_buffer_size(self):
    return self.ovs.size()

  def restart(self):
    self.current_state = self.initial_state

  def add_to_ovs(self, left_img, right_img, loss, batch_idx):
    """Offers a (novel) pair to the reservoir; a DONE machine is restarted by any offer (adapt.py:101-115)."""
    did_add = self.ovs.add(left_img.detach(), right_img.detach(), loss.detach() if torch.is_tensor(loss) else loss,
                           batch_idx)
    if did_add:
      self.ovs_did_change = True
    if self.current_state == State.DONE:
      self.restart()
    return did_add

  def validate(self, loss_fn):
    """Re-scores every buffered pair with the current weights.  ``loss_fn(left, right) -> float`` must run
    the networks in eval mode without gradients (adapt.py:121-142)."""
    for i in range(self.ovs.size()):
      _, _, left, right = self.ovs.buf[i]
      self.ovs.update_value(i, float(loss_fn(left, right)))

  def transition(self, val_improve_retries):
    """adapt.py:144-166: stop when the OVS loss did not improve for `val_improve_retries` validations in a
    row while the buffer was unchanged; otherwise (improved, or buffer changed) keep adapting."""
    ovs_loss = float(self.ovs.average_value())
    if ovs_loss >= self.prev_ovs_loss and not self.ovs_did_change:
      self.ovs_iters_without_improvement += 1
      if self.ovs_iters_without_improvement >= val_improve_retries:
        self.current_state = State.DONE
        self.prev_ovs_loss = float("inf")
    else:
      self.ovs_did_change = False
      self.ovs_iters_without_improvement = 0
      self.prev_ovs_loss = ovs_loss
    return self.current_state


class AdaptationLoop(object):
  """The body of the reference's ``for inputs in adapt_loader`` (adapt.py:290-396) around an OnlineAdapter.

  mode                NONSTOP | VS | ER | VS+ER | NONE
  ovs_validate_hz     validate the OVS every this many steps (while IN_PROGRESS and non-empty)
  val_improve_retries see StateMachine.transition
  ood_threshold       a pair is novel when the smoothed FCS is below it
  er_loss_weight      weight of the Khamis loss on the replayed training pair (ER modes)
  """

  def __init__(self, adapter, mode="NONSTOP", ovs_buffer_size=10, ovs_validate_hz=100, val_improve_retries=1,
               ood_threshold=15.0, er_loss_weight=0.05):
    if mode not in MODES:
      raise ValueError("adapt_mode must be one of %s" % (MODES,))
    self.adapter = adapter
    self.mode = mode
    self.ovs_validate_hz = ovs_validate_hz
    self.val_improve_retries = val_improve_retries
    self.ood_threshold = ood_threshold
    self.er_loss_weight = er_loss_weight
    initial = State.DONE if mode == "NONE" else State.IN_PROGRESS
    self.state_machine = StateMachine(initial, ovs_buffer_size=ovs_buffer_size)
    self.step = 0
    self.gradient_updates = 0

  def _validation_loss(self, left, right):
    return self.adapter.validation_loss(left, right)

  def process(self, left, right, batch_idx, replay=None):
    """One batch.  ``replay`` = (left, right, gt_disp) of a training-domain pair for the ER modes."""
    sm = self.state_machine
    if (self.step % self.ovs_validate_hz == 0) and sm.ovs_buffer_size() > 0 and sm.state() == State.IN_PROGRESS:
      sm.validate(self._validation_loss)
      if self.mode not in ("NONSTOP", "ER", "NONE"):
        sm.transition(self.val_improve_retries)

    adapting = sm.state() == State.IN_PROGRESS
    use_replay = self.mode in ("ER", "VS+ER") and replay is not None
    gate = self.mode not in ("NONSTOP", "ER", "NONE")
    # Forward (+ loss) first; whether the backward/optimizer part runs is decided after the OOD gate.
    result = self.adapter.forward_loss(left, right, train=adapting, replay=replay if use_replay else None,
                                       er_loss_weight=self.er_loss_weight)
    did_add = False
    if gate:
      novel = float(result["fcs_smoothed"]) < self.ood_threshold          # the one host read-back (VS modes)
      if novel:
        did_add = bool(sm.add_to_ovs(left, right, result["loss"], batch_idx))
    updated = False
    if sm.state() == State.IN_PROGRESS and adapting and not did_add:
      self.adapter.backward_update(result)
      self.gradient_updates += 1
      updated = True
    self.step += 1
    result.update(state=sm.state(), added_to_ovs=did_add, updated=updated)
    return result
