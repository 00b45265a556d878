"""One online-adaptation step, as the reference's adapt.py performs it, on MI355X.

Reference sequence (adapt.py:304-396, NONSTOP mode; evaluation/stereonet_timing.py:44-72):
  feature_net(left), feature_net(right) -> stereo_net(..., output_cost_volume=True)
  -> LinearWarping(right, pred) -> monodepth_loss[0][mask].mean() -> FCS mean
  -> zero_grad, backward -> clip_grad_norm_(stereo_net.parameters(), 1.0) -> Adam.step()

What is different here (by design, not semantics):
  * parameters, gradients and both Adam moments live in FLAT fp32 arenas (stereo_net first, then
    feature_net — the reference's param-group order, adapt.py:208-209).  Module parameters are views
    into the arena, ``.grad`` tensors are views into the gradient arena, so autograd accumulates in
    place, the clip is one sum-of-squares kernel over a slice, Adam is one kernel per group, and —
    with more than one GPU — the whole gradient arena (+4 scalars) is ONE RCCL all-reduce over xGMI per step.
  * the masked mean is a fused reduction (no boolean-index host sync); the step issues no host sync.

Data parallel semantics (one process per GPU, ``torch.distributed`` backend "nccl" = RCCL):
  independent stereo pairs are sharded over ranks; the loss is the mean over the valid pixels of
  the WHOLE batch (adapt.py:83) = (sum over ranks of the local masked sums) / N_total.  Backward is linear in the
  incoming gradient, so each rank back-propagates its local SUM, gradients and the four step scalars (valid count, loss
  sum, FCS sum, FCS count) are all-reduced in one message, and the summed gradients are divided by N_total before
  clip + Adam.  BatchNorm statistics are per-replica
  (each rank normalises with its own shard's statistics), as in torch DDP without SyncBatchNorm.
"""
import weakref

import torch
import torch.distributed as dist

from . import _native as nat
from . import hip_ops
from .hip_ops import masked_mean, MaskedMeanFn
from .models.linear_warping import LinearWarping
from .utils.loss_functions import monodepth_loss
from .utils.feature_contrast import feature_contrast_mean
from .utils.ema import online_ema


def _release_comm(comm):
  """Fallback release of a communicator whose adapter was dropped (or is still alive at interpreter exit) without close().
  Ranks reach this point at different times, or not at all (a crashed rank), the process group may be gone and a captured
  hipGraph may still hold nodes of the communicator: a blocking destroy here could hang the surviving ranks, so the teardown is
  ncclCommAbort (does not wait for outstanding work or for peers).  close() is the orderly, collective release."""
  import sys
  try:
    print("adaptive_stereo: an OnlineAdapter with a live RCCL communicator was finalized without close(): aborting the "
          "communicator (call close() on every rank before dropping the adapter or destroying the process group)", file=sys.stderr)
    comm.abort()
  except Exception as e:                   # noqa: BLE001 — interpreter shutdown: report what can still be reported
    try:
      print("adaptive_stereo: ncclCommAbort in the finalizer failed: %r" % (e,), file=sys.stderr)
    except Exception:                      # noqa: BLE001
      pass


def never_executed_names(module):
  """Names (as in ``module.named_parameters()``) of the parameters of every BasicBlock.conv2 under ``module``: constructed but
  never called by the reference (stereo_net.py:40 against :44-51) — part of the state_dict, no gradient (``grad is None``:
  torch.optim.Adam skips them, adapt.py:208-210), never changed.  Keyed on the OWNING MODULE's type, not on a substring of
  the name: a layer of some other module that happens to be called conv2 trains and synchronises like any other."""
  from .models.stereo_net import _ResidualBlock2d as BasicBlock     # (the reference's BasicBlock: stereo_net.py:33-51)
  names = set()
  for mname, m in module.named_modules():
    if isinstance(m, BasicBlock):
      prefix = (mname + "." if mname else "") + "conv2."
      names.update(prefix + n for n, _ in m.conv2.named_parameters())
  return names


class FlatArena(object):
  """Re-homes the parameters of ``modules`` that take part in a step into one flat fp32 buffer (and a twin for gradients).
  The never-executed ``conv2.*`` tensors (111,744 floats at k=4) stay where they are, with ``grad = None`` as in the
  reference: they ride neither in the gradient all-reduce nor in the clip norm nor in Adam (313,698 floats do)."""

  def __init__(self, modules):
    self.entries = []          # (module index, name, param, offset, numel): the parameters that live in the arena
    self.group_bounds = []     # (start, end) per module, in floats
    self.all_params = []       # per module: [(name, param, in the arena?)] in named_parameters() order (adam.pth indices)
    offset = 0
    params = []
    for mi, m in enumerate(modules):
      start = offset
      listed = []
      dead = never_executed_names(m)
      for name, p in m.named_parameters():
        live = name not in dead
        listed.append((name, p, live))
        if not live:
          continue
        n = p.numel()
        self.entries.append((mi, name, p, offset, n))
        params.append(p)
        offset += n
        offset = (offset + 3) // 4 * 4        # keep every tensor 16-byte aligned
      self.all_params.append(listed)
      self.group_bounds.append((start, offset))
    self.numel = offset
    dev = params[0].device
    self.params = torch.zeros(self.numel, dtype=torch.float32, device=dev)
    # gradients and, behind them, the step's four scalars [valid-pixel count, loss sum, FCS sum, FCS count]: ONE
    # contiguous buffer, so that data-parallel stepping needs a single all-reduce per step (grads_and_scalars)
    self.grads_and_scalars = torch.zeros(self.numel + 4, dtype=torch.float32, device=dev)
    self.grads = self.grads_and_scalars[:self.numel]
    self.step_scalars = self.grads_and_scalars[self.numel:]
    with torch.no_grad():
      for _, _, p, off, n in self.entries:
        self.params[off:off + n].copy_(p.detach().reshape(-1))
        p.data = self.params[off:off + n].view(p.shape)
        p.grad = self.grads[off:off + n].view(p.shape)
        p._as_grad_sink = True     # hip_ops.grad_sinks: backward kernels accumulate straight into the arena

  def rebind_grads(self):
    """Makes sure every ``p.grad`` is still the arena view (user code may have set it to None)."""
    for _, _, p, off, n in self.entries:
      if p.grad is None or p.grad.data_ptr() != self.grads.data_ptr() + 4 * off:
        p.grad = self.grads[off:off + n].view(p.shape)

  def zero_grads(self):
    self.grads.zero_()


class FusedClipAdam(object):
  """clip_grad_norm_(group 0, max_norm) + Adam over the arena (as_sumsq / as_adam_step)."""

  def __init__(self, arena, lr, betas=(0.9, 0.999), eps=1e-8, clip_group=0, max_norm=1.0):
    self.arena, self.lr, self.betas, self.eps = arena, lr, betas, eps
    self.clip_group, self.max_norm = clip_group, max_norm
    dev = arena.params.device
    self.exp_avg = torch.zeros_like(arena.params)
    self.exp_avg_sq = torch.zeros_like(arena.params)
    self.step_count = 0
    self.step_dev = torch.zeros(1, dtype=torch.float32, device=dev)   # same count on the device (hipGraph replay)
    self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
    self.coef = torch.ones(1, dtype=torch.float32, device=dev)
    self.ws = torch.empty(nat.load().as_sumsq_workspace(arena.numel), dtype=torch.float32, device=dev)

  def step(self, clip=True):
    a = self.arena
    self.step_count += 1
    # the device-side step counter must move before the first Adam launch reads it: it rides on the clip's finalize launch
    # when the clipped group is the first one (the reference's: stereo_net), else it is an add of its own
    ride = bool(clip) and self.clip_group == 0
    if not ride:
      self.step_dev += 1.0
    for gi, (s, e) in enumerate(a.group_bounds):
      scale = None
      if clip and gi == self.clip_group:
        # torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1 — sum of squares, coefficient and
        # the step counter in two launches
        nat.call("as_sumsq_clip", nat.ptr(a.grads[s:e]), e - s, float(self.max_norm), nat.ptr(self.sumsq), nat.ptr(self.coef),
                 nat.ptr(self.step_dev) if (ride and gi == 0) else None, nat.ptr(self.ws), nat.stream())
        scale = self.coef
      nat.call("as_adam_step", nat.ptr(a.params[s:e]), nat.ptr(a.grads[s:e]), nat.ptr(self.exp_avg[s:e]),
               nat.ptr(self.exp_avg_sq[s:e]), e - s, nat.ptr(scale), self.lr, self.betas[0], self.betas[1],
               self.eps, self.step_count, nat.ptr(self.step_dev), nat.stream())

  def grad_norm(self):
    """Pre-clip L2 norm of the clipped group at the last step (device scalar)."""
    return torch.sqrt(self.sumsq)

  def state_dict(self):
    """torch.optim.Adam-compatible layout (what the reference writes to adam.pth, train.py:136-137): one
    entry per parameter that has received a gradient, param groups in arena order."""
    state, groups, index = {}, [], 0
    where = {id(p): (off, n) for _, _, p, off, n in self.arena.entries}
    for gi, listed in enumerate(self.arena.all_params):
      ids = []
      for name, p, live in listed:             # every parameter has an index; the never-executed ones have no state
        ids.append(index)
        if live and self.step_count > 0:
          off, n = where[id(p)]
          if float(self.exp_avg_sq[off:off + n].abs().sum()) > 0:
            state[index] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.exp_avg[off:off + n].view(p.shape).detach().cpu().clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + n].view(p.shape).detach().cpu().clone()}
        index += 1
      groups.append({"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                     "params": ids})
    return {"state": state, "param_groups": groups}


class OnlineAdapter(object):
  """feature_net + stereo_net + warper + optimiser bound together for the per-step sequence."""

  def __init__(self, feature_net, stereo_net, height, width, lr=5e-5, clip_grad_norm=True,
               smoothness_weight=1e-3, fcs_ema_weight=0.999, process_group=None, sync_bn=False,
               overlap_features=True, force_data_parallel=False, pair_features=True, native_collectives=True):
    self.feature_net, self.stereo_net = feature_net, stereo_net
    self.scale = stereo_net.input_scale
    self.coarse_scale = stereo_net.input_scale + stereo_net.k
    self.warper = LinearWarping(height, width)
    self.clip = clip_grad_norm
    self.sw = smoothness_weight
    self.fcs_ema_weight = fcs_ema_weight
    self.fcs_smoothed = None
    self.arena = FlatArena([stereo_net, feature_net])      # adapt.py:208-209 order
    self.optimizer = FusedClipAdam(self.arena, lr)
    self.pg = process_group
    self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
    # dp: the data-parallel step (local loss sums, one all-reduce of [gradients | 4 scalars], 1/N_total after it).  On
    # with more than one rank; force_data_parallel runs the very same sequence in a process group of ONE rank, which is
    # how a one-GPU box exercises RCCL's communicator, its stream ordering and the two-graph replay.
    self.dp = self.world > 1 or bool(force_data_parallel)
    if self.dp and not (dist.is_available() and dist.is_initialized()):
      raise RuntimeError("OnlineAdapter(force_data_parallel=True) needs an initialised torch.distributed process group")
    # sync_bn: train-mode BatchNorm over the batches of all ranks (= the reference's single-process batch); default is
    # per-replica statistics, as DistributedDataParallel without SyncBatchNorm.  Eager stepping only.
    # comm: this library's own RCCL communicator (rccl.py) — its collectives can be captured into a hipGraph, c10d's cannot
    # on this build; None (gloo groups, a failed creation, native_collectives=False) keeps torch.distributed collectives
    # outside any capture (two graphs per step)
    self.comm = None
    if self.dp and native_collectives:
      from . import rccl
      self.comm = rccl.try_create(process_group)
      if self.comm is not None:
        # close() is the orderly release (collective, before the process group goes); an adapter that is dropped or still alive
        # at interpreter exit releases its communicator through this finalizer instead of leaking it
        self._comm_finalizer = weakref.finalize(self, _release_comm, self.comm)
    self.bn_sync = hip_ops.BnSync(process_group, self.comm) if (sync_bn and (self.world > 1 or force_data_parallel)) else None
    dev = self.arena.params.device
    self.scalars = self.arena.step_scalars       # [valid count, loss sum, FCS sum, FCS count], behind the gradients
    self._graph = None
    self.plan = hip_ops.StepPlan()     # one-launch weight packing / batch counters (recorded on the first step)
    self.infer_plan = hip_ops.StepPlan()   # same for the eval-mode forward (+ all BatchNorm affines in one launch)
    self._infer_graph = None
    self._side = None                      # second stream for the right image's feature extraction
    self._capture_origin = None            # handle of the stream a capture opened by capture()/capture_infer() runs on
    self.fork_fallbacks = 0                # times _features_two_streams declined to fork (foreign or nested capture)
    self.overlap_features = overlap_features
    # train mode: both images of a pair through the feature extractor in ONE pass with two BatchNorm statistics groups
    # (FeatureExtractorNetwork.forward_pair); False: two passes, side by side on two streams (the round-2 arrangement)
    self.pair_features = pair_features
    self.infer_batched_features_max = 1 << 30     # pairs per call up to which inference batches left and right images

  # -- forward only: evaluate_model.py:52-60 / train.py:94-96 ------------------------------------
  @torch.no_grad()
  def infer(self, left, right):
    if self._infer_graph is not None:
      if left.data_ptr() != self._infer_left.data_ptr():
        self._infer_left.copy_(left)
      if right.data_ptr() != self._infer_right.data_ptr():
        self._infer_right.copy_(right)
      self._infer_graph.replay()
      return self._infer_result
    return self._infer_eager(left, right)

  @torch.no_grad()
  def _infer_eager(self, left, right):
    self.feature_net.eval(); self.stereo_net.eval()
    self.infer_plan.begin()
    try:
      fl, fr = self._features_eval(left, right)
      out = self.stereo_net(left, fl, fr, "l", output_cost_volume=True)
    finally:
      self.infer_plan.end()
    fcs = feature_contrast_mean(out["cost_volume_l/{}".format(self.coarse_scale)])
    return out, fcs

  def _features_eval(self, left, right):
    """Eval mode is stateless and batch-independent (bit for bit): both images go through the feature extractor as ONE
    batch — half the launches of a chain that is latency-bound (measured better than two streams at every batch size:
    +5 % at one pair, +1 % at eight)."""
    if left.shape[0] > self.infer_batched_features_max:
      return self._features_two_streams(left, right)
    both = self.feature_net(_adjacent_or_cat(left, right))
    return both[:left.shape[0]], both[left.shape[0]:]

  def _features_train(self, left, right):
    if self.pair_features and hip_ops.trunk_enabled() and hasattr(self.feature_net, "forward_pair"):
      return self.feature_net.forward_pair(left, right)
    return self._features_two_streams(left, right)

  def _features_two_streams(self, left, right):
    """The two feature extractions of a pair are independent and, at 1/16 resolution, far too small to fill the chip
    (a 24x78 map is 59 workgroups): the right image's runs on a second HIP stream next to the left one's.  Inside a
    captured graph the fork/join become two parallel branches."""
    if not self.overlap_features:
      return self.feature_net(left), self.feature_net(right)
    main = torch.cuda.current_stream()
    if torch.cuda.is_current_stream_capturing() and main.cuda_stream != self._capture_origin:
      # A capture this object did not open, or a stream that was itself forked inside one: a fork from an already
      # forked stream makes hipStreamEndCapture of ROCm 7.2 crash the process (seen once: DESIGN 4).  One stream, the
      # same kernels in the one-stream order — bit-identical results (test_two_stream_feature_extraction_equals_one_stream).
      self.fork_fallbacks += 1
      return self.feature_net(left), self.feature_net(right)
    if self._side is None:
      self._side = torch.cuda.Stream()
    self._side.wait_stream(main)
    fl = self.feature_net(left)
    with torch.cuda.stream(self._side):
      fr = self.feature_net(right)
    main.wait_stream(self._side)
    return fl, fr

  @torch.no_grad()
  def capture_infer(self, left, right, warmup=2):
    """Captures the eval-mode forward (~75 launches) into a hipGraph; infer() replays it from then on.  The graph
    reads the weights where they live (the flat arena), so it stays valid across adaptation steps."""
    self._refuse_nested_capture("capture_infer")
    pair = torch.cat([left, right])                      # one buffer: the two images are its halves, so the batched
    self._infer_left, self._infer_right = pair[:left.shape[0]], pair[left.shape[0]:]   # feature pass needs no copy
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
      for _ in range(max(2, warmup)):          # the first call records the plan, the second runs it
        self._infer_eager(self._infer_left, self._infer_right)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    self._capture_origin = side.cuda_stream
    try:
      with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):   # same stream as the warm-up: the
        self._infer_result = self._infer_eager(self._infer_left, self._infer_right)     # buffer pool is per stream
    finally:
      self._capture_origin = None
    self._infer_graph = graph
    return self

  # -- one adaptation step: adapt.py:304-396 (NONSTOP) --------------------------------------------
  def step(self, left, right):
    if self._graph is not None:
      return self._replay(left, right)
    return self._step_eager(left, right)

  def _forward_maps(self, left, right):
    if self.bn_sync is None:
      fl, fr = self._features_train(left, right)
    else:
      fl, fr = self.feature_net(left), self.feature_net(right)   # collectives inside: one stream, one order
    out = self.stereo_net(left, fl, fr, "l", output_cost_volume=True)
    pred = out["pred_disp_l/{}".format(self.scale)]
    if pred.shape[-2:] != (self.warper._height, self.warper._width):
      raise AssertionError("OnlineAdapter: images of %s, built for %dx%d" % (tuple(pred.shape[-2:]), self.warper._height,
                                                                                self.warper._width))
    # warp + monodepth loss + masked mean (adapt.py:78-86) as one autograd node: (mean, sum, valid count, warped, mask)
    mean, lsum, count, warped, _mask = hip_ops.MaskedPhotometricFn.apply(pred, left, right, self.sw)
    fcs_map = feature_contrast_mean(out["cost_volume_l/{}".format(self.coarse_scale)])
    return (mean, lsum, count), fcs_map, out, warped

  def _step_eager(self, left, right):
    self.feature_net.train(); self.stereo_net.train()
    self.arena.rebind_grads()
    self.arena.zero_grads()
    self.plan.begin()
    prev_sync = hip_ops.set_bn_sync(self.bn_sync)
    hip_ops.rmw_order_reset(True)        # two streams update the same gradient sinks / running statistics: keep order
    try:
      (loss, lsum, count), fcs_map, out, warped = self._forward_maps(left, right)

      if not self.dp:
        fcs = fcs_map.mean()
        loss.backward()
      else:
        loss, fcs = self._distributed_backward(lsum, count, fcs_map)
    finally:
      hip_ops.rmw_order_reset(False)
      hip_ops.set_bn_sync(prev_sync)
      self.plan.end()

    self.optimizer.step(clip=self.clip)
    # FCS EMA (adapt.py:356-359), in place so that a captured graph keeps updating the same tensor
    if self.fcs_smoothed is None:
      self.fcs_smoothed = fcs.detach().clone()
    else:
      self.fcs_smoothed.mul_(self.fcs_ema_weight).add_(fcs.detach(), alpha=1.0 - self.fcs_ema_weight)
    out["left_warped/{}".format(self.scale)] = warped
    return {"loss": loss.detach(), "fcs": fcs, "fcs_smoothed": self.fcs_smoothed, "outputs": out}

  # -- split step for the control plane (control.AdaptationLoop): the OOD gate sits between forward and backward
  def forward_loss(self, left, right, train=True, replay=None, er_loss_weight=0.05):
    """Forward + losses (+ FCS EMA).  With train=False the networks run in eval mode without gradients
    (State.DONE, adapt.py:309-311).  ``replay`` = (left, right, gt) adds the experience-replay Khamis
    term (adapt.py:339-349).

    Data parallel: the reported loss, replay loss and FCS are those of the WHOLE batch (all ranks' pairs), from one
    24-byte all-reduce of [valid count, loss sum, FCS sum, FCS count, replay gt count, replay loss sum] — the OOD gate
    and the state machine must decide identically on every rank, and the reference's loss is the masked mean over the
    valid pixels of the whole batch (adapt.py:83), not a mean of per-rank means."""
    from .utils.loss_functions import khamis_robust_loss
    self.feature_net.train(train); self.stereo_net.train(train)
    if train:
      self.arena.rebind_grads()
      self.arena.zero_grads()
    if train:
      self.plan.begin()
    prev_sync = hip_ops.set_bn_sync(self.bn_sync if train else None)
    two_streams = self.bn_sync is None or not train
    hip_ops.rmw_order_reset(two_streams)
    try:
      with torch.set_grad_enabled(train):
        if not train:
          fl, fr = self._features_eval(left, right)
        else:
          fl, fr = self._features_train(left, right) if two_streams else (self.feature_net(left), self.feature_net(right))
        out = self.stereo_net(left, fl, fr, "l", output_cost_volume=True)
        pred = out["pred_disp_l/{}".format(self.scale)]
        loss, lsum, count, warped, _mask = hip_ops.MaskedPhotometricFn.apply(pred, left, right, self.sw)
        backprop = loss
        replay_loss = None
        if replay is not None:
          rl, rr, rgt = replay
          rfl, rfr = self._features_train(rl, rr) if two_streams else (self.feature_net(rl), self.feature_net(rr))
          rout = self.stereo_net(rl, rfl, rfr, "l", output_cost_volume=True)
          replay_loss = khamis_robust_loss(rout["pred_disp_l/{}".format(self.scale)], rgt)
          backprop = loss + er_loss_weight * replay_loss
    finally:
      hip_ops.rmw_order_reset(False)
      hip_ops.set_bn_sync(prev_sync)
      if train:
        self.plan.end(final=False)
    fcs_map = feature_contrast_mean(out["cost_volume_l/{}".format(self.coarse_scale)])
    fcs = fcs_map.mean()
    dp_terms = None
    if self.dp:
      self._dp_local_sums(lsum, count, fcs_map)             # this rank's [count, loss sum, FCS sum, FCS count]
      six = torch.zeros(6, dtype=torch.float32, device=lsum.device)
      six[:4] = self.scalars
      if replay_loss is not None:
        # the UNCLAMPED local count travels: the reference's whole-batch denominator is max(sum_r n_r, 1)
        # (loss_functions.py:13), not sum_r max(n_r, 1) — a rank whose replay ground truth has no valid pixel adds 0
        # (its local khamis term is 0 as well: KhamisLossFn returns 0 for an empty mask)
        n_gt = (replay[2] > 0).sum().to(torch.float32)
        six[4] = n_gt; six[5] = replay_loss.detach() * n_gt
      self._all_reduce_small(six)
      loss = six[1] / six[0]
      fcs = six[2] / six[3]
      if replay_loss is not None:
        # whole-batch Khamis loss = (sum over ranks of the local sums) / (count over ranks); the gradient arena is
        # divided by the valid-pixel count N after its all-reduce, so this rank back-propagates
        # sum_r(monodepth) + w * (N / M) * sum_r(khamis) = ... + w * (N / M) * n_r * khamis_r
        replay_coef, replay_whole = replay_whole_batch_terms(six, n_gt, er_loss_weight)
        dp_terms = (lsum, replay_loss, replay_coef, six[0])
        replay_loss = replay_whole
      else:
        dp_terms = (lsum, None, None, six[0])
    if self.fcs_smoothed is None:
      self.fcs_smoothed = fcs.detach().clone()
    else:
      self.fcs_smoothed.mul_(self.fcs_ema_weight).add_(fcs.detach(), alpha=1.0 - self.fcs_ema_weight)
    out["left_warped/{}".format(self.scale)] = warped
    return {"loss": loss.detach(), "replay_loss": None if replay_loss is None else replay_loss.detach(),
            "backprop_loss": backprop, "dp_terms": dp_terms, "fcs": fcs, "fcs_smoothed": self.fcs_smoothed,
            "outputs": out}

  def backward_update(self, result):
    """backward + clip + Adam for a result of forward_loss(train=True) (adapt.py:381-394).  Data parallel: every rank
    back-propagates its local loss SUM (backward is linear in the incoming gradient), ONE all-reduce sums the gradient
    arena, and the division by the whole batch's valid-pixel count follows it — the same whole-batch masked mean that
    step() implements (_distributed_backward), not a mean of per-rank means."""
    self.plan.begin(resume=True)  # backward re-packs the (unchanged) weights: one launch
    prev_sync = hip_ops.set_bn_sync(self.bn_sync)
    hip_ops.rmw_order_reset(self.bn_sync is None)
    try:
      if self.dp:
        lsum, replay_loss, replay_coef, n_total = result["dp_terms"]
        if replay_loss is None:
          lsum.backward()
        else:
          torch.autograd.backward([lsum, replay_loss], [torch.ones_like(lsum), replay_coef.reshape(replay_loss.shape)])
      else:
        result["backprop_loss"].backward()
    finally:
      hip_ops.rmw_order_reset(False)
      hip_ops.set_bn_sync(prev_sync)
      self.plan.end()
    if self.dp:
      self._all_reduce_gradients()                             # the same single message step() sends
      self.arena.grads.div_(n_total)
    self.optimizer.step(clip=self.clip)

  @torch.no_grad()
  def validation_loss(self, left, right):
    """monodepth_single_loss in eval mode, as StateMachine.validate uses it (adapt.py:121-142)."""
    was_f, was_s = self.feature_net.training, self.stereo_net.training
    self.feature_net.eval(); self.stereo_net.eval()
    fl, fr = self._features_eval(left, right)
    out = self.stereo_net(left, fl, fr, "l", output_cost_volume=True)
    pred = out["pred_disp_l/{}".format(self.scale)]
    loss, lsum, count, _warped, _mask = hip_ops.MaskedPhotometricFn.apply(pred, left, right, self.sw)
    if self.dp:
      # every rank scores its own buffered pair; the state machine must see ONE number on all ranks: the masked mean
      # over the pairs of all ranks (buffers fill in lock-step: the OOD gate is decided on all-reduced scalars)
      two = torch.stack([lsum.detach(), count.detach()])
      self._all_reduce_small(two)
      loss = two[0] / two[1]
    self.feature_net.train(was_f); self.stereo_net.train(was_s)
    return float(loss)

  def close(self):
    """Releases the library's own RCCL communicator (collective: every rank calls it, before the process group goes)."""
    if self.comm is not None:
      torch.cuda.synchronize()
      self._graph = None                 # a captured graph holds nodes of this communicator
      self.comm.destroy()
      self.comm = None
      self._comm_finalizer.detach()
      if self.bn_sync is not None:
        self.bn_sync.comm = None

  def _all_reduce_small(self, t):
    if self.comm is not None:
      self.comm.all_reduce(t)
    else:
      dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)

  def _all_reduce_gradients(self):
    """THE collective of a data-parallel step: [gradient arena | 4 step scalars] summed over the ranks, one message."""
    if self.comm is not None:
      self.comm.all_reduce(self.arena.grads_and_scalars)       # on the current stream: a graph node under capture
    else:
      allreduce_gradients_and_scalars(self.arena, self.pg)

  # -- hipGraph capture of the whole step ------------------------------------------------------------
  def capture(self, left, right, warmup=3):
    """Captures one adaptation step (forward, loss, backward, clip, Adam, EMA: ~130 kernel launches) into
    hipGraphs and replays them from then on.  (capture_error_mode="thread_local": the process-group watchdog thread
    queries events while a capture is open; only this thread's calls have to be capture-safe.)  Every entry point of the C ABI only enqueues work on the current
    stream, so the capture sees them as plain kernel nodes; the Adam step count lives on the device.  Inputs
    are copied into static buffers before each replay.
    One GPU: a single graph.  Data parallel with this library's own RCCL communicator (self.comm): a single graph as
    well — the all-reduce (and, with sync_bn, the BatchNorm collectives) are graph nodes.  Data parallel over
    torch.distributed collectives (gloo, or no native communicator): two graphs (forward + backward of the local loss
    sum | 1/N_total scaling + clip + Adam + EMA) with the step's single all-reduce issued between them, outside any
    capture; cross-replica BatchNorm cannot be captured then."""
    if self.bn_sync is not None and self.comm is None:
      raise RuntimeError("OnlineAdapter.capture: cross-replica BatchNorm puts collectives inside forward and backward; "
                         "over torch.distributed collectives a step cannot be captured with sync_bn=True (step() runs eagerly)")
    self._refuse_nested_capture("capture")
    # one buffer, the two images its halves: the pair pass of the feature extractor then needs no concatenation copy
    pair = torch.cat([left, right])
    self._static_left, self._static_right = pair[:left.shape[0]], pair[left.shape[0]:]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
      for _ in range(max(1, warmup)):
        self._step_eager(self._static_left, self._static_right)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    self.optimizer.step_count_at_capture = self.optimizer.step_count
    self._capture_origin = side.cuda_stream
    prev_origin = hip_ops.set_fork_origin(side.cuda_stream)      # (the weight gradients launched beside the data gradients)
    try:
      self._capture_graphs(side)
    finally:
      self._capture_origin = None
      hip_ops.set_fork_origin(prev_origin)
    # capture only records: the python-side counter advanced, the device-side one did not
    self.optimizer.step_count = self.optimizer.step_count_at_capture
    return self

  def _refuse_nested_capture(self, what):
    if torch.cuda.is_current_stream_capturing():
      raise RuntimeError("OnlineAdapter.%s: the current stream is already being captured; a capture inside a capture "
                         "(and the stream fork it implies) crashes hipStreamEndCapture on ROCm 7.2 — capture from an "
                         "ordinary stream, or call step()/infer() inside your own capture (they then run the "
                         "one-stream order)" % what)

  def _capture_graphs(self, side):
    left, right = self._static_left, self._static_right
    if not self.dp or self.comm is not None:
      if self.dp:
        dist.barrier(group=self.pg)
      graph = torch.cuda.CUDAGraph()
      with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):   # the warm-up's stream: its
        self._static_result = self._step_eager(self._static_left, self._static_right)   # pooled buffers are reused
      self._graph = graph
    else:
      if self.pg is not None or dist.is_initialized():
        dist.barrier(group=self.pg)
      cap = side                         # forward and backward are captured on the same stream (autograd replays
      g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()              # backward where forward ran)
      with torch.cuda.graph(g1, stream=cap, capture_error_mode="thread_local"):
        self.feature_net.train(); self.stereo_net.train()
        self.arena.rebind_grads()
        self.arena.zero_grads()
        self.plan.begin()
        hip_ops.rmw_order_reset(True)
        try:
          (_, lsum, count), fcs_map, out, warped = self._forward_maps(self._static_left, self._static_right)
          self._dp_local_sums(lsum, count, fcs_map)
          self._dp_backward(lsum)
        finally:
          hip_ops.rmw_order_reset(False)
          self.plan.end()
      torch.cuda.synchronize()
      allreduce_gradients_and_scalars(self.arena, self.pg)
      torch.cuda.synchronize()
      with torch.cuda.graph(g2, pool=g1.pool(), stream=cap, capture_error_mode="thread_local"):
        loss, fcs = self._dp_results()
        self.optimizer.step(clip=self.clip)
        self.fcs_smoothed.mul_(self.fcs_ema_weight).add_(fcs.detach(), alpha=1.0 - self.fcs_ema_weight)
        out["left_warped/{}".format(self.scale)] = warped
        self._static_result = {"loss": loss.detach(), "fcs": fcs, "fcs_smoothed": self.fcs_smoothed, "outputs": out}
      self._graph = (g1, g2)

  def graph_count(self):
    """hipGraphs a captured step replays (0: eager stepping)."""
    if self._graph is None:
      return 0
    return len(self._graph) if isinstance(self._graph, tuple) else 1

  def graph_inputs(self):
    """The captured step's own input buffers (after capture()): a producer that decodes or copies the next pair
    straight into them — and then passes them to step() — saves the two device copies a replay otherwise starts with."""
    return self._static_left, self._static_right

  def infer_inputs(self):
    """Same for the captured inference graph (after capture_infer())."""
    return self._infer_left, self._infer_right

  def _replay(self, left, right):
    if left.data_ptr() != self._static_left.data_ptr():
      self._static_left.copy_(left)
    if right.data_ptr() != self._static_right.data_ptr():
      self._static_right.copy_(right)
    if not isinstance(self._graph, tuple):
      self._graph.replay()
    else:
      g1, g2 = self._graph
      g1.replay()
      allreduce_gradients_and_scalars(self.arena, self.pg)
      g2.replay()
    self.optimizer.step_count += 1          # host mirror of the device-side counter
    return self._static_result

  # -- data-parallel step in three device-side phases with the two collectives between them ---------------
  def _dp_local_sums(self, lsum, count, fcs_map):
    """Phase 1 tail: this rank's [valid count, loss sum, FCS sum, FCS count] into self.scalars (no communication)."""
    fill_step_scalars(self.scalars, lsum.detach(), count.detach(), fcs_map.sum(), float(fcs_map.numel()))

  def _dp_backward(self, lsum):
    """Phase 1 tail: the gradient of this rank's masked loss SUM.  The whole-batch masked mean of the reference
    (adapt.py:83) is (sum over ranks of these sums) / N_total, and backward is linear in the incoming gradient: the
    division by N_total waits until gradients and counts have been all-reduced together."""
    lsum.backward()
    hip_ops.flush_deferred_reductions()          # the gradients must be complete before the all-reduce reads them

  def _dp_results(self):
    """Phase 2 head (after the all-reduce): scales the summed gradients by 1/N_total; returns (loss, FCS)."""
    s = self.scalars
    self.arena.grads.div_(s[0])
    return s[1] / s[0], s[2] / s[3]

  def _distributed_backward(self, lsum, count, fcs_map):
    self._dp_local_sums(lsum, count, fcs_map)
    self._dp_backward(lsum)
    self._all_reduce_gradients()
    return self._dp_results()


def replay_whole_batch_terms(six, n_gt_local, er_loss_weight):
    """The experience-replay (Khamis) term under data parallelism.  ``six`` = the all-reduced
    [valid count N, loss sum, FCS sum, FCS count, replay ground-truth count M (UNCLAMPED per rank), replay loss sum].
    Returns (the coefficient this rank's local khamis MEAN is back-propagated with, the whole-batch replay loss).
    Whole batch: khamis = (sum_r S_r) / max(sum_r n_r, 1) (loss_functions.py:13).  The gradient arena is divided by N after
    its all-reduce, and a rank holds its local mean S_r / max(n_r, 1), so the rank back-propagates
    w * (N / M) * n_r * mean_r with M = max(sum_r n_r, 1); a rank without a valid ground-truth pixel contributes 0."""
    m_total = six[4].clamp(min=1.0)
    return er_loss_weight * six[0] / m_total * n_gt_local, six[5] / m_total


_adjacent_or_cat = hip_ops.adjacent_or_cat


def fill_step_scalars(buf, loss_sum, valid_count, fcs_sum, fcs_count):
    buf[0] = valid_count; buf[1] = loss_sum; buf[2] = fcs_sum
    buf[3:4].fill_(fcs_count)        # a python float: fill_ is a kernel (capturable), item assignment a host copy
    return buf


def allreduce_step_scalars(buf, loss_sum, valid_count, fcs_sum, fcs_count, group=None):
    """One 16-byte all-reduce(sum) of [valid-pixel count, loss sum, FCS sum, FCS count], issued BEFORE
    backward: the reference's loss is the mean over the valid pixels of the whole batch (adapt.py:83),
    so every rank must scale its local gradient by 1/N_total, not 1/N_rank."""
    fill_step_scalars(buf, loss_sum, valid_count, fcs_sum, fcs_count)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf


_COMM_STREAMS = {}


def _all_reduce_on_comm_stream(flat, group):
    """all-reduce(sum) of a CUDA tensor on a HIP stream of its own, ordered against the current stream by events.
    c10d issues a blocking collective on the CURRENT stream and its watchdog thread keeps polling the work's end event;
    when that stream then starts a graph capture (the step's second graph follows the collective on the capture stream)
    the poll fails with hipErrorCapturedEvent and invalidates the capture — seen as a sporadic failure of capture() under
    RCCL.  A stream that never captures keeps the collective's events away from the graphs."""
    if not flat.is_cuda:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        return
    cur = torch.cuda.current_stream(flat.device)
    comm = _COMM_STREAMS.get(flat.device.index)
    if comm is None:
        comm = _COMM_STREAMS[flat.device.index] = torch.cuda.Stream(flat.device)
    comm.wait_stream(cur)
    with torch.cuda.stream(comm):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    cur.wait_stream(comm)


def allreduce_gradients_and_scalars(arena, group=None):
    """THE collective of a data-parallel adaptation step: one all-reduce(sum) of the flat gradient arena with the
    step's four scalars riding behind it (313,702 floats at k=4; latency-bound on xGMI, hence one message)."""
    _all_reduce_on_comm_stream(arena.grads_and_scalars, group)
    return arena.grads_and_scalars


def allreduce_gradients(flat_grads, group=None):
    """ONE all-reduce(sum) of the flat gradient arena (313,698 floats at k=4): on xGMI this message is
    latency-bound, so a single bucket beats per-tensor or per-layer buckets."""
    _all_reduce_on_comm_stream(flat_grads, group)
    return flat_grads
