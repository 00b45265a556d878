"""Data-parallel semantics of the adaptation step, on CPU with gloo (world size 2).

Covers the host logic of adaptive_stereo.adaptation that a multi-GPU run relies on: FlatArena
(parameters / gradients as views of flat buffers, autograd accumulating in place), the scalar
all-reduce issued before backward, and the single flat-bucket gradient all-reduce.  The compute is
the oracle (CPU) in eval-mode BatchNorm, for which sharding pairs over ranks is exactly equivalent
to one process with the whole batch: sum_r grad(sum_r/N_total) == grad(whole-batch masked mean).
"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO, PKG


def _free_port():
  s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
  return p


def _build():
  from adaptive_stereo.models.stereo_net import StereoNet, FeatureExtractorNetwork
  from adaptive_stereo.utils import synthetic as syn
  k, maxdisp = 3, 64
  fnet, snet = FeatureExtractorNetwork(k), StereoNet(k, 1, 0, maxdisp=maxdisp)
  fsd = syn.synthetic_state_dict(fnet.state_dict(), seed=123)
  ssd = syn.synthetic_state_dict(snet.state_dict(), seed=123, logit_gain=5.0)
  left, right = syn.stereo_pair(2, 48, 64, seed=4, disparities=(3.0, 6.0))
  return k, maxdisp, fsd, ssd, left, right


def _loss_terms(fp, sp, left, right, k, maxdisp):
  """Per-rank pieces of monodepth_single_loss with eval-mode BN: (loss map, mask, fcs map)."""
  from oracle import stereo_oracle as orc
  fl = orc.feature_extractor(fp, left, k, False)
  fr = orc.feature_extractor(fp, right, k, False)
  out = orc.stereo_forward(sp, left, fl, fr, k, 0, maxdisp, "l", False, True)
  pred = out["pred_disp_l/0"]
  warped, mask = orc.linear_warp(right, pred, True)
  total = orc.monodepth_loss(pred, left, warped, 1e-3)[0]
  return total, mask, orc.feature_contrast_mean(out["cost_volume_l/%d" % k])


class _Holder(torch.nn.Module):
  """Wraps an oracle parameter dict as a Module so FlatArena can re-home it."""

  def __init__(self, params):
    super().__init__()
    self.names = []
    for i, (name, t) in enumerate(params.items()):
      if t.requires_grad:
        self.register_parameter("p%d" % i, torch.nn.Parameter(t.detach().clone()))
        self.names.append((name, "p%d" % i))

  def as_dict(self, template):
    out = dict(template)
    for name, attr in self.names:
      out[name] = getattr(self, attr)
    return out


def _worker(rank, world, port, result_path):
  for p in (REPO, PKG):
    if p not in sys.path:
      sys.path.insert(0, p)
  os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  torch.set_num_threads(2)
  from adaptive_stereo.adaptation import FlatArena, fill_step_scalars, allreduce_gradients_and_scalars
  from oracle import stereo_oracle as orc
  k, maxdisp, fsd, ssd, left, right = _build()
  fp0, sp0 = orc.make_params(fsd, True), orc.make_params(ssd, True)
  hs, hf = _Holder(sp0), _Holder(fp0)
  arena = FlatArena([hs, hf])                       # stereo first, then feature (adapt.py:208-209)
  sp, fp = hs.as_dict(sp0), hf.as_dict(fp0)
  # parameters are views of the arena
  assert all(p.data_ptr() >= arena.params.data_ptr() for p in list(hs.parameters()) + list(hf.parameters()))

  lo, hi = rank * (2 // world), (rank + 1) * (2 // world)
  total, mask, fcs = _loss_terms(fp, sp, left[lo:hi], right[lo:hi], k, maxdisp)
  # the product's recipe (OnlineAdapter._distributed_backward): gradient of this rank's masked loss SUM, then ONE
  # all-reduce of [gradients | valid count, loss sum, FCS sum, FCS count], then the division by N_total
  arena.zero_grads()
  s = fill_step_scalars(arena.step_scalars, (total.detach() * mask).sum(), mask.sum().float(), fcs.sum(), float(fcs.numel()))
  total.backward(mask.float())
  # autograd accumulated into the arena views in place
  assert float(arena.grads.abs().sum()) > 0
  assert arena.grads_and_scalars.numel() == arena.numel + 4 and arena.step_scalars.data_ptr() == arena.grads.data_ptr() + 4 * arena.numel
  allreduce_gradients_and_scalars(arena)
  arena.grads.div_(s[0])
  if rank == 0:
    torch.save({"grads": arena.grads.clone(), "loss": float(s[1] / s[0]), "fcs": float(s[2] / s[3]),
                "bounds": arena.group_bounds, "n": arena.numel}, result_path)
  dist.barrier()
  dist.destroy_process_group()


def test_two_rank_gradient_equals_whole_batch(tmp_path):
  from adaptive_stereo.adaptation import FlatArena
  from oracle import stereo_oracle as orc
  result = str(tmp_path / "dp.pt")
  mp.spawn(_worker, args=(2, _free_port(), result), nprocs=2, join=True)
  got = torch.load(result)

  # single process, whole batch
  k, maxdisp, fsd, ssd, left, right = _build()
  fp0, sp0 = orc.make_params(fsd, True), orc.make_params(ssd, True)
  hs, hf = _Holder(sp0), _Holder(fp0)
  arena = FlatArena([hs, hf])
  sp, fp = hs.as_dict(sp0), hf.as_dict(fp0)
  total, mask, fcs = _loss_terms(fp, sp, left, right, k, maxdisp)
  loss = total[mask].mean()
  arena.zero_grads()
  loss.backward()
  assert got["n"] == arena.numel and got["bounds"] == arena.group_bounds
  assert abs(got["loss"] - float(loss)) < 1e-6
  assert abs(got["fcs"] - float(fcs.mean())) < 1e-5
  ref = arena.grads
  err = float((got["grads"] - ref).norm() / ref.norm())
  assert err < 1e-5, "2-rank all-reduced gradient differs from the whole-batch gradient: rel %.2e" % err


def test_flat_arena_views_and_rebinding():
  from adaptive_stereo.adaptation import FlatArena
  a, b = torch.nn.Linear(3, 5), torch.nn.Linear(5, 2)
  w0 = a.weight.detach().clone()
  arena = FlatArena([a, b])
  assert torch.equal(a.weight, w0)
  assert arena.numel % 4 == 0 and len(arena.group_bounds) == 2
  assert arena.group_bounds[0][0] == 0 and arena.group_bounds[1][0] == arena.group_bounds[0][1]
  for _, _, p, off, n in arena.entries:
    assert off % 4 == 0 and p.data_ptr() == arena.params.data_ptr() + 4 * off
    assert p.grad.data_ptr() == arena.grads.data_ptr() + 4 * off
  b(a(torch.ones(1, 3))).sum().backward()
  assert float(arena.grads.abs().sum()) > 0
  a.weight.grad = None                       # e.g. optimizer.zero_grad(set_to_none=True) in user code
  arena.rebind_grads()
  assert a.weight.grad.data_ptr() == arena.grads.data_ptr()
  arena.params[: a.weight.numel()] += 1.0    # an update of the arena is an update of the module
  assert torch.allclose(a.weight, w0 + 1.0)


def test_replay_term_denominator_is_clamped_after_the_reduction():
  """Data-parallel experience replay: a rank whose replay ground truth has NO valid pixel must not inflate the whole-batch
  denominator (the reference divides by max(sum_r n_r, 1), loss_functions.py:13), nor contribute a gradient."""
  from adaptive_stereo.adaptation import replay_whole_batch_terms
  w = 0.05
  for counts in ((0.0, 17.0), (5.0, 3.0), (0.0, 0.0)):
    sums = [2.5 * n for n in counts]                      # local khamis SUMS
    local_mean = [s_ / max(n, 1.0) for s_, n in zip(sums, counts)]
    six = torch.tensor([1000.0, 1.0, 1.0, 1.0, sum(counts), sum(m * n for m, n in zip(local_mean, counts))])
    M = max(sum(counts), 1.0)
    total_grad_weight = 0.0
    for r, n in enumerate(counts):
      coef, whole = replay_whole_batch_terms(six, torch.tensor(n), w)
      assert abs(float(whole) - sum(sums) / M) < 1e-6
      total_grad_weight += float(coef) * local_mean[r]     # what this rank back-propagates (before the division by N)
      if n == 0.0:
        assert float(coef) == 0.0
    assert abs(total_grad_weight / 1000.0 - w * sum(sums) / M) < 1e-6


# ---------------------------------------------------------------------------------------------------------------------
# The staged construction of the library's own RCCL communicator (adaptive_stereo/rccl.py: _create_staged), driven over a
# gloo group with stand-ins for the five RCCL stages: a failure injected at EVERY stage, on either rank, must leave both
# ranks with None — together, within the time limit, having run the same sequence of process-group collectives (a rank that
# left early would make the trailing barrier below hang or fail).  Round 3's constructor raised out of ncclGetUniqueId on
# rank 0 before the broadcast the other ranks were already waiting in.
# ---------------------------------------------------------------------------------------------------------------------
class _FakeComm(object):
  def __init__(self, rank, world):
    self.rank, self.world, self.destroyed = rank, world, False

  def destroy(self):
    self.destroyed = True


class _FakeStages(object):
  agree_device = "cpu"

  def __init__(self):
    self.calls = []

  def load(self):
    self.calls.append("load")

  def unique_id(self):
    self.calls.append("unique_id")
    return bytes(range(128))

  def prepare(self):
    self.calls.append("prepare")
    return torch.ones(1)

  def agree_flag(self):
    return torch.zeros(1, dtype=torch.int32)

  def init(self, uid, group, rank, world):
    assert uid == bytes(range(128))
    self.calls.append("init")
    return _FakeComm(rank, world)

  def probe(self, comm, t):
    self.calls.append("probe")
    dist.all_reduce(t)                      # a collective of the communicator: entered by every rank or by none
    assert int(t) == comm.world


RCCL_STAGES = ("load", "unique_id", "receive", "prepare", "init", "probe")


def _staged_worker(rank, world, port, out_dir):
  import datetime
  for p in (REPO, PKG):
    if p not in sys.path:
      sys.path.insert(0, p)
  os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
  dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
  from adaptive_stereo import rccl
  rows = []
  cases = [None] + ["%s:%d" % (s_, r_) for s_ in RCCL_STAGES for r_ in range(world) if not (s_ == "unique_id" and r_ != 0)]
  for spec in cases:
    if spec is None:
      os.environ.pop("AS_RCCL_FAIL_AT", None)
    else:
      os.environ["AS_RCCL_FAIL_AT"] = spec
    stages = _FakeStages()
    comm = rccl._create_staged(None, stages)
    dist.barrier()                          # both ranks are out of the protocol, in step
    rows.append((spec, comm is not None, list(stages.calls), rccl.last_error))
  os.environ.pop("AS_RCCL_FAIL_AT", None)
  torch.save(rows, os.path.join(out_dir, "staged_%d.pt" % rank))
  dist.destroy_process_group()


def _agree_tensor_worker(rank, world, port, out_dir):
  import datetime
  for p in (REPO, PKG):
    if p not in sys.path:
      sys.path.insert(0, p)
  os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
  dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
  from adaptive_stereo import rccl
  os.environ["AS_RCCL_FAIL_AT"] = "agree_tensor:0"
  stages = _FakeStages()
  try:
    rccl._create_staged(None, stages)
    verdict = "returned"
  except RuntimeError as e:
    verdict = "raised: %s" % e
  os.environ.pop("AS_RCCL_FAIL_AT", None)
  torch.save((verdict, list(stages.calls)), os.path.join(out_dir, "agree_%d.pt" % rank))
  dist.destroy_process_group()


def test_a_rank_without_its_agreement_buffer_raises_before_any_collective(tmp_path):
  """The documented residual case of the staged construction: a rank that cannot allocate the agreements' flag buffer on its
  device cannot take part in ANY collective — it must raise before the first one is entered (its peers then leave through the
  process group's timeout), not inside an agreement as round 4 did.  One rank: nothing may have run when it raises."""
  mp.spawn(_agree_tensor_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
  verdict, calls = torch.load(str(tmp_path / "agree_0.pt"))
  assert verdict.startswith("raised:") and "agree_tensor" in verdict, verdict
  assert calls == [], calls


def test_staged_communicator_construction_never_strands_a_rank(tmp_path):
  mp.spawn(_staged_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
  r0, r1 = (torch.load(str(tmp_path / ("staged_%d.pt" % r))) for r in range(2))
  assert len(r0) == len(r1) == 1 + 2 * len(RCCL_STAGES) - 1
  for (spec, ok0, calls0, err0), (spec1, ok1, calls1, err1) in zip(r0, r1):
    assert spec == spec1
    if spec is None:
      assert ok0 and ok1 and calls0 == ["load", "unique_id", "prepare", "init", "probe"] and calls1 == ["load", "prepare", "init", "probe"]
      continue
    stage, who = spec.split(":")
    assert not ok0 and not ok1, "a failure at %s must give None on EVERY rank" % spec
    failed_err, other_err = (err0, err1) if who == "0" else (err1, err0)
    assert failed_err.startswith(stage + ":") and "injected" in failed_err, (spec, failed_err)
    if spec == "unique_id:0":               # the None that travelled instead of an id is the other rank's own evidence
      assert other_err.startswith("receive:") and "no unique id arrived" in other_err, (spec, other_err)
    else:
      assert other_err == "another rank could not build its communicator", (spec, other_err)
    # nobody entered a stage behind the failed one's agreement: the collective init / probe only run when every rank got there
    later = RCCL_STAGES[RCCL_STAGES.index(stage) + 1:]
    gate = {"load": ("unique_id", "prepare", "init", "probe"), "unique_id": ("prepare", "init", "probe"),
            "receive": ("prepare", "init", "probe"), "prepare": ("init", "probe"), "init": ("probe",), "probe": ()}[stage]
    for calls in (calls0, calls1):
      assert not (set(calls) & set(gate)), (spec, calls, later)
    if stage in ("init", "probe"):          # the collective stages were entered by BOTH ranks (the error came out of the call)
      assert stage in calls0 and stage in calls1, (spec, calls0, calls1)
