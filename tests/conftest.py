import json
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "adaptive-stereo-icra-2021_amd")
for p in (REPO, PKG):
  if p not in sys.path:
    sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")
SUB_LIMIT = 4096      # must match tests/golden/make_golden.py


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


# What the parity tests SAW (not only that they passed): arg-max mismatch counts, EPE, gradient errors ... one line per
# note at the end of the run (also with -q), so the numbers are in every test log.
PARITY_NOTES = []


def parity_note(test, **numbers):
  PARITY_NOTES.append((test, numbers))


def pytest_terminal_summary(terminalreporter, exitstatus, config):
  if not PARITY_NOTES:
    return
  terminalreporter.write_line("parity numbers seen by this run (%d notes):" % len(PARITY_NOTES))
  for test, numbers in PARITY_NOTES:
    terminalreporter.write_line("  parity %s %s" % (test, json.dumps(numbers, sort_keys=True)))


class Golden(object):
  """One tests/golden/<case>.npz: arrays produced by the reference itself."""

  def __init__(self, name):
    self.name = name
    self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    self.meta = json.loads(str(self.z["meta"]))
    self.no_grad_keys = json.loads(str(self.z["no_grad_keys"]))

  def has(self, key):
    return ("full__" + key) in self.z or ("sub__" + key) in self.z

  def scalar(self, key):
    return float(self.z[key])

  def full(self, key):
    return torch.from_numpy(self.z["full__" + key])

  def keys(self, prefix):
    out = []
    for k in self.z.files:
      if k.startswith("sum__" + prefix):
        out.append(k[len("sum__"):])
    return out

  def expected(self, key):
    """(values, is_full)"""
    if ("full__" + key) in self.z:
      return torch.from_numpy(self.z["full__" + key]), True
    return torch.from_numpy(self.z["sub__" + key]), False

  def compare(self, key, got, atol, rtol=0.0):
    """Returns max abs error; asserts shape and closeness against the fixture."""
    from adaptive_stereo.utils.synthetic import subsample
    got = got.detach().cpu()
    if got.dtype == torch.bool:
      got = got.to(torch.uint8)
    shape = tuple(int(v) for v in self.z["shape__" + key])
    assert tuple(got.shape) == shape, "%s: shape %s != golden %s" % (key, tuple(got.shape), shape)
    exp, is_full = self.expected(key)
    val = got if is_full else subsample(got, SUB_LIMIT)
    val = val.reshape(exp.shape).to(torch.float64)
    exp = exp.to(torch.float64)
    err = (val - exp).abs()
    tol = atol + rtol * exp.abs()
    worst = float(err.max()) if err.numel() else 0.0
    atol_shown = float(atol.max()) if torch.is_tensor(atol) else float(atol)
    assert bool((err <= tol).all()), "%s/%s: max abs err %.3e (atol %.1e rtol %.1e), %d/%d over" % (
        self.name, key, worst, atol_shown, rtol, int((err > tol).sum()), err.numel())
    return worst


GOLDEN_CASES = [
  "plumbing_240x320_k3_b1",
  "plumbing_240x320_k3_b2",
  "crop_96x256_k4_b1",
  "crop_96x256_k4_b2_trained",
  "odd_75x131_k3_b1",
  "kitti_375x1242_k4_b1",
  "kitti_375x1242_k4_b4",
  "sceneflow_540x960_k4_b1",
]


@pytest.fixture(scope="session")
def golden_loader():
  cache = {}

  def load(name):
    if name not in cache:
      cache[name] = Golden(name)
    return cache[name]
  return load
