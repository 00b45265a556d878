"""Deterministic synthetic stereo pairs and network weights.

There are no datasets or checkpoints in the build/bench environment, so every
test, golden fixture and benchmark draws its inputs from the recipes below.
Everything is generated on the CPU with a private ``torch.Generator`` so that
the result depends only on (seed, shape, torch version) and never on the global
RNG state or on the order in which modules were constructed (the reference
re-creates ``conv3d_alone`` four times inside its constructor loop,
``adaptive_stereo/models/stereo_net.py:155-162``, so init order is not a safe
thing to rely on).

This module is host-side plumbing only: it does not touch the GPU and it does
not depend on the HIP library.
"""
import zlib
from collections import OrderedDict

import torch
import torch.nn.functional as F


def _gen(seed: int) -> torch.Generator:
  g = torch.Generator(device="cpu")
  g.manual_seed(int(seed) & 0x7FFFFFFF)
  return g


def stereo_pair(batch: int, height: int, width: int, seed: int = 1,
                disparities=(5.0, 17.0, 40.0), noise: float = 0.02):
  """Returns (left, right) fp32 images in [0, 1], shape [B, 3, H, W].

  The left image is a band-limited random field (two octaves) plus 5% pixel noise. The right image
  is the left image shifted by a per-sample integer disparity d0 (so that
  R(x) = L(x + d0), i.e. a left pixel x appears at x - d0 in the right view)
  plus a little independent noise. A constant-zero image, as used by the
  reference's timing scripts (test/test_stereo_net.py:20), would make the cost
  volume identically zero and is useless for a parity check.
  """
  g = _gen(seed)

  def field(div):
    lo = torch.rand(batch, 3, max(height // div, 2), max(width // div, 2), generator=g)
    return F.interpolate(lo, size=(height, width), mode="bicubic", align_corners=False)

  # Band-limited like real imagery: structure at 1/16 and 1/4 resolution plus a little
  # per-pixel sensor noise.  (Per-pixel white noise at large amplitude would make the
  # warp's disparity gradient (I[x+1]-I[x]) chaotic under 1e-4-relative perturbations.)
  pixel = torch.rand(batch, 3, height, width, generator=g)
  left = (0.55 * field(16) + 0.40 * field(4) + 0.05 * pixel).clamp_(0.0, 1.0)

  right = torch.empty_like(left)
  for b in range(batch):
    d0 = int(disparities[b % len(disparities)])
    right[b] = torch.roll(left[b], shifts=-d0, dims=-1)
  right = (right + noise * (torch.rand(batch, 3, height, width, generator=g) - 0.5)).clamp_(0.0, 1.0)
  return left.contiguous(), right.contiguous()


def _is_bn_prefix(keys, prefix: str) -> bool:
  return (prefix + ".running_mean") in keys


def synthetic_state_dict(reference_state: "OrderedDict[str, torch.Tensor]", seed: int = 123,
                         logit_gain: float = 1.0) -> "OrderedDict[str, torch.Tensor]":
  """Builds a state_dict with the same keys/shapes/dtypes as ``reference_state``.

  Each tensor is drawn from its own generator seeded by crc32(key) ^ seed, so
  the values do not depend on key order. Convolutions get the PyTorch default
  scale U(-1/sqrt(fan_in), 1/sqrt(fan_in)); BatchNorm layers get non-trivial
  affine parameters and running statistics so that every term of the BN
  arithmetic is exercised.

  ``logit_gain`` multiplies ``conv3d_alone.{weight,bias}``: with these random
  weights the post-aggregation logits span about +-1 (FCS ~0.6); a gain of ~20
  gives a "trained-like" range (FCS ~12) comparable to the reference's plots
  (evaluation/cost_volume_analysis.py:147-150) and its OOD thresholds (~12).
  """
  keys = set(reference_state.keys())
  out = OrderedDict()
  for key, ref in reference_state.items():
    g = _gen(zlib.crc32(key.encode()) ^ seed)
    prefix, _, leaf = key.rpartition(".")
    if leaf == "num_batches_tracked":
      val = torch.zeros_like(ref)
    elif leaf == "running_mean":
      val = (torch.rand(ref.shape, generator=g) - 0.5) * 0.2
    elif leaf == "running_var":
      val = 0.5 + torch.rand(ref.shape, generator=g)
    elif _is_bn_prefix(keys, prefix):
      if leaf == "weight":
        val = 0.5 + torch.rand(ref.shape, generator=g)
      else:
        val = (torch.rand(ref.shape, generator=g) - 0.5) * 0.4
    else:
      # Convolution weight [Cout, Cin, *k] or bias [Cout]; the bound needs the
      # weight's fan-in, which for a bias is read from its sibling weight.
      w = reference_state[prefix + ".weight"]
      fan_in = int(w[0].numel())
      bound = 1.0 / (fan_in ** 0.5)
      val = (torch.rand(ref.shape, generator=g) * 2.0 - 1.0) * bound
      if prefix == "conv3d_alone":
        val = val * logit_gain
    out[key] = val.to(ref.dtype).contiguous()
  return out


def checksum(t: torch.Tensor):
  """(sum, sum of squares) in float64 — a cheap fingerprint for fixtures."""
  d = t.detach().double().reshape(-1)
  return float(d.sum()), float((d * d).sum())


def subsample(t: torch.Tensor, limit: int = 8192) -> torch.Tensor:
  """Deterministic strided subsample of a flattened tensor (<= limit values)."""
  flat = t.detach().reshape(-1)
  stride = max(1, -(-flat.numel() // limit))
  return flat[::stride].clone()
