"""Backward of the aggregation tail (a5 + a4) at the bench geometry (12 x 24 x 78 per pair): one launch
(csrc/agg_tail_bwd.hip) against as_softargmax_bwd + as_conv3d_out_bwd.  usage: python tests/tools/tail_bwd_bench.py [B ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "adaptive-stereo-icra-2021_amd"))
import torch
from adaptive_stereo import _native as nat, hip_ops as ops
from adaptive_stereo._native import Pcl

DEV = "cuda:0"
lib = nat.load()


def timed(fn, n=50, warm=5):
  for _ in range(warm):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n):
    fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / n


for B in [int(x) for x in sys.argv[1:]] or [1, 4, 16, 32]:
  D, H, W = 12, 24, 78
  g = Pcl(B, D, H, W, 1, 1, 1)
  gen = torch.Generator().manual_seed(1)
  logits = (torch.randn(B, D, H, W, generator=gen) * 6).to(DEV)
  gp = torch.randn(B, H, W, generator=gen).to(DEV)
  a = ops.ncdhw_to_pcl(torch.randn(B, 32, D, H, W, generator=gen).to(DEV), g)
  w = (torch.randn(1, 32, 3, 3, 3, generator=gen) * 0.05).to(DEV)
  ga = ops.pcl_zeros(g, DEV); gw = torch.empty_like(w); gb = torch.empty(1, device=DEV)
  gl = torch.empty(B, D, H, W, device=DEV)
  ws = torch.empty(lib.as_conv3d_out_bwd_workspace(g), device=DEV)
  ws2 = torch.empty(lib.as_agg_tail_bwd_workspace(g), device=DEV)

  def old():
    nat.call("as_softargmax_bwd", nat.ptr(logits), nat.ptr(gp), None, B, D, H, W, nat.ptr(gl), nat.stream())
    nat.call("as_conv3d_out_bwd", nat.ptr(gl), nat.ptr(a), g, nat.ptr(w), nat.ptr(ga), nat.ptr(gw), nat.ptr(gb), 0, nat.ptr(ws), nat.stream())

  def new():
    nat.call("as_agg_tail_bwd", nat.ptr(logits), nat.ptr(gp), None, nat.ptr(a), g, nat.ptr(w), nat.ptr(ga), nat.ptr(gw), nat.ptr(gb), 0,
             nat.ptr(ws2), nat.stream())

  to, tn = timed(old), timed(new)
  V = B * 32 * D * H * W * 4
  print("B=%2d  one launch (+ slab reduction) %6.1f us  (%5.0f GB/s of V read + V written)   three launches %6.1f us" % (B, tn, 2 * V / tn / 1e3, to), flush=True)
