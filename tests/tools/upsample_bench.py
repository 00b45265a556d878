"""Bilinear up-sampling 24x78 -> 375x1242 (a6) forward and adjoint, HIP-event times.  usage: python tests/tools/upsample_bench.py [B ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "adaptive-stereo-icra-2021_amd"))
import torch
from adaptive_stereo import _native as nat
DEV = "cuda:0"
for B in [int(a) for a in sys.argv[1:]] or [1, 4, 16, 32]:
  h, w, H, W = 24, 78, 375, 1242
  src = torch.randn(B, h, w, device=DEV); dst = torch.empty(B, H, W, device=DEV); gs = torch.empty(B, h, w, device=DEV)
  def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
  tf = t(lambda: nat.call("as_upsample_bilinear_fwd", nat.ptr(src), B, h, w, nat.ptr(dst), H, W, W / w, nat.stream()))
  tb = t(lambda: nat.call("as_upsample_bilinear_bwd", nat.ptr(dst), B, H, W, nat.ptr(gs), h, w, W / w, nat.stream()))
  by = 4.0 * B * (H * W + h * w)
  print("B=%2d  fwd %6.1f us (%5.0f GB/s)   bwd %6.1f us (%5.0f GB/s)" % (B, tf, by / tf / 1e3, tb, by / tb / 1e3), flush=True)
