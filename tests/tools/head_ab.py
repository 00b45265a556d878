"""Strided head of the feature towers at the bench workload: as_conv32_fwd (5x5 stride 2), as_conv32_dgrad_s2 and as_conv32_wgrad with
as_conv32_s2_enable(0 / 1), HIP-event timing.  usage: python tests/tools/head_ab.py [images, default 8]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "adaptive-stereo-icra-2021_amd"))
import torch
from adaptive_stereo import _native as nat, hip_ops as ops
from adaptive_stereo._native import Pcl, ConvShape
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
DEV = torch.device("cuda", 0)
lib = nat.load()
shape = ConvShape(1, 5, 5, 0, 2, 2, 1, 2)
for (H, W) in ((188, 621), (94, 311)):
  Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
  gin, gout = Pcl(B, 1, H, W, 0, 2, 2), Pcl(B, 1, Ho, Wo, 0, 2, 2)
  g = torch.Generator().manual_seed(0)
  xb = ops.ncdhw_to_pcl(torch.randn(B, 32, 1, H, W, generator=g).to(DEV), gin)
  gzb = ops.ncdhw_to_pcl(torch.randn(B, 32, 1, Ho, Wo, generator=g).to(DEV), gout)
  w = (torch.randn(32, 32, 5, 5, generator=g) * 0.03).to(DEV); b = torch.zeros(32, device=DEV)
  wp = ops.pack_weights(w, shape, False)
  ws = torch.empty(lib.as_conv32_dgrad_s2_workspace(), device=DEV)
  nat.call("as_conv32_dgrad_s2_pack", nat.ptr(w), nat.ptr(ws), nat.stream())
  z, gx = ops.pcl_zeros(gout, DEV), ops.pcl_zeros(gin, DEV)
  flops = 2.0 * B * Ho * Wo * 1024 * 25
  for on in (0, 1):
    lib.as_conv32_s2_enable(on)
    for name, run in (("forward", lambda: ops.conv32(xb, gin, wp, b, gout, shape, out=z)),
                      ("data gradient", lambda: nat.call("as_conv32_dgrad_s2_packed", nat.ptr(gzb), gout, nat.ptr(ws), nat.ptr(gx), gin, nat.stream())),
                      ("weight grad. (+ reduce)", lambda: ops.conv32_wgrad(xb, gin, gzb, gout, shape))):
      for _ in range(3): run()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(20): run()
      e1.record(); torch.cuda.synchronize()
      us = e0.elapsed_time(e1) * 1e3 / 20
      print("%d x %dx%d -> %dx%d  %-23s %s %8.1f us  %6.1f TFLOP/s (%.3f of 157.3)" % (B, H, W, Ho, Wo, name, "staged " if on else "generic", us,
            flops / us * 1e-6, flops / us * 1e-6 / 157.3), flush=True)
