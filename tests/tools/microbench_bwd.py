"""Per-launch time of the full-resolution layer backward: one fused launch (as_conv32_bwd_fused) against the two launches it
replaces (as_conv32_wgrad_bnapply + as_conv32_fwd_bnbwd), HIP events around 20 back-to-back launches each.
  python tests/tools/microbench_bwd.py [pairs] [H] [W]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "adaptive-stereo-icra-2021_amd"))
from adaptive_stereo import _native as nat, hip_ops as ops
from adaptive_stereo.hip_ops import Pcl

DEV = torch.device("cuda:0")


def timed(fn, reps=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(reps): fn()
  b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b) / reps * 1e3


def main():
  B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
  H = int(sys.argv[2]) if len(sys.argv) > 2 else 375
  W = int(sys.argv[3]) if len(sys.argv) > 3 else 1242
  lib = nat.load()
  g = Pcl(B, 1, H, W, 0, 8, 8)
  gen = torch.Generator().manual_seed(0)
  def tensor():
    return ops.ncdhw_to_pcl(torch.randn(B, 32, 1, H, W, generator=gen).to(DEV), g)
  x, g_a, z, zn = tensor(), tensor(), tensor(), tensor()
  st, stn = ops.BnState(DEV), ops.BnState(DEV)
  for s in (st, stn):
    s.mean.normal_(0, 0.1); s.invstd.uniform_(0.5, 1.5); s.scale.copy_(s.invstd); s.shift.copy_(-s.mean * s.scale)
  coef = torch.rand(96, device=DEV) * 0.1
  flops = 2 * 2.0 * B * H * W * 1024 * 9
  for dil in (1, 2, 4, 8, 1):                            # (the first line also carries the clock ramp: 1 is measured again last)
    shape = ops.conv_shape_2d(dil)
    w = (torch.randn(32, 32, 3, 3, generator=gen) * 0.06).to(DEV)
    wp_t = ops.pack_weights(w, shape, True)
    gz, gx = ops.pcl_zeros(g, DEV), ops.pcl_zeros(g, DEV)
    dW, db = torch.zeros(32, 32, 3, 3, device=DEV), torch.zeros(32, device=DEV)
    wws = torch.empty(lib.as_conv32_wgrad_workspace(g, g, shape), device=DEV)
    nws = torch.empty(lib.as_bn_bwd_workspace(g), device=DEV)
    fws = torch.empty(lib.as_conv32_bwd_fused_workspace(), device=DEV)
    stream = nat.stream()
    def wgrad():
      nat.call("as_conv32_wgrad_bnapply", nat.ptr(x), g, nat.ptr(g_a), nat.ptr(z), g, shape, nat.ptr(st.scale), nat.ptr(st.shift),
               nat.ptr(st.mean), nat.ptr(coef), 0.2, nat.ptr(gz), nat.ptr(dW), nat.ptr(db), 0, nat.ptr(wws), stream)
    def dgrad():
      nat.call("as_conv32_fwd_bnbwd", nat.ptr(gz), g, nat.ptr(wp_t), nat.ptr(gx), g, shape, nat.ptr(g_a), nat.ptr(zn),
               nat.ptr(stn.scale), nat.ptr(stn.shift), nat.ptr(stn.mean), 0.2, nat.ptr(nws), stream)
    def fused():
      nat.call("as_conv32_bwd_fused", nat.ptr(x), g, nat.ptr(g_a), nat.ptr(z), g, shape, nat.ptr(wp_t), nat.ptr(st.scale),
               nat.ptr(st.shift), nat.ptr(st.mean), nat.ptr(coef), 0.2, nat.ptr(zn), nat.ptr(stn.scale), nat.ptr(stn.shift),
               nat.ptr(stn.mean), nat.ptr(gx), nat.ptr(dW), nat.ptr(db), 0, nat.ptr(nws), nat.ptr(fws), stream)
    tw, td, tf = timed(wgrad), timed(dgrad), timed(fused)
    if os.environ.get("AS_BW_TIMING"):                     # diagnostic build of the library (make EXTRA=-DBW_TIMING_BUILD)
      import numpy as np
      fused(); torch.cuda.synchronize()
      t = np.fromfile(os.path.join(ROOT, "gpurun_out", "bwd_timing.bin"), dtype=np.int64).reshape(-1, 4, 8)
      busy = t[t[:, 0, 7] > 0]
      names = ("decode", "run-in", "matrix", "wait B1", "vector", "wait B2", "drain")
      for role, sl in (("data-gradient waves", slice(0, 2)), ("weight-gradient waves", slice(2, 4))):
        m = busy[:, sl, :].mean(axis=(0, 1))
        tot = m[:7].sum()
        print("    %s: %s | cycles %.0f, 100-MHz ticks %.0f -> %.2f GHz" % (
            role, ", ".join("%s %.1f %%" % (n, 100 * v / tot) for n, v in zip(names, m[:7])), tot, m[7], tot / m[7] / 10))
    print("pairs %d dil %d: wgrad %.1f us + dgrad %.1f us = %.1f us (%.1f TFLOP/s) | fused %.1f us (%.1f TFLOP/s, %.0f %% of 157.3)"
          % (B, dil, tw, td, tw + td, flops / (tw + td) / 1e6, tf, flops / tf / 1e6, flops / tf / 1e6 / 157.3 * 100), flush=True)


if __name__ == "__main__":
  main()
