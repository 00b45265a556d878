"""Per-launch time of the full-resolution layer's training forward: as_conv32_act_fwd (previous BatchNorm + LeakyReLU + skip
applied on the way in) against the two launches it replaces (as_bn_act_fwd of the previous layer + as_conv32_fwd).
  python tests/tools/microbench_act.py [pairs] [H] [W]"""
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
  z_prev, a_pp = tensor(), tensor()
  a_ref, z_ref, a_out, z = (ops.pcl_zeros(g, DEV) for _ in range(4))
  st = ops.BnState(DEV); st.scale.fill_(1.2); st.shift.fill_(0.1)
  b = torch.zeros(32, device=DEV)
  flops = 2.0 * B * H * W * 1024 * 9
  for dil in (1, 2, 4, 8, 1):                            # (the first line also carries the clock ramp)
    shape = ops.conv_shape_2d(dil)
    w = (torch.randn(32, 32, 3, 3, generator=gen) * 0.06).to(DEV)
    wp = ops.pack_weights(w, shape, False)
    stats_ref = ops.conv32_stat_parts(g, g, shape, DEV)
    stats = ops.StatParts(lib.as_conv32_act_parts(), DEV)
    stream = nat.stream()
    def act():
      ops.bn_act(z_prev, st, g, residual=a_pp, out=a_ref)
    def conv():
      ops.conv32(a_ref, g, wp, b, g, shape, out=z_ref, stats=stats_ref)
    def fused():
      nat.call("as_conv32_act_fwd", nat.ptr(z_prev), nat.ptr(a_pp), nat.ptr(st.scale), nat.ptr(st.shift), nat.ptr(a_out), g,
               nat.ptr(wp), nat.ptr(b), 0.2, nat.ptr(z), g, shape, nat.ptr(stats.mean), nat.ptr(stats.m2), nat.ptr(stats.cnt),
               stream)
    ta, tc, tf = timed(act), timed(conv), timed(fused)
    print("pairs %d dil %d: bn_act %.1f us + conv %.1f us (%.1f TFLOP/s) = %.1f us | fused %.1f us (%.1f TFLOP/s, %.0f %% of 157.3)"
          % (B, dil, ta, tc, flops / tc / 1e6, ta + tc, tf, flops / tf / 1e6, flops / tf / 1e6 / 157.3 * 100), flush=True)


if __name__ == "__main__":
  main()
