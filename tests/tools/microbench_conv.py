"""Times conv32 forward / wgrad and the BN passes at the benchmark's shapes (diagnostic, not a test)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "adaptive-stereo-icra-2021_amd"))
import torch
from adaptive_stereo import hip_ops as ops, _native as nat
from adaptive_stereo.hip_ops import Pcl, ConvShape

dev = "cuda:0"

def timeit(fn, reps=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps * 1e3   # us

cases = [
  ("2D 375x1242 B4 dil1", Pcl(4, 1, 375, 1242, 0, 8, 8), ops.conv_shape_2d(1)),
  ("2D 375x1242 B4 dil8", Pcl(4, 1, 375, 1242, 0, 8, 8), ops.conv_shape_2d(8)),
  ("2D 375x1242 B1 dil1", Pcl(1, 1, 375, 1242, 0, 8, 8), ops.conv_shape_2d(1)),
  ("3D 12x24x78 B4", Pcl(4, 12, 24, 78, 1, 1, 1), ops.CONV3D_333),
  ("3D 12x24x78 B1", Pcl(1, 12, 24, 78, 1, 1, 1), ops.CONV3D_333),
  ("3D 12x24x78 B16", Pcl(16, 12, 24, 78, 1, 1, 1), ops.CONV3D_333),
  ("3D 12x24x78 B64", Pcl(64, 12, 24, 78, 1, 1, 1), ops.CONV3D_333),
]
only = sys.argv[1:]
for name, g, shape in cases:
  if only and not any(o in name for o in only): continue
  taps = shape.taps()
  x = torch.randn(g.numel(), device=dev) * 0.5
  xv = ops.pcl_view(x, g).clone(); ops.pcl_interior(xv, g).zero_(); x = x - xv.view(-1)   # zero halo
  gz = x.clone()
  w = torch.randn(32, 32, *((3, 3, 3) if shape.kd > 1 else (3, 3)), device=dev) * 0.05
  b = torch.zeros(32, device=dev)
  wp = ops.pack_weights(w, shape, False)
  z = torch.zeros(g.numel(), device=dev)
  stats = ops.conv32_stat_parts(g, g, shape, dev)
  flops = 2.0 * g.voxels() * 1024 * taps
  t_f = timeit(lambda: ops.conv32(x, g, wp, b, g, shape, out=z, stats=stats))
  one = b + 1
  t_fe = timeit(lambda: ops.conv32(x, g, wp, b, g, shape, out=z, epilogue=1, scale=one, shift=b))
  t_res = timeit(lambda: ops.conv32(x, g, wp, None, g, shape, out=z, residual=gz))
  t_fres = timeit(lambda: ops.conv32(x, g, wp, b, g, shape, out=z, epilogue=1, scale=one, shift=b, residual=gz))
  t_fus = float("nan")
  if nat.load().as_conv32_bnbwd_parts(g, g, shape) > 0:
    stb = ops.BnState(dev); stb.mean.zero_(); stb.invstd.fill_(1.0); stb.scale.fill_(1.0); stb.shift.zero_()
    zb = x.clone()
    def fused():
      r = ops.conv32_dgrad_bnbwd(x, g, wp, shape, gz, zb, stb); ops.POOL.put(r[0], g)
    t_fus = timeit(fused)
  ws = torch.empty(nat.load().as_conv32_wgrad_workspace(g, g, shape), device=dev)
  dW = torch.empty_like(w); db = torch.empty(32, device=dev)
  t_w = timeit(lambda: nat.call("as_conv32_wgrad", nat.ptr(x), g, nat.ptr(gz), g, shape, nat.ptr(dW), nat.ptr(db), 0, nat.ptr(ws), nat.stream()))
  st = ops.BnState(dev); st.mean.zero_(); st.invstd.fill_(1.0); st.scale.fill_(1.0); st.shift.zero_()
  a = torch.zeros(g.numel(), device=dev)
  t_a = timeit(lambda: ops.bn_act(z, st, g, residual=x, out=a))
  gam = torch.ones(32, device=dev)
  def bwd():
    gzz, _, _ = ops.bn_act_bwd(a, z, st, gam, g, True); ops.POOL.put(gzz, g)
  t_b = timeit(bwd)
  t_fin = timeit(lambda: ops.bn_train_stats(stats, gam, b, None, None))
  byts = g.voxels() * 128
  print("%-22s dgrad+res %8.1f us %6.1f TF | fused-ep+res %8.1f us %6.1f TF | dgrad+res+bn-sums %8.1f us %6.1f TF" % (
      name, t_res, flops / t_res / 1e6, t_fres, flops / t_fres / 1e6, t_fus, flops / t_fus / 1e6), flush=True)
  print("%-22s fwd %8.1f us %6.1f TF | fused-ep %8.1f us %6.1f TF | wgrad %8.1f us %6.1f TF | bn_act %7.1f us %5.2f TB/s | bn_bwd %7.1f us %5.2f TB/s | finalize %6.1f us" % (
      name, t_f, flops / t_f / 1e6, t_fe, flops / t_fe / 1e6, t_w, flops / t_w / 1e6, t_a, 3 * byts / t_a / 1e6, t_b, 5 * byts / t_b / 1e6, t_fin), flush=True)
