"""Per-launch time of one 3-D aggregation layer: first generation (conv3d_lds_kernel [+ bn_finalize + bn_act_fwd]) against
the rolling-window kernel (agg3d_kernel), per flavour and batch size.  Run on the GPU box."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import conftest   # noqa: F401,E402
from adaptive_stereo import _native as nat       # noqa: E402
from adaptive_stereo import hip_ops as ops       # noqa: E402
from adaptive_stereo.hip_ops import Pcl          # noqa: E402

DEV = "cuda:0"


def timeit(fn, reps=50):
  for _ in range(5):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record(); torch.cuda.synchronize()
  return 1e3 * e0.elapsed_time(e1) / reps      # us


def main():
  D, H, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (12, 24, 78)))
  shape = ops.CONV3D_333
  lib = nat.load()
  print("geometry D=%d H=%d W=%d; us per launch (TFLOP/s)" % (D, H, W))
  for B in (1, 2, 4, 8, 16):
    g = Pcl(B, D, H, W, 1, 1, 1)
    if lib.as_agg3d_ok(g) != 1:
      print("B=%d: agg3d not applicable" % B); continue
    flops = 2.0 * B * D * H * W * 1024 * 27
    x = ops.ncdhw_to_pcl(torch.randn(B, 32, D, H, W, device=DEV), g)
    w = torch.randn(32, 32, 3, 3, 3, device=DEV) / 29.4
    b = torch.randn(32, device=DEV) * 0.1
    gamma, beta = torch.rand(32, device=DEV) + 0.5, torch.randn(32, device=DEV) * 0.1
    rm, rv = torch.zeros(32, device=DEV), torch.ones(32, device=DEV)
    wp = ops.pack_weights(w, shape, False)
    z, a, z2 = ops.pcl_zeros(g, DEV), ops.pcl_zeros(g, DEV), ops.pcl_zeros(g, DEV)
    nparts = lib.as_agg3d_parts(g)
    stats_new = ops.StatParts(nparts, DEV)
    ops.set_agg3d(False)
    stats_old = ops.conv32_stat_parts(g, g, shape, DEV)
    t_old_conv = timeit(lambda: ops.conv32(x, g, wp, b, g, shape, out=z, stats=stats_old))
    st = ops.bn_train_stats(stats_old, gamma, beta, rm, rv)

    def old_layer():
      ops.conv32(x, g, wp, b, g, shape, out=z, stats=stats_old)
      s_ = ops.bn_train_stats(stats_old, gamma, beta, rm, rv)
      ops.bn_act(z, s_, g, out=a)
    t_old_layer = timeit(old_layer)
    t_old_plain = timeit(lambda: ops.conv32(x, g, wp, None, g, shape, out=z))
    ops.set_agg3d(True)
    t_new_plain = timeit(lambda: ops.agg3d(x, g, wp, None, z=z, epilogue=2))
    t_new_moments = timeit(lambda: ops.agg3d(x, g, wp, b, z=z, stats=stats_new))
    pend = ops.PendingBn(stats_new, gamma, beta, rm, rv)
    stats2 = ops.StatParts(nparts, DEV)
    t_new_fused = timeit(lambda: ops.agg3d(z, g, wp, b, z=z2, in_bn=pend, a_out=a, stats=stats2))
    t_new_fused_noout = timeit(lambda: ops.agg3d(z, g, wp, b, z=z2, in_bn=pend, stats=stats2))
    t_new_affine = timeit(lambda: ops.agg3d(z, g, wp, b, z=z2, in_state=st, a_out=a, stats=stats2))
    t_new_eval = timeit(lambda: ops.agg3d(x, g, wp, b, z=z, epilogue=1, ep_state=st))
    # the tail: bn_act + conv3d_alone + soft-argmax as three launches against the fused kernel
    w1 = torch.randn(1, 32, 3, 3, 3, device=DEV) * 0.03
    b1 = torch.zeros(1, device=DEV)
    logits = torch.empty(B, D, H, W, device=DEV); pred = torch.empty(B, H, W, device=DEV)
    am = torch.empty(B, H, W, dtype=torch.int32, device=DEV); fcs = torch.empty(B, H, W, device=DEV)

    def old_tail():
      ops.bn_act(z, st, g, out=a)
      nat.call("as_conv3d_out_fwd", nat.ptr(a), g, nat.ptr(w1), nat.ptr(b1), nat.ptr(logits), nat.stream())
      nat.call("as_softargmax_fwd", nat.ptr(logits), B, D, H, W, nat.ptr(pred), nat.ptr(am), nat.ptr(fcs), nat.stream())

    def new_tail(with_out):
      nat.call("as_agg_tail_fwd", nat.ptr(z), g, None, None, pend.block, nat.ptr(a) if with_out else None, nat.ptr(w1), nat.ptr(b1),
               0.2, nat.ptr(logits), nat.ptr(pred), nat.ptr(am), nat.ptr(fcs), nat.stream())
    t_old_tail, t_new_tail, t_new_tail_no = timeit(old_tail), timeit(lambda: new_tail(True)), timeit(lambda: new_tail(False))
    tf = lambda us: flops / us / 1e6
    print("B=%2d units=%4d | old: conv+moments %6.1f (%5.1f)  plain %6.1f (%5.1f)  conv+finalize+bn_act %6.1f | new: plain %6.1f (%5.1f)  "
          "moments %6.1f (%5.1f)  merge+act-in-LDS+by-product+moments %6.1f (%5.1f)  same without by-product %6.1f  affine given %6.1f  "
          "eval %6.1f (%5.1f) | tail: 3 launches %6.1f  fused %6.1f  fused without by-product %6.1f" % (
              B, nparts, t_old_conv, tf(t_old_conv), t_old_plain, tf(t_old_plain), t_old_layer, t_new_plain, tf(t_new_plain),
              t_new_moments, tf(t_new_moments), t_new_fused, tf(t_new_fused), t_new_fused_noout, t_new_affine, t_new_eval,
              tf(t_new_eval), t_old_tail, t_new_tail, t_new_tail_no), flush=True)


if __name__ == "__main__":
  main()
