"""Times as_conv32_act_fwd against as_conv32_wino_fwd at the bench workload (B pairs x 375 x 1242), every dilation.
usage: python tests/tools/wino_microbench.py [B]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "adaptive-stereo-icra-2021_amd"))
import torch
from adaptive_stereo import _native as nat, hip_ops as ops
from adaptive_stereo._native import Pcl

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
DEV = torch.device("cuda", 0)
H, W = 375, 1242
g = Pcl(B, 1, H, W, 0, 8, 8)
lib = nat.load()
gen = torch.Generator().manual_seed(0)
z_prev = ops.ncdhw_to_pcl(torch.randn(B, 32, 1, H, W, generator=gen).to(DEV), g)
a_pp = ops.ncdhw_to_pcl(torch.randn(B, 32, 1, H, W, generator=gen).to(DEV), g)
w = (torch.randn(32, 32, 3, 3, generator=gen) * 0.06).to(DEV)
b = torch.zeros(32, device=DEV)
st = ops.BnState(DEV); st.scale.fill_(1.0); st.shift.fill_(0.1)
a_out, z = ops.pcl_zeros(g, DEV), ops.pcl_zeros(g, DEV)
ww = torch.empty(16 * 1024, device=DEV)
nat.call("as_conv32_wino_pack_weights", nat.ptr(w), nat.ptr(ww), 0, nat.stream())
flops = 2.0 * B * H * W * 1024 * 9
ONLY_BWD = os.environ.get("WMB_ONLY") == "bwd"
for dil in (() if ONLY_BWD else (1, 2, 4, 8)):
  shape = ops.conv_shape_2d(dil)
  wp = ops.pack_weights(w, shape, False)
  for skip in (True, False):
    for name, wt, parts, fg in (("act", wp, lib.as_conv32_act_parts(), None), ("wino", ww, lib.as_conv32_wino_parts(), None)):
      stats = ops.StatParts(parts, DEV)
      def run():
        nat.call("as_conv32_%s_fwd" % name, nat.ptr(z_prev), nat.ptr(a_pp) if skip else None, nat.ptr(st.scale), nat.ptr(st.shift),
                 nat.ptr(a_out), g, nat.ptr(wt), nat.ptr(b), 0.2, nat.ptr(z), g, shape, nat.ptr(stats.mean), nat.ptr(stats.m2),
                 nat.ptr(stats.cnt), nat.stream())
      for _ in range(3): run()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(20): run()
      e1.record(); torch.cuda.synchronize()
      us = e0.elapsed_time(e1) * 1e3 / 20
      print("B%d d%d %-5s %-4s%s %8.1f us  %6.1f TFLOP/s algorithmic (%.3f of 157.3)" % (B, dil, "skip" if skip else "plain", name, "" if fg is None else " g%d" % fg, us,
            flops / us * 1e-6, flops / us * 1e-6 / 157.3), flush=True)

# ---- inference block (as_conv32_wino_eval) ----
if not ONLY_BWD:
  e_out = ops.pcl_zeros(g, DEV)
  for dil in (1, 2, 4, 8):
    shape = ops.conv_shape_2d(dil)
    for fg in (1,):
      def run():
        nat.call("as_conv32_wino_eval", nat.ptr(z_prev), g, shape, nat.ptr(ww), nat.ptr(b), nat.ptr(st.scale), nat.ptr(st.shift), 0.2, 1,
                 nat.ptr(e_out), nat.stream())
      for _ in range(3): run()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(20): run()
      e1.record(); torch.cuda.synchronize()
      us = e0.elapsed_time(e1) * 1e3 / 20
      print("B%d d%d inference block %8.1f us" % (B, dil, us), flush=True)

# ---- backward: as_conv32_bwd_fused against as_conv32_wino_bwd (data gradient + weight gradient launches) ----
g_a, zn = a_pp, z_prev
x = ops.ncdhw_to_pcl(torch.randn(B, 32, 1, H, W, generator=gen).to(DEV), g)
zz = ops.ncdhw_to_pcl(torch.randn(B, 32, 1, H, W, generator=gen).to(DEV), g)
st.mean.fill_(0.05); st.invstd.fill_(1.0)
coef = torch.full((96,), 0.01, device=DEV); coef[64:] = 1.0
gz, gx = ops.pcl_zeros(g, DEV), ops.pcl_zeros(g, DEV)
dW, db = torch.zeros(32, 32, 3, 3, device=DEV), torch.zeros(32, device=DEV)
nws = torch.empty(lib.as_bn_bwd_workspace(g), device=DEV)
ww_t = torch.empty(16 * 1024, device=DEV)
nat.call("as_conv32_wino_pack_weights", nat.ptr(w), nat.ptr(ww_t), 1, nat.stream())
for dil in (1, 2, 4, 8):
  shape = ops.conv_shape_2d(dil)
  wp_t = ops.pack_weights(w, shape, True)
  fws = torch.empty(max(lib.as_conv32_bwd_fused_workspace(), lib.as_conv32_wino_bwd_workspace()), device=DEV)
  def run_fused():
    nat.call("as_conv32_bwd_fused", nat.ptr(x), g, nat.ptr(g_a), nat.ptr(zz), g, shape, nat.ptr(wp_t), nat.ptr(st.scale),
             nat.ptr(st.shift), nat.ptr(st.mean), nat.ptr(coef), 0.2, nat.ptr(zn), nat.ptr(st.scale), nat.ptr(st.shift),
             nat.ptr(st.mean), nat.ptr(gx), nat.ptr(dW), nat.ptr(db), 1, nat.ptr(nws), nat.ptr(fws), nat.stream())
  def run_wino():
    nat.call("as_conv32_wino_bwd", nat.ptr(x), g, nat.ptr(g_a), nat.ptr(zz), g, shape, nat.ptr(ww_t), nat.ptr(st.scale),
             nat.ptr(st.shift), nat.ptr(st.mean), nat.ptr(coef), 0.2, nat.ptr(zn), nat.ptr(st.scale), nat.ptr(st.shift),
             nat.ptr(st.mean), nat.ptr(gz), nat.ptr(gx), nat.ptr(dW), nat.ptr(db), 1, nat.ptr(nws), nat.ptr(fws), nat.stream())
  fws1 = torch.empty(lib.as_conv32_wino_bwd_fused_workspace(), device=DEV)
  def run_wino_one():
    nat.call("as_conv32_wino_bwd_fused", nat.ptr(x), g, nat.ptr(g_a), nat.ptr(zz), g, shape, nat.ptr(ww_t), nat.ptr(st.scale),
             nat.ptr(st.shift), nat.ptr(st.mean), nat.ptr(coef), 0.2, nat.ptr(zn), nat.ptr(st.scale), nat.ptr(st.shift),
             nat.ptr(st.mean), nat.ptr(gx), nat.ptr(dW), nat.ptr(db), 1, nat.ptr(nws), nat.ptr(fws1), nat.stream())
  gens = [int(v) for v in os.environ.get("WMB_GEN", "2").split(",") if v]      # data-gradient kernel generation(s): "1,2" times both
  legs = [("wino g%d" % gn, run_wino, gn) for gn in gens]
  legs = legs + [("wino 1L", run_wino_one, None)]           # both gradients in one launch (conv32_wino_bwd.hip)
  if not ONLY_BWD:
    legs = [("fused", run_fused, None)] + legs
  for name, run, gn in legs:
    if gn is not None:
      lib.as_conv32_wino_bwd_generation(gn)
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print("B%d d%d backward %-7s %8.1f us (incl. the slab reduce)  %6.1f TFLOP/s algorithmic (%.3f of 157.3)" % (
        B, dil, name, us, 2 * flops / us * 1e-6, 2 * flops / us * 1e-6 / 157.3), flush=True)
