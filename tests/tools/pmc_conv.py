"""Launches the dominant kernel (conv32, 2-D 3x3 at the benchmark shape) a few times, for rocprofv3 --pmc runs."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "adaptive-stereo-icra-2021_amd"))
import torch
from adaptive_stereo import hip_ops as ops, _native as nat
from adaptive_stereo.hip_ops import Pcl
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
g = Pcl(B, 1, 375, 1242, 0, 8, 8)
shape = ops.conv_shape_2d(1)
x = torch.randn(g.numel(), device=dev) * 0.5
xv = ops.pcl_view(x, g).clone(); ops.pcl_interior(xv, g).zero_(); x = x - xv.view(-1)
w = torch.randn(32, 32, 3, 3, device=dev) * 0.05
b = torch.zeros(32, device=dev)
wp = ops.pack_weights(w, shape, False)
z = torch.zeros(g.numel(), device=dev)
stats = ops.conv32_stat_parts(g, g, shape, dev)
ws = torch.empty(nat.load().as_conv32_wgrad_workspace(g, g, shape), device=dev)
dW = torch.empty_like(w); db = torch.empty(32, device=dev)
wpt = ops.pack_weights(w, shape, True)
gx = torch.zeros(g.numel(), device=dev)
lib = nat.load()
st = ops.BnState(dev); st.mean.zero_(); st.invstd.fill_(1.0); st.scale.fill_(1.0); st.shift.zero_()
gam = torch.ones(32, device=dev)
bws = torch.empty(lib.as_bn_bwd_workspace(g), device=dev)
coef = bws[lib.as_bn_bwd_coef_offset():]; coef.zero_(); coef[64:96] = 1.0
gzo = torch.zeros(g.numel(), device=dev)
bws2 = torch.empty(lib.as_bn_bwd_workspace(g), device=dev)
fws = torch.empty(lib.as_conv32_bwd_fused_workspace(), device=dev)
zb = x.clone()            # a distinct buffer for the next layer's pre-activation (a shared one would be read from HBM once)
for _ in range(5):
  ops.conv32(x, g, wp, b, g, shape, out=z, stats=stats)              # conv32_lds_kernel<0,false>: training forward
  ops.conv32(z, g, wpt, None, g, shape, out=gx, residual=x)           # conv32_lds_kernel<2,true>: data gradient + skip
  # conv32_lds_kernel<3,true>: data gradient + skip + stage 1 of the next BatchNorm backward (what a step launches)
  nat.call("as_conv32_fwd_bnbwd", nat.ptr(z), g, nat.ptr(wpt), nat.ptr(gx), g, shape, nat.ptr(x), nat.ptr(zb), nat.ptr(st.scale),
           nat.ptr(st.shift), nat.ptr(st.mean), 0.2, nat.ptr(bws), nat.stream())
  # conv32_wgrad_lds2_kernel<true>: weight gradient + stage 3 of the layer's BatchNorm backward (B >= 2 at this size)
  if lib.as_conv32_wgrad_bnapply_ok(g, g, shape) == 1:
    nat.call("as_conv32_wgrad_bnapply", nat.ptr(x), g, nat.ptr(gx), nat.ptr(zb), g, shape, nat.ptr(st.scale), nat.ptr(st.shift),
             nat.ptr(st.mean), nat.ptr(coef), 0.2, nat.ptr(gzo), nat.ptr(dW), nat.ptr(db), 0, nat.ptr(ws), nat.stream())
  nat.call("as_conv32_wgrad", nat.ptr(x), g, nat.ptr(z), g, shape, nat.ptr(dW), nat.ptr(db), 0, nat.ptr(ws), nat.stream())
  # conv32_bwd_fused_kernel: the whole backward of a full-resolution layer in one launch (what a step launches when
  # as_conv32_bwd_fused_ok): x, g_a (= gzo here), z, z_next read, g_x written
  if lib.as_conv32_bwd_fused_ok(g, g, shape) == 1:
    nat.call("as_conv32_bwd_fused", nat.ptr(x), g, nat.ptr(gzo), nat.ptr(z), g, shape, nat.ptr(wpt), nat.ptr(st.scale),
             nat.ptr(st.shift), nat.ptr(st.mean), nat.ptr(coef), 0.2, nat.ptr(zb), nat.ptr(st.scale), nat.ptr(st.shift),
             nat.ptr(st.mean), nat.ptr(gx), nat.ptr(dW), nat.ptr(db), 0, nat.ptr(bws2), nat.ptr(fws), nat.stream())
# conv32_act_kernel<true>: the training forward with the previous BatchNorm + LeakyReLU + skip applied on the way in
if lib.as_conv32_act_ok(g, g, shape) == 1:
  stats_a = ops.StatParts(lib.as_conv32_act_parts(), dev)
  a_by = torch.zeros(g.numel(), device=dev)
  for _ in range(5):
    nat.call("as_conv32_act_fwd", nat.ptr(zb), nat.ptr(x), nat.ptr(st.scale), nat.ptr(st.shift), nat.ptr(a_by), g, nat.ptr(wp),
             nat.ptr(b), 0.2, nat.ptr(z), g, shape, nat.ptr(stats_a.mean), nat.ptr(stats_a.m2), nat.ptr(stats_a.cnt), nat.stream())
# the same layers by minimal filtering (what a step launches when as_conv32_wino_ok): forward with / without skip input,
# data gradient (conv32_wino_kernel<2, 0>) + weight gradient (conv32_wino_wgrad_kernel<0>)
if lib.as_conv32_wino_ok(g, g, shape) == 1:
  ww, ww_t = torch.empty(16 * 1024, device=dev), torch.empty(16 * 1024, device=dev)
  nat.call("as_conv32_wino_pack_weights", nat.ptr(w), nat.ptr(ww), 0, nat.stream())
  nat.call("as_conv32_wino_pack_weights", nat.ptr(w), nat.ptr(ww_t), 1, nat.stream())
  stats_w = ops.StatParts(lib.as_conv32_wino_parts(), dev)
  a_by = torch.zeros(g.numel(), device=dev)
  gzw = torch.zeros(g.numel(), device=dev)
  fws_w = torch.empty(lib.as_conv32_wino_bwd_workspace(), device=dev)
  fws_1 = torch.empty(lib.as_conv32_wino_bwd_fused_workspace(), device=dev)
  for _ in range(5):
    for skip in (x, None):
      nat.call("as_conv32_wino_fwd", nat.ptr(zb), nat.ptr(skip), nat.ptr(st.scale), nat.ptr(st.shift), nat.ptr(a_by), g, nat.ptr(ww),
               nat.ptr(b), 0.2, nat.ptr(z), g, shape, nat.ptr(stats_w.mean), nat.ptr(stats_w.m2), nat.ptr(stats_w.cnt), nat.stream())
    nat.call("as_conv32_wino_bwd", nat.ptr(x), g, nat.ptr(gzo), nat.ptr(z), g, shape, nat.ptr(ww_t), nat.ptr(st.scale),
             nat.ptr(st.shift), nat.ptr(st.mean), nat.ptr(coef), 0.2, nat.ptr(zb), nat.ptr(st.scale), nat.ptr(st.shift),
             nat.ptr(st.mean), nat.ptr(gzw), nat.ptr(gx), nat.ptr(dW), nat.ptr(db), 0, nat.ptr(bws2), nat.ptr(fws_w), nat.stream())
    # conv32_wino_bwd_kernel<0>: both gradients in ONE launch (what a step launches since round 5): x, g_a, z, z_next read, g_x written
    nat.call("as_conv32_wino_bwd_fused", nat.ptr(x), g, nat.ptr(gzo), nat.ptr(z), g, shape, nat.ptr(ww_t), nat.ptr(st.scale),
             nat.ptr(st.shift), nat.ptr(st.mean), nat.ptr(coef), 0.2, nat.ptr(zb), nat.ptr(st.scale), nat.ptr(st.shift),
             nat.ptr(st.mean), nat.ptr(gx), nat.ptr(dW), nat.ptr(db), 0, nat.ptr(bws2), nat.ptr(fws_1), nat.stream())
# a3: one 3-D cost-aggregation layer: rolling-window forward (plain, with moments, with the previous BatchNorm merged and
# applied in LDS + by-product), the fused tail (a4 + a5 + a8) and the LDS weight gradient
g3 = Pcl(B, 12, 24, 78, 1, 1, 1)
x3 = torch.randn(g3.numel(), device=dev) * 0.5
x3v = ops.pcl_view(x3, g3).clone(); ops.pcl_interior(x3v, g3).zero_(); x3 = x3 - x3v.view(-1)
w3 = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.03
wp3 = ops.pack_weights(w3, ops.CONV3D_333, False)
z3, z3b, a3 = torch.zeros(g3.numel(), device=dev), torch.zeros(g3.numel(), device=dev), torch.zeros(g3.numel(), device=dev)
nparts = lib.as_agg3d_parts(g3)
ws3 = torch.empty(nat.load().as_conv32_wgrad_workspace(g3, g3, ops.CONV3D_333), device=dev)
dW3 = torch.empty_like(w3)
w1 = torch.randn(1, 32, 3, 3, 3, device=dev) * 0.03
logits = torch.empty(B, 12, 24, 78, device=dev); pred = torch.empty(B, 24, 78, device=dev)
am = torch.empty(B, 24, 78, dtype=torch.int32, device=dev); fcs = torch.empty(B, 24, 78, device=dev)
for _ in range(5):
  ops.agg3d(x3, g3, wp3, None, z=z3, epilogue=2)                                   # agg3d_kernel<0, 2>: data gradient flavour
  pend = ops.PendingBn(ops.StatParts(nparts, dev), gam, torch.zeros(32, device=dev), torch.zeros(32, device=dev), torch.ones(32, device=dev))
  ops.agg3d(x3, g3, wp3, b, z=z3, stats=pend.stats)                                # agg3d_kernel<0, 0>: layer 1
  st2 = ops.StatParts(nparts, dev)
  ops.agg3d(z3, g3, wp3, b, z=z3b, in_bn=pend, a_out=a3, stats=st2)                # agg3d_kernel<2, 0>: layers 2-4
  pend2 = ops.PendingBn(st2, gam, torch.zeros(32, device=dev), torch.zeros(32, device=dev), torch.ones(32, device=dev))
  nat.call("as_agg_tail_fwd", nat.ptr(z3b), g3, None, None, pend2.block, nat.ptr(a3), nat.ptr(w1), None, 0.2,
           nat.ptr(logits), nat.ptr(pred), nat.ptr(am), nat.ptr(fcs), nat.stream())  # agg_tail_kernel<2, 8, true>
  nat.call("as_conv32_wgrad", nat.ptr(x3), g3, nat.ptr(z3), g3, ops.CONV3D_333, nat.ptr(dW3), nat.ptr(db), 0, nat.ptr(ws3), nat.stream())
torch.cuda.synchronize()
print("done")
