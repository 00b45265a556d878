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
# a3: one 3-D cost-aggregation layer (forward + weight gradient), 4 pairs
g3 = Pcl(B, 12, 24, 78, 1, 1, 1)
x3 = torch.randn(g3.numel(), device=dev) * 0.5
x3v = ops.pcl_view(x3, g3).clone(); ops.pcl_interior(x3v, g3).zero_(); x3 = x3 - x3v.view(-1)
w3 = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.03
wp3 = ops.pack_weights(w3, ops.CONV3D_333, False)
z3 = torch.zeros(g3.numel(), device=dev)
st3 = ops.conv32_stat_parts(g3, g3, ops.CONV3D_333, dev)
ws3 = torch.empty(nat.load().as_conv32_wgrad_workspace(g3, g3, ops.CONV3D_333), device=dev)
dW3 = torch.empty_like(w3)
for _ in range(5):
  ops.conv32(x3, g3, wp3, b, g3, ops.CONV3D_333, out=z3, stats=st3)
  nat.call("as_conv32_wgrad", nat.ptr(x3), g3, nat.ptr(z3), g3, ops.CONV3D_333, nat.ptr(dW3), nat.ptr(db), 0, nat.ptr(ws3), nat.stream())
torch.cuda.synchronize()
print("done")
