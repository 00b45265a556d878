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
for _ in range(5):
  ops.conv32(x, g, wp, b, g, shape, out=z, stats=stats)              # conv32_lds_kernel<0,false>: training forward
  ops.conv32(z, g, wpt, None, g, shape, out=gx, residual=x)           # conv32_lds_kernel<2,true>: data gradient + skip
  nat.call("as_conv32_wgrad", nat.ptr(x), g, nat.ptr(z), g, shape, nat.ptr(dW), nat.ptr(db), 0, nat.ptr(ws), nat.stream())
torch.cuda.synchronize()
print("done")
