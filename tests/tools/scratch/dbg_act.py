import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "adaptive-stereo-icra-2021_amd"))
from adaptive_stereo import _native as nat, hip_ops as ops
from adaptive_stereo.hip_ops import Pcl
DEV = torch.device("cuda:0")
B, H, W, dil = 2, 160, 1242, 1
g = Pcl(B, 1, H, W, 0, 8, 8); shape = ops.conv_shape_2d(dil); lib = nat.load()
gen = torch.Generator().manual_seed(0)
def T(): return ops.ncdhw_to_pcl(torch.randn(B, 32, 1, H, W, generator=gen).to(DEV), g)
z_prev, a_pp = T(), T()
w = (torch.randn(32, 32, 3, 3, generator=gen) * 0.06).to(DEV); b = torch.zeros(32, device=DEV)
wp = ops.pack_weights(w, shape, False)
st = ops.BnState(DEV); st.scale.fill_(1.5); st.shift.fill_(0.1)
a_ref = ops.bn_act(z_prev, st, g, residual=a_pp, out=ops.pcl_zeros(g, DEV))
z_ref = ops.conv32(a_ref, g, wp, b, g, shape, out=ops.pcl_zeros(g, DEV))
a_out, z = ops.pcl_zeros(g, DEV), ops.pcl_zeros(g, DEV)
stats = ops.StatParts(lib.as_conv32_act_parts(), DEV)
nat.call("as_conv32_act_fwd", nat.ptr(z_prev), nat.ptr(a_pp), nat.ptr(st.scale), nat.ptr(st.shift), nat.ptr(a_out), g,
         nat.ptr(wp), nat.ptr(b), 0.2, nat.ptr(z), g, shape, nat.ptr(stats.mean), nat.ptr(stats.m2), nat.ptr(stats.cnt), nat.stream())
torch.cuda.synchronize()
for name, got, ref in (("a", a_out, a_ref), ("z", z, z_ref)):
  gi, ri = ops.pcl_interior(ops.pcl_view(got, g), g), ops.pcl_interior(ops.pcl_view(ref, g), g)
  bad = (gi != ri)
  print(name, "bad", int(bad.sum()), "of", bad.numel(), "shape", tuple(bad.shape))
  # bad: [B, D, H, W, 32]?
  bb = bad.reshape(B, H, W, 32) if bad.dim() == 5 and bad.shape[1] == 1 else bad
  print(" per image", bb.flatten(1).sum(1).tolist())
  rows = bb.any(-1).sum(-1)[0]           # bad pixels per row, image 0
  print(" rows with bad px (img0):", [(int(i), int(v)) for i, v in enumerate(rows.tolist()) if v][:40])
  cols = bb.any(-1).sum(0 if bb.dim()==3 else 1)
  c0 = bb[0].any(-1).sum(0)
  print(" cols with bad px (img0):", [(int(i), int(v)) for i, v in enumerate(c0.tolist()) if v][:60])
  ch = bb[0].sum((0, 1))
  print(" channels:", ch.tolist())
