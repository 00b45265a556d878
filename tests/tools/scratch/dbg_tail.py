import os, sys
sys.path.insert(0, "/root/repo/adaptive-stereo-icra-2021_amd"); sys.path.insert(0, "/root/repo/tests")
import torch
from adaptive_stereo import _native as nat, hip_ops as ops
from adaptive_stereo._native import Pcl
DEV="cuda:0"; lib=nat.load()
def rnd(*s, seed=0, scale=1.0):
  return torch.randn(*s, generator=torch.Generator().manual_seed(seed)) * scale
for (B,D,H,W) in [(1,8,5,9),(1,3,1,1),(4,12,24,78)]:
  g = Pcl(B, D, H, W, 1, 1, 1)
  logits = (rnd(B, D, H, W, seed=1) * 6.0).to(DEV); gp = rnd(B, H, W, seed=2).to(DEV)
  a = ops.ncdhw_to_pcl(rnd(B, 32, D, H, W, seed=4).to(DEV), g)
  w = (rnd(1, 32, 3, 3, 3, seed=5, scale=0.05)).to(DEV).contiguous()
  gl = torch.empty(B, D, H, W, device=DEV)
  nat.call("as_softargmax_bwd", nat.ptr(logits), nat.ptr(gp), None, B, D, H, W, nat.ptr(gl), nat.stream())
  ga_ref = ops.pcl_zeros(g, DEV); gw_ref = torch.empty_like(w); gb_ref = torch.empty(1, device=DEV)
  ws = torch.empty(lib.as_conv3d_out_bwd_workspace(g), device=DEV)
  nat.call("as_conv3d_out_bwd", nat.ptr(gl), nat.ptr(a), g, nat.ptr(w), nat.ptr(ga_ref), nat.ptr(gw_ref), nat.ptr(gb_ref), 0, nat.ptr(ws), nat.stream())
  ga = ops.pcl_zeros(g, DEV); gw = torch.zeros_like(w); gb = torch.zeros(1, device=DEV)
  ws2 = torch.zeros(lib.as_agg_tail_bwd_workspace(g), device=DEV)
  nat.call("as_agg_tail_bwd", nat.ptr(logits), nat.ptr(gp), None, nat.ptr(a), g, nat.ptr(w), nat.ptr(ga), nat.ptr(gw), nat.ptr(gb), 0, nat.ptr(ws2), nat.stream())
  torch.cuda.synchronize()
  r = (gw / gw_ref).flatten()
  print((B,D,H,W), "ga equal", torch.equal(ga, ga_ref), "gb", float(gb), float(gb_ref), "gw ratio quantiles", [round(float(x),4) for x in torch.quantile(r.cpu(), torch.tensor([0.,.25,.5,.75,1.]))])
  print("  gw[0,:2,0,0,:3]", gw[0,:2,0,0,:3].cpu().tolist(), "ref", gw_ref[0,:2,0,0,:3].cpu().tolist())
