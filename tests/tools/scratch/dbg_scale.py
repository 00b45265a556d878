import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "adaptive-stereo-icra-2021_amd"))
from adaptive_stereo import _native as nat, hip_ops as ops
from adaptive_stereo.hip_ops import Pcl
DEV = torch.device("cuda:0")
B, H, W = 2, 96, 256
g4 = g = Pcl(B, 1, H, W, 0, 8, 8); shape = ops.conv_shape_2d(1); lib = nat.load()
gen = torch.Generator().manual_seed(0)
def R(*s): return torch.randn(*s, generator=gen)
x4 = torch.zeros(lib.as_pcl4_numel(g4), device=DEV)
nat.call("as_pack_in4", nat.ptr(R(B,1,H,W).to(DEV)), nat.ptr(R(B,3,H,W).to(DEV)), 3, nat.ptr(x4), g4, nat.stream())
g_a0 = ops.ncdhw_to_pcl(R(B,32,1,H,W).to(DEV), g); z = ops.ncdhw_to_pcl(R(B,32,1,H,W).to(DEV), g)
g_pre0 = R(B,1,H,W).to(DEV).contiguous(); w0 = (R(32,4,3,3)*0.2).to(DEV)
st = ops.BnState(DEV); st.mean.copy_(R(32).to(DEV)*0.1); st.invstd.copy_(R(32).abs().to(DEV)+0.5)
gamma = (R(32).abs()+0.5).to(DEV); st.scale.copy_(st.invstd*gamma); st.shift.copy_(R(32).to(DEV)*0.1 - st.mean*st.scale)
ws = torch.empty(lib.as_conv4_wgrad_workspace(g, shape), device=DEV)
w_ch0 = w0[:,0].flip(-1,-2).reshape(32,9).contiguous(); w_proj = w_ch0.t().contiguous()
def run(scale, proj):
  g_a, g_pre = g_a0*scale, g_pre0*scale
  bws = torch.empty(lib.as_bn_bwd_workspace(g), device=DEV)
  gg, gb = torch.zeros(32, device=DEV), torch.zeros(32, device=DEV)
  nat.call("as_bn_act_bwd", nat.ptr(g_a), nat.ptr(z), nat.ptr(st.scale), nat.ptr(st.shift), nat.ptr(st.mean), nat.ptr(st.invstd),
           nat.ptr(gamma), 0.2, 1, None, nat.ptr(gg), nat.ptr(gb), 0, nat.ptr(bws), g, nat.stream())
  coef = bws[lib.as_bn_bwd_coef_offset():]
  dW = torch.zeros(32,4,3,3, device=DEV); db = torch.zeros(32, device=DEV); g_up = torch.empty(B,1,H,W, device=DEV)
  if proj:
    h = torch.empty(B*9*H*W, device=DEV)
    nat.call("as_conv4_wgrad_bnapply_proj", nat.ptr(x4), g4, nat.ptr(g_a), nat.ptr(z), g, shape, 4, nat.ptr(st.scale), nat.ptr(st.shift),
             nat.ptr(st.mean), nat.ptr(coef), 0.2, nat.ptr(w_proj), nat.ptr(h), nat.ptr(dW), nat.ptr(db), 0, nat.ptr(ws), nat.stream())
    nat.call("as_tap_gather", nat.ptr(h), nat.ptr(g_pre), nat.ptr(g_up), B, H, W, nat.stream())
  else:
    gz = ops.pcl_zeros(g, DEV)
    nat.call("as_conv4_wgrad_bnapply", nat.ptr(x4), g4, nat.ptr(g_a), nat.ptr(z), g, shape, 4, nat.ptr(st.scale), nat.ptr(st.shift),
             nat.ptr(st.mean), nat.ptr(coef), 0.2, nat.ptr(gz), nat.ptr(dW), nat.ptr(db), 0, nat.ptr(ws), nat.stream())
    nat.call("as_conv32to1_fwd", nat.ptr(gz), g, shape, nat.ptr(w_ch0), None, nat.ptr(g_pre), 0, nat.ptr(g_up), nat.stream())
  torch.cuda.synchronize()
  return g_up.double().cpu()/scale, dW.double().cpu()/scale, gg.double().cpu()/scale
s = 1.0/48611.0
for proj in (False, True):
  a, b = run(1.0, proj), run(s, proj)
  for name, u, v in zip(("g_up", "dW", "g_gamma"), a, b):
    print("proj", proj, name, "rel L2 diff between unit and 1/N-scaled flows: %.3e; max abs %.3e (scale %.3e)" % (float((u-v).norm()/u.norm()), float((u-v).abs().max()), float(u.abs().max())))
