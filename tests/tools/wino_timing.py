"""Phase costs of the minimal-filtering kernels (diagnostic build: make -C adaptive-stereo-icra-2021_amd/csrc EXTRA=-DWN_TIMING_BUILD).
usage: AS_WN_TIMING=1 python tests/tools/wino_timing.py [pairs] [dilation]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "adaptive-stereo-icra-2021_amd"))
import numpy as np
import torch
from adaptive_stereo import _native as nat, hip_ops as ops
from adaptive_stereo._native import Pcl

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dil = int(sys.argv[2]) if len(sys.argv) > 2 else 2
DEV = torch.device("cuda", 0)
H, W = 375, 1242
g = Pcl(B, 1, H, W, 0, 8, 8)
lib = nat.load()
gen = torch.Generator().manual_seed(0)
T = lambda: ops.ncdhw_to_pcl(torch.randn(B, 32, 1, H, W, generator=gen).to(DEV), g)
x, g_a, z, zn = T(), T(), T(), T()
w = (torch.randn(32, 32, 3, 3, generator=gen) * 0.06).to(DEV)
b = torch.zeros(32, device=DEV)
st = ops.BnState(DEV); st.scale.fill_(1.0); st.shift.fill_(0.1); st.mean.fill_(0.05); st.invstd.fill_(1.0)
coef = torch.full((96,), 0.01, device=DEV); coef[64:] = 1.0
out1, out2 = ops.pcl_zeros(g, DEV), ops.pcl_zeros(g, DEV)
dW, db = torch.zeros(32, 32, 3, 3, device=DEV), torch.zeros(32, device=DEV)
nws = torch.empty(lib.as_bn_bwd_workspace(g), device=DEV)
fws = torch.empty(lib.as_conv32_wino_bwd_workspace(), device=DEV)
ww, ww_t = torch.empty(16 * 1024, device=DEV), torch.empty(16 * 1024, device=DEV)
nat.call("as_conv32_wino_pack_weights", nat.ptr(w), nat.ptr(ww), 0, nat.stream())
nat.call("as_conv32_wino_pack_weights", nat.ptr(w), nat.ptr(ww_t), 1, nat.stream())
shape = ops.conv_shape_2d(dil)
stats = ops.StatParts(lib.as_conv32_wino_parts(), DEV)
def fwd(skip):
  nat.call("as_conv32_wino_fwd", nat.ptr(z), nat.ptr(g_a) if skip else None, nat.ptr(st.scale), nat.ptr(st.shift), nat.ptr(out1), g,
           nat.ptr(ww), nat.ptr(b), 0.2, nat.ptr(out2), g, shape, nat.ptr(stats.mean), nat.ptr(stats.m2), nat.ptr(stats.cnt), nat.stream())
def bwd():
  nat.call("as_conv32_wino_bwd", nat.ptr(x), g, nat.ptr(g_a), nat.ptr(z), g, shape, nat.ptr(ww_t), nat.ptr(st.scale),
           nat.ptr(st.shift), nat.ptr(st.mean), nat.ptr(coef), 0.2, nat.ptr(zn), nat.ptr(st.scale), nat.ptr(st.shift),
           nat.ptr(st.mean), nat.ptr(out1), nat.ptr(out2), nat.ptr(dW), nat.ptr(db), 1, nat.ptr(nws), nat.ptr(fws), nat.stream())
names = ("decode", "run-in", "matrix", "T + exchange write + late loads", "wait B1", "exchange read + Y", "stores + moments",
         "wait rows", "convert", "wait B2", "drain")
for mode, run in ((0, lambda: fwd(False)), (1, lambda: fwd(True)), (2, bwd)):
  for _ in range(3): run()
  torch.cuda.synchronize()
  path = os.path.join(REPO, "gpurun_out", "wino_timing_m%d.bin" % mode)
  if not os.path.exists(path):
    print("no timing dump (production build?)"); break
  t = np.fromfile(path, dtype=np.int64).reshape(-1, 4, 12)
  m = t.mean(axis=(0, 1)); tot = m[:11].sum()
  print("mode %d, %d pairs, dilation %d: %.0f cycles per wave, %.0f 100-MHz ticks -> %.2f GHz" % (mode, B, dil, tot, m[11], tot / m[11] / 10))
  for n, v in zip(names, m[:11]):
    print("    %-34s %5.1f %%  %9.0f cycles" % (n, 100 * v / tot, v))
