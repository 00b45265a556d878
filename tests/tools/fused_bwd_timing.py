"""Phase costs of the one-launch backward (diagnostic build: make -C adaptive-stereo-icra-2021_amd/csrc EXTRA=-DFB_TIMING_BUILD).
usage: AS_FB_TIMING=1 python tests/tools/fused_bwd_timing.py [pairs] [dilation]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "adaptive-stereo-icra-2021_amd"))
import numpy as np
import torch
from adaptive_stereo import _native as nat, hip_ops as ops
from adaptive_stereo._native import Pcl

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dil = int(sys.argv[2]) if len(sys.argv) > 2 else 1
DEV = torch.device("cuda", 0)
H, W = 375, 1242
g = Pcl(B, 1, H, W, 0, 8, 8)
lib = nat.load()
gen = torch.Generator().manual_seed(0)
T = lambda: ops.ncdhw_to_pcl(torch.randn(B, 32, 1, H, W, generator=gen).to(DEV), g)
x, g_a, z, zn = T(), T(), T(), T()
w = (torch.randn(32, 32, 3, 3, generator=gen) * 0.06).to(DEV)
st = ops.BnState(DEV); st.scale.fill_(1.0); st.shift.fill_(0.1); st.mean.fill_(0.05); st.invstd.fill_(1.0)
coef = torch.full((96,), 0.01, device=DEV); coef[64:] = 1.0
gx = ops.pcl_zeros(g, DEV)
dW, db = torch.zeros(32, 32, 3, 3, device=DEV), torch.zeros(32, device=DEV)
nws = torch.empty(lib.as_bn_bwd_workspace(g), device=DEV)
fws = torch.empty(lib.as_conv32_wino_bwd_fused_workspace(), device=DEV)
ww_t = torch.empty(16 * 1024, device=DEV)
nat.call("as_conv32_wino_pack_weights", nat.ptr(w), nat.ptr(ww_t), 1, nat.stream())
shape = ops.conv_shape_2d(dil)
def bwd():
  nat.call("as_conv32_wino_bwd_fused", nat.ptr(x), g, nat.ptr(g_a), nat.ptr(z), g, shape, nat.ptr(ww_t), nat.ptr(st.scale),
           nat.ptr(st.shift), nat.ptr(st.mean), nat.ptr(coef), 0.2, nat.ptr(zn), nat.ptr(st.scale), nat.ptr(st.shift),
           nat.ptr(st.mean), nat.ptr(gx), nat.ptr(dW), nat.ptr(db), 1, nat.ptr(nws), nat.ptr(fws), nat.stream())
names = ("decode + run-in", "matrix phase (data gradient)", "work up to B1", "wait at B1", "P2 work (convert | output rows | wgrad steps)",
         "wait at B2", "wait for loads", "drain")
for _ in range(3): bwd()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); bwd(); e1.record(); torch.cuda.synchronize()
print("launch + reduce + dump: %.1f us" % (e0.elapsed_time(e1) * 1e3))
path = os.path.join(REPO, "gpurun_out", "fused_bwd_timing.bin")
if not os.path.exists(path):
  print("no timing dump (production build?)"); sys.exit(0)
t = np.fromfile(path, dtype=np.int64).reshape(-1, 8, 10)
tiles = B * ((W + 63) // 64) * sum(((H - r + dil - 1) // dil + 1) // 2 for r in range(dil)) / t.shape[0]
for wv, nm in enumerate(("outer 0", "inner 1", "inner 2", "outer 3", "wgrad 0", "wgrad 1", "wgrad 2", "wgrad 3")):
  m = t[:, wv].mean(axis=0); tot = m[:8].sum()
  print("wave %d (%s), %d pairs, dilation %d: %.0f cycles per wave = %.0f per tile (%.1f tiles per workgroup), %.0f 100-MHz ticks -> %.2f GHz" % (
      wv, nm, B, dil, tot, tot / tiles, tiles, m[8], tot / max(m[8], 1) / 10))
  for n, v in zip(names, m[:8]):
    print("    %-50s %5.1f %%  %9.0f cycles  %7.0f per tile" % (n, 100 * v / tot, v, v / tiles))
hw = t[:, :, 9]
print("SIMD of waves 0..7 (HW_REG_HW_ID bits 5:4), first 12 workgroups:")
for b in range(12):
  print("   wg %3d: simd %s  wave slot %s  cu %s" % (b, [int(v >> 4) & 3 for v in hw[b]], [int(v) & 15 for v in hw[b]], [int(v >> 8) & 15 for v in hw[b]]))
import collections
pat = collections.Counter(tuple(int(v >> 4) & 3 for v in hw[b]) for b in range(hw.shape[0]))
print("placement patterns over all workgroups:", pat.most_common(8))
