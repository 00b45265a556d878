"""Per-tensor gradient errors of the feature extractor's pair pass against the oracle, trunk kernels vs generic route.
usage (GPU box): python tests/tools/trunk_debug.py"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
for p in (REPO, os.path.join(REPO, "adaptive-stereo-icra-2021_amd")):
  sys.path.insert(0, p)
import torch
from adaptive_stereo import hip_ops
from adaptive_stereo.models.stereo_net import FeatureExtractorNetwork
from adaptive_stereo.utils import synthetic as syn
from oracle import stereo_oracle as orc

def rel(a, b):
  a, b = a.detach().cpu().double(), b.detach().cpu().double()
  return float((a - b).norm() / (b.norm() + 1e-30))

for (B, H, W, k) in [(4, 130, 700, 4), (4, 130, 700, 3), (2, 130, 700, 4)]:
  fnet0 = FeatureExtractorNetwork(k)
  fsd = syn.synthetic_state_dict(fnet0.state_dict(), seed=123)
  g = torch.Generator().manual_seed(16)
  left, right = torch.rand(B, 3, H, W, generator=g), torch.rand(B, 3, H, W, generator=g)
  fp = orc.make_params(fsd, True)
  fl_ref = orc.feature_extractor(fp, left, k, True)
  fr_ref = orc.feature_extractor(fp, right, k, True)
  gl, gr = torch.rand(fl_ref.shape, generator=g) - 0.5, torch.rand(fl_ref.shape, generator=g) - 0.5
  torch.autograd.backward([fl_ref, fr_ref], [gl, gr])
  out = {}
  for flag in (True, False):
    prev = hip_ops.set_trunk(flag)
    fnet = FeatureExtractorNetwork(k); fnet.load_state_dict(fsd); fnet = fnet.cuda().train()
    if flag:
      fl, fr = fnet.forward_pair(left.cuda(), right.cuda())
    else:
      fl, fr = fnet(left.cuda()), fnet(right.cuda())
    torch.autograd.backward([fl, fr], [gl.cuda(), gr.cuda()])
    out[flag] = {n: rel(p.grad, fp[n].grad) for n, p in fnet.named_parameters() if p.grad is not None and fp[n].grad is not None}
    out[flag]["__feat"] = float((fl.cpu() - fl_ref.detach()).abs().max())
    hip_ops.set_trunk(prev)
  print("case", (B, H, W, k), "feat err trunk %.2e generic %.2e" % (out[True]["__feat"], out[False]["__feat"]))
  for n in out[True]:
    if n != "__feat" and not n.endswith("conv1.0.0.bias"):
      print("   %-44s trunk %.2e   generic %.2e" % (n, out[True][n], out[False][n]))
