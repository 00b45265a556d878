"""Same-box A/B of a route switch of hip_ops (set_winograd, set_defer_reduce, set_tail_bnsums, set_head_proj, set_fwd_act, set_bwd_fused,
set_agg3d, set_agg_tail, set_head_staged, set_refine_out): alternates on / off three times each and prints ms/step of the graph-replayed adaptation step.
Box-to-box variation is +-2 %; a 1 % effect only shows on one box, interleaved.  (The switches themselves are exercised by the
parity tests; timing them is this tool's business, not an environment variable's.)

usage (GPU box): python tests/tools/ab_switch.py set_fwd_act [pairs per step, default 4] [steps, default 40]"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
for p in (REPO, os.path.join(REPO, "adaptive-stereo-icra-2021_amd")):
  sys.path.insert(0, p)
import torch
from adaptive_stereo import hip_ops
from adaptive_stereo.adaptation import OnlineAdapter
from adaptive_stereo.models.stereo_net import StereoNet, FeatureExtractorNetwork
from adaptive_stereo.utils import synthetic as syn

setter = getattr(hip_ops, sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
left, right = (t.cuda() for t in syn.stereo_pair(B, 375, 1242, seed=1))
for rep in range(3):
  for on in (True, False):
    prev = setter(on)
    try:
      fnet, snet = FeatureExtractorNetwork(4), StereoNet(4, 1, 0, maxdisp=192)
      fnet.load_state_dict(syn.synthetic_state_dict(fnet.state_dict(), seed=123))
      snet.load_state_dict(syn.synthetic_state_dict(snet.state_dict(), seed=123))
      ad = OnlineAdapter(fnet.cuda(), snet.cuda(), 375, 1242, lr=5e-5)
      for _ in range(3):
        ad.step(left, right)
      ad.capture(left, right, warmup=1)
      l, r = ad.graph_inputs(); l.copy_(left); r.copy_(right)
      ad.step(l, r); torch.cuda.synchronize()
      t0 = time.perf_counter()
      for _ in range(steps):
        ad.step(l, r)
      torch.cuda.synchronize()
      print("%s(%s) batch %d: %.3f ms/step" % (sys.argv[1], on, B, 1e3 * (time.perf_counter() - t0) / steps), flush=True)
    finally:
      setter(prev)
    del ad, fnet, snet
