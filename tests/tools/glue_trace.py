"""Which Python lines launch the remaining torch (at::native / memcpy / fill) kernels of an eager adaptation step?
usage (GPU box): python tests/tools/glue_trace.py [pairs]"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
for p in (REPO, os.path.join(REPO, "adaptive-stereo-icra-2021_amd")):
  sys.path.insert(0, p)
import torch
from torch.profiler import profile, ProfilerActivity
from adaptive_stereo.adaptation import OnlineAdapter
from adaptive_stereo.models.stereo_net import StereoNet, FeatureExtractorNetwork
from adaptive_stereo.utils import synthetic as syn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
fnet, snet = FeatureExtractorNetwork(4), StereoNet(4, 1, 0, maxdisp=192)
fnet.load_state_dict(syn.synthetic_state_dict(fnet.state_dict(), seed=123)); snet.load_state_dict(syn.synthetic_state_dict(snet.state_dict(), seed=123))
ad = OnlineAdapter(fnet.cuda(), snet.cuda(), 375, 1242, lr=5e-5)
l, r = (t.cuda() for t in syn.stereo_pair(B, 375, 1242, seed=1))
pair = torch.cat([l, r]); l, r = pair[:B], pair[B:]
for _ in range(3):
  ad.step(l, r)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
  ad.step(l, r)
  torch.cuda.synchronize()
seen = {}
for ev in prof.events():
  if ev.device_type.name != "CPU" or not ev.name.startswith("aten::"):
    continue
  if not ev.kernels:
    continue
  here = [f for f in (ev.stack or []) if "adaptive_stereo" in f or "adaptation" in f]
  key = (ev.name, here[0] if here else (ev.stack[0] if ev.stack else "?"), str(ev.input_shapes)[:60])
  seen.setdefault(key, [0, 0.0])
  seen[key][0] += 1; seen[key][1] += sum(k.duration for k in ev.kernels)
for (name, where, shapes), (n, us) in sorted(seen.items(), key=lambda kv: -kv[1][1]):
  print("%-28s x%d %7.1f us  %s  %s" % (name, n, us, shapes, where[-110:]))
