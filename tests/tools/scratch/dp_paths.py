import os, sys, collections
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(REPO, "adaptive-stereo-icra-2021_amd"))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29656")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from adaptive_stereo import hip_ops, _native as nat
from adaptive_stereo.adaptation import OnlineAdapter
from adaptive_stereo.models.stereo_net import StereoNet, FeatureExtractorNetwork
from adaptive_stereo.utils import synthetic as syn
cnt = collections.Counter()
orig = hip_ops.call
def logged(name, *a):
  cnt[name] += 1
  return orig(name, *a)
hip_ops.call = logged
B, H, W, k = 2, 375, 1242, 4
left, right = (t.cuda() for t in syn.stereo_pair(B, H, W, seed=1))
for dp in (False, True):
  fnet, snet = FeatureExtractorNetwork(k), StereoNet(k, 1, 0, maxdisp=192)
  fnet.load_state_dict(syn.synthetic_state_dict(fnet.state_dict(), seed=123))
  snet.load_state_dict(syn.synthetic_state_dict(snet.state_dict(), seed=123, logit_gain=1.0))
  ad = OnlineAdapter(fnet.cuda(), snet.cuda(), H, W, lr=5e-5, force_data_parallel=dp)
  ad.step(left, right); cnt.clear()
  ad.step(left, right); torch.cuda.synchronize()
  print("dp" if dp else "plain", {k_: v for k_, v in cnt.items() if "conv32" in k_ and ("bwd" in k_ or "wgrad" in k_ or "wino" in k_ or "act" in k_ or "dgrad" in k_)}, flush=True)
  if dp: ad.close()
