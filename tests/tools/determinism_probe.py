"""Runs the same three adaptation steps twice from the same state and compares losses and parameters bit for bit, with the
minimal-filtering kernels off / forward only / on.  usage: python tests/tools/determinism_probe.py [pairs]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "adaptive-stereo-icra-2021_amd"))
import torch
from adaptive_stereo import hip_ops
from adaptive_stereo.adaptation import OnlineAdapter
from adaptive_stereo.models.stereo_net import StereoNet, FeatureExtractorNetwork
from adaptive_stereo.utils import synthetic as syn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
DP = len(sys.argv) > 2 and sys.argv[2] == "dp"
if DP:
  import torch.distributed as dist
  os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
  torch.cuda.set_device(0)
  dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
H, W, k = 375, 1242, 4
left, right = (t.cuda() for t in syn.stereo_pair(B, H, W, seed=1))

def run():
  fnet, snet = FeatureExtractorNetwork(k), StereoNet(k, 1, 0, maxdisp=192)
  fnet.load_state_dict(syn.synthetic_state_dict(fnet.state_dict(), seed=123))
  snet.load_state_dict(syn.synthetic_state_dict(snet.state_dict(), seed=123, logit_gain=1.0))
  ad = OnlineAdapter(fnet.cuda(), snet.cuda(), H, W, lr=5e-5, force_data_parallel=DP)
  losses = [float(ad.step(left, right)["loss"]) for _ in range(3)]
  torch.cuda.synchronize()
  params = ad.arena.params.detach().clone()
  if DP:
    ad.close()
  return losses, params

for name, fwd, bwd in (("direct", False, False), ("minimal filtering: forward only", True, False), ("minimal filtering: forward + backward", True, True)):
  hip_ops.set_winograd(fwd, backward=bwd)
  runs = [run() for _ in range(3)]
  same = all(r[0] == runs[0][0] and torch.equal(r[1], runs[0][1]) for r in runs[1:])
  print("%-42s %s   losses %s" % (name, "DETERMINISTIC" if same else "DIFFERS", [r[0][2] for r in runs]), flush=True)
