import os, sys, time, socket
sys.path[:0] = ["/root/repo", "/root/repo/adaptive-stereo-icra-2021_amd"]
import torch, torch.distributed as dist
sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
torch.cuda.set_device(0)
t0 = time.time()
dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
print("pg init %.1fs" % (time.time() - t0), flush=True)
from adaptive_stereo import rccl
for i in range(2):
  t0 = time.time()
  c = rccl.try_create(None)
  print("try_create -> %s in %.1fs, last_error %s" % (c, time.time() - t0, rccl.last_error), flush=True)
dist.destroy_process_group()
