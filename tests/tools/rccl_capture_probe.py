"""Can a RCCL collective be captured INSIDE a hipGraph on this ROCm / PyTorch?  One-rank "nccl" process group on the box's
GPU: all-reduce and all-gather issued while a torch.cuda.graph capture is open, then replayed.  Prints one JSON line.
usage (GPU box): timeout -k 10 120 python tests/tools/rccl_capture_probe.py"""
import json, os, socket, sys, traceback
import torch
import torch.distributed as dist

sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
out = {"torch": torch.__version__, "hip": torch.version.hip}
side = torch.cuda.Stream()
for name in ("all_reduce", "all_gather", "all_reduce_between_kernels"):
  try:
    x = torch.arange(1024, dtype=torch.float32, device=dev)
    y = torch.zeros(1024, dtype=torch.float32, device=dev)
    gathered = [torch.zeros_like(x)]
    # warm-up outside capture (communicator creation must not happen inside a capture)
    with torch.cuda.stream(side):
      dist.all_reduce(x.clone())
      dist.all_gather([torch.zeros_like(x)], x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
      if name == "all_reduce":
        dist.all_reduce(x)
      elif name == "all_gather":
        dist.all_gather(gathered, x)
      else:
        x.mul_(2.0)
        dist.all_reduce(x)
        y.copy_(x).add_(1.0)
    torch.cuda.synchronize()
    x.copy_(torch.arange(1024, dtype=torch.float32, device=dev))
    for _ in range(3):
      g.replay()
    torch.cuda.synchronize()
    ref = torch.arange(1024, dtype=torch.float32, device=dev)
    if name == "all_reduce":
      ok = bool(torch.equal(x, ref))
    elif name == "all_gather":
      ok = bool(torch.equal(gathered[0], ref))
    else:
      ok = bool(torch.equal(x, ref * 8.0)) and bool(torch.equal(y, ref * 8.0 + 1.0))
    out[name] = {"captured": True, "replay_correct": ok}
  except Exception as e:
    out[name] = {"captured": False, "error": repr(e)[:400]}
    traceback.print_exc()
    torch.cuda.synchronize()
print(json.dumps(out))
dist.destroy_process_group()
