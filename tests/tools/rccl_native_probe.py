"""RCCL driven directly (ctypes on the librccl.so PyTorch ships): a one-rank communicator, ncclAllReduce / ncclAllGather issued
INSIDE a hipGraph capture on a torch stream, replayed.  c10d's ProcessGroupNCCL cannot do this on this build (its watchdog
polls an event of the capturing stream: hipErrorCapturedEvent, tests/tools/rccl_capture_probe.py).  Prints one JSON line.
usage (GPU box): timeout -k 10 120 python tests/tools/rccl_native_probe.py"""
import ctypes, glob, json, os, sys
import torch

libs = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*"))
rccl = ctypes.CDLL(libs[0])

class UniqueId(ctypes.Structure):
  _fields_ = [("internal", ctypes.c_char * 128)]

rccl.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
rccl.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
rccl.ncclAllGather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
rccl.ncclGetErrorString.restype = ctypes.c_char_p
NCCL_FLOAT, NCCL_SUM = 7, 0

def check(rc, what):
  if rc != 0:
    raise RuntimeError("%s: %s" % (what, rccl.ncclGetErrorString(rc).decode()))

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
uid = UniqueId()
check(rccl.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
comm = ctypes.c_void_p()
check(rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0), "ncclCommInitRank")
out = {"librccl": libs[0]}
side = torch.cuda.Stream()
x = torch.arange(4096, dtype=torch.float32, device=dev)
y = torch.zeros_like(x)
with torch.cuda.stream(side):
  check(rccl.ncclAllReduce(x.data_ptr(), x.data_ptr(), x.numel(), NCCL_FLOAT, NCCL_SUM, comm, side.cuda_stream), "warm-up all-reduce")
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
  x.mul_(2.0)
  check(rccl.ncclAllReduce(x.data_ptr(), x.data_ptr(), x.numel(), NCCL_FLOAT, NCCL_SUM, comm, torch.cuda.current_stream().cuda_stream), "captured all-reduce")
  check(rccl.ncclAllGather(x.data_ptr(), y.data_ptr(), x.numel(), NCCL_FLOAT, comm, torch.cuda.current_stream().cuda_stream), "captured all-gather")
  y.add_(1.0)
torch.cuda.synchronize()
ref = torch.arange(4096, dtype=torch.float32, device=dev)
x.copy_(ref)
for _ in range(3):
  g.replay()
torch.cuda.synchronize()
out["captured"] = True
out["replay_correct"] = bool(torch.equal(x, ref * 8.0)) and bool(torch.equal(y, ref * 8.0 + 1.0))
check(rccl.ncclCommDestroy(comm), "ncclCommDestroy")
print(json.dumps(out))
