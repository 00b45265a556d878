"""Phase timestamps of trunk_fwd_kernel (diagnostic build: make -C adaptive-stereo-icra-2021_amd/csrc EXTRA=-DTR_TIMING_BUILD -B trunk.o).
Runs one train-mode pair pass of the feature extractor at KITTI size with AS_TR_TIMING=1 and prints, per phase, the mean / max
over the workgroups of the LAST trunk forward launch in wall-clock ticks (100 MHz) converted to microseconds.
usage (GPU box): AS_TR_TIMING=1 python tests/tools/trunk_timing.py [pairs]"""
import os, struct, sys
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
for p in (REPO, os.path.join(REPO, "adaptive-stereo-icra-2021_amd")):
  sys.path.insert(0, p)
import torch
from adaptive_stereo.models.stereo_net import FeatureExtractorNetwork
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
fnet = FeatureExtractorNetwork(4).cuda().train()
l, r = torch.rand(B, 3, 375, 1242, device="cuda"), torch.rand(B, 3, 375, 1242, device="cuda")
for _ in range(3):
  fl, fr = fnet.forward_pair(l, r)
torch.cuda.synchronize()
raw = open(os.path.join(REPO, "gpurun_out", "trunk_timing.bin"), "rb").read()
n = struct.unpack("i", raw[:4])[0]
t = torch.frombuffer(bytearray(raw[4:4 + n * 64]), dtype=torch.int64).view(n, 8).double()
names = ["start->weights+loads issued", "BatchNorm merge", "activation + LDS staging (to barrier)", "matrix phase", "split-K reduce",
         "epilogue (stores + moments)", "loop end -> kernel end"]
tick_us = 1e6 / 1e8          # wall_clock64: 100 MHz
for i, nm in enumerate(names):
  d = (t[:, i + 1] - t[:, i]) * tick_us
  print("%-40s mean %6.2f us   max %6.2f us" % (nm, float(d.mean()), float(d.max())))
print("%-40s mean %6.2f us   max %6.2f us" % ("whole workgroup", float(((t[:, 7] - t[:, 0]) * tick_us).mean()), float(((t[:, 7] - t[:, 0]) * tick_us).max())))
print("first start -> last end: %.2f us over %d workgroups" % (float((t[:, 7].max() - t[:, 0].min()) * tick_us), n))
