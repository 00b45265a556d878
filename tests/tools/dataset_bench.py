"""Dataset layer: samples/s of the device-side decode (raw uint8 / uint16 samples -> cropped, flipped float tensors +
pyramid) next to the host path (the reference's torch ops on the CPU) at KITTI size.  Diagnostic, not a test.
usage (GPU box): python tests/tools/dataset_bench.py"""
import os, sys, tempfile, time, random
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "adaptive-stereo-icra-2021_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
from adaptive_stereo.datasets.stereo_dataset import StereoDataset
from dataset_fixture import make_tree

root = tempfile.mkdtemp()
data, splits = make_tree(root, "KittiStereo2015", n=8, H0=375, W0=1242)
kw = dict(scales=[0, 1, 2, 3], do_hflip=True, random_crop=True, splits_path=splits)
host = StereoDataset(data, "KittiStereo2015", "tiny", 320, 960, "train", **kw)
dev = StereoDataset(data, "KittiStereo2015", "tiny", 320, 960, "train", device="cuda:0", **kw)

def run(ds, n):
  t0 = time.perf_counter()
  for k in range(n):
    s = ds[k % len(ds)]
  if ds.device is not None:
    torch.cuda.synchronize()
  return (time.perf_counter() - t0) / n

random.seed(0); run(dev, 8); run(host, 4)
random.seed(1); t_dev = run(dev, 64)
random.seed(1); t_host = run(host, 32)
# device-side part alone: decode kernels on already uploaded samples (what a pipelined loader overlaps with file parsing)
from adaptive_stereo import _native as nat
import numpy as np
u8 = torch.randint(0, 256, (375, 1242, 3), dtype=torch.uint8, device="cuda:0")
u16 = torch.randint(0, 30000, (375, 1242), dtype=torch.int16, device="cuda:0")
rgb = torch.empty(3, 320, 960, device="cuda:0"); dsp = torch.empty(1, 320, 960, device="cuda:0")
lvl = [torch.empty(4, 320 >> s, 960 >> s, device="cuda:0") for s in (1, 2, 3)]
def kernels():
  for _ in range(2):
    nat.call("as_decode_rgb8", nat.ptr(u8), 375, 1242, 20, 100, 320, 960, 1, nat.ptr(rgb), nat.stream())
    nat.call("as_decode_plane", nat.ptr(u16), 1, 375, 1242, 20, 100, 320, 960, 1, 0, 1.0 / 256, 0, nat.ptr(dsp), nat.stream())
    for s, d in zip((1, 2, 3), lvl):
      nat.call("as_upsample_bilinear_fwd", nat.ptr(rgb), 3, 320, 960, nat.ptr(d), 320 >> s, 960 >> s, 1.0, nat.stream())
      nat.call("as_upsample_bilinear_fwd", nat.ptr(dsp), 1, 320, 960, nat.ptr(d), 320 >> s, 960 >> s, 1.0 / 2 ** s, nat.stream())
for _ in range(3): kernels()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): kernels()
e1.record(); torch.cuda.synchronize()
t_k = e0.elapsed_time(e1) / 20 * 1e-3
from adaptive_stereo.datasets.prefetch import DevicePrefetcher
def run_prefetch(threads, epochs=4):
  random.seed(2)
  n = 0
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(epochs):
    for batch in DevicePrefetcher(dev, batch_size=4, num_threads=threads):
      n += batch["color_l/0"].shape[0]
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / n
run_prefetch(8, 1)
t_pf = {th: run_prefetch(th) for th in (1, 4, 8, 16)}
print("KITTI 375x1242 -> 320x960 crop, flip, scales 0-3, both views + both disparities, PNG files on tmpfs:")
print("  host path (reference's torch ops, 1 process): %6.1f ms/sample = %6.1f samples/s" % (1e3 * t_host, 1 / t_host))
print("  device path (PIL parse + upload + HIP kernels): %6.1f ms/sample = %6.1f samples/s" % (1e3 * t_dev, 1 / t_dev))
print("  device kernels alone (16 launches per sample): %6.3f ms/sample = %6.0f samples/s" % (1e3 * t_k, 1 / t_k))
print("  DevicePrefetcher (thread-pool parsing, side-stream decode, batches of 4): " + ", ".join(
    "%d threads %.0f samples/s" % (th, 1 / t) for th, t in t_pf.items()))
