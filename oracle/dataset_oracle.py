"""CPU restatement of the reference's dataset layer — TEST INFRASTRUCTURE ONLY: nothing outside tests/ may import it.

Follows adaptive_stereo/datasets/stereo_dataset.py:49-143 and utils/dataset_utils.py:19-57 step by step with numpy
and torch-CPU ops, independently of the product code under adaptive-stereo-icra-2021_amd/: full-size float images,
flip, crop, F.interpolate pyramid.  Parity is pinned by construction here (the reference module itself cannot be
imported in this container: torchvision, imageio and cv2 are absent): every step is one documented library call —
ToTensor = uint8/255, imageio/cv2 PNG decode = the stored integers, PFM = big/little-endian float32 rows bottom-up.

PARITY: PINNED for exactly two parts, UNPINNED for the rest.
  * pinned — ``pfm`` / the SceneFlow branch of ``load_disp``: tests/test_dataset_cpu.py runs them on the reference's own
    sample (resources/0008.pfm, copied as tests/golden/dataset/0008.pfm) against the output of the reference's readPFM
    (utils/io.py:37-80, imported from /root/reference by tests/golden/make_dataset_golden.py), bit for bit;
  * pinned — manifest handling (one sample per line, four paths per line): the data set lengths the reference's own test
    asserts (test/test_stereo_dataset.py:24-97) re-counted from the reference's manifests, tests/golden/dataset/;
  * unpinned — crop / flip / pyramid / PNG and NPY decoders: the reference's dataset module cannot be imported here
    (torchvision, imageio, cv2 are absent) and the reference holds no fixtures for them, so those functions rest on
    this file's reading of the reference source alone."""
import re

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image


def to_tensor(path):                                  # torchvision ToTensor (stereo_dataset.py:90-91)
  arr = np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)
  return torch.from_numpy(arr.astype(np.float32) / np.float32(255.0)).permute(2, 0, 1).contiguous()


def pfm(path):                                        # utils/io.py:37-80
  with open(path, "rb") as f:
    color = f.readline().rstrip().decode("ascii") == "PF"
    w, h = map(int, re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("ascii")).groups())
    scale = float(f.readline().decode("ascii").rstrip())
    data = np.frombuffer(f.read(), dtype=("<f4" if scale < 0 else ">f4"))
  return np.flipud(data.reshape((h, w, 3) if color else (h, w))).astype(np.float32)


def load_disp(dataset, path):                         # utils/dataset_utils.py:26-57
  if dataset.startswith("SceneFlow"):
    return torch.from_numpy(pfm(path).copy()).unsqueeze(0)
  if dataset in ("KittiStereo2015", "KittiStereo2012"):
    return torch.from_numpy(np.array(Image.open(path)).astype(np.float32) / 256.0).unsqueeze(0)
  if dataset == "KittiRaw":
    return (torch.from_numpy(np.load(path).astype(np.float32)) / 128.0).unsqueeze(0)
  if dataset == "VirtualKitti":
    depth = 0.01 * np.array(Image.open(path)).astype(np.float64)
    return torch.from_numpy((0.532725 * 725.0087 / depth).astype(np.float32)).unsqueeze(0)
  raise KeyError(dataset)


def sample(dataset, paths, height, width, scales, window, flip, load_left=True, load_right=True):
  """One sample given the (already drawn) crop window (i, j) and flip decision."""
  rgb_l, rgb_r = to_tensor(paths[0]), to_tensor(paths[1])
  disp_l = load_disp(dataset, paths[2]) if load_left else None
  disp_r = load_disp(dataset, paths[3]) if load_right else None
  if flip:                                            # stereo_dataset.py:66-70
    rgb_l, rgb_r = torch.flip(rgb_r, dims=(-1,)), torch.flip(rgb_l, dims=(-1,))
    if disp_l is not None and disp_r is not None:
      disp_l, disp_r = torch.flip(disp_r, dims=(-1,)), torch.flip(disp_l, dims=(-1,))
  i, j = window
  crop = lambda t: None if t is None else t[:, i:i + height, j:j + width]
  rgb_l, rgb_r, disp_l, disp_r = crop(rgb_l), crop(rgb_r), crop(disp_l), crop(disp_r)
  out = {"color_l/0": rgb_l, "color_r/0": rgb_r}
  if load_left: out["gt_disp_l/0"] = disp_l
  if load_right: out["gt_disp_r/0"] = disp_r
  for s in scales:
    if s == 0:
      continue
    size = (height // 2 ** s, width // 2 ** s)
    rs = lambda t: F.interpolate(t.unsqueeze(0), size=size, mode="bilinear", align_corners=False).squeeze(0)
    out["color_l/%d" % s], out["color_r/%d" % s] = rs(rgb_l), rs(rgb_r)
    if load_left: out["gt_disp_l/%d" % s] = rs(disp_l) / 2 ** s
    if load_right: out["gt_disp_r/%d" % s] = rs(disp_r) / 2 ** s
  return out
