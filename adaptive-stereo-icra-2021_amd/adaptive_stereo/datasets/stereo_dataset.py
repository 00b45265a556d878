"""Dataset layer.  Reference: adaptive_stereo/datasets/stereo_dataset.py:13-186.

Same constructor, same sample dictionary (``color_{l,r}/{s}``, ``gt_disp_{l,r}/{s}``), same semantics:
  * a split manifest line holds four paths relative to ``dataset_path``: left image, right image, left and right
    ground-truth disparity (stereo_dataset.py:86-88);
  * images -> float [3,H,W] in [0,1] (torchvision ToTensor, :90-91); disparities through the dataset's decoder (:93-94);
  * optional horizontal flip with probability 1/2 — both views mirrored AND swapped, ground truth likewise — then a
    random or centred crop to (height, width) (:49-77);
  * every extra scale s: bilinear resize (align_corners=False) to (height // 2^s, width // 2^s), disparity / 2^s (:98-135).

Two execution paths with identical results:
  * ``device=None``: host tensors through plain torch ops, what a DataLoader worker of the reference produces;
  * ``device="cuda"``: the host only parses the files (PIL / PFM / NPY headers) and uploads the raw samples; crop, flip,
    uint8 / uint16 -> float, scaling and depth -> disparity run in two HIP gather kernels (as_decode_rgb8,
    as_decode_plane) and the pyramid in as_upsample_bilinear_fwd — no full-size host float copies, and the sample is
    born where the network consumes it.  Raises if the HIP library is missing (no silent fallback).
The split manifests live outside the package (the reference keeps them under ``splits/``): pass ``splits_path``."""
import os
import random

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image
from torch.utils.data import Dataset

from ..utils.dataset_utils import read_lines, get_disp_loader, get_raw_disp_loader, flip_stereo_pair

BASELINE_METERS = {
  "KittiStereo2012": 0.54, "KittiStereo2015": 0.54, "KittiRaw": 0.54,
  "SceneFlowFlying": 1.0, "SceneFlowMonkaa": 1.0, "SceneFlowDriving": 1.0,
  "VirtualKitti": 0.532725,
}


class StereoDataset(Dataset):
  def __init__(self, dataset_path, dataset_name, split, height, width, subsplit, scales=[0], do_hflip=False,
               random_crop=False, load_disp_left=True, load_disp_right=True, splits_path=None, device=None):
    super(StereoDataset, self).__init__()
    self.dataset_path = dataset_path
    self.dataset = dataset_name
    self.height, self.width = height, width
    self.scales = list(scales)
    self.do_hflip, self.random_crop = do_hflip, random_crop
    self.load_disp_left, self.load_disp_right = load_disp_left, load_disp_right
    if splits_path is None:
      splits_path = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "splits"))
    self.lines = read_lines(os.path.join(splits_path, split, "{}_lines.txt".format(subsplit)))
    self.load_disp_fn = get_disp_loader(dataset_name)
    self.load_raw_disp_fn = get_raw_disp_loader(dataset_name)
    self.device = None if device is None else torch.device(device)

  def __len__(self):
    return len(self.lines)

  # -- sampling decisions (shared by both paths) ------------------------------------------------------
  def _window(self, H0, W0):
    assert self.height <= H0 and self.width <= W0
    if self.random_crop:
      i = random.randint(0, H0 - self.height)
      j = random.randint(0, W0 - self.width)
    else:
      i, j = (H0 - self.height) // 2, (W0 - self.width) // 2
    return i, j

  def _paths(self, index):
    return [os.path.join(self.dataset_path, p) for p in self.lines[index].split(" ")]

  def __getitem__(self, index):
    if self.device is None:
      return self._getitem_host(index)
    return self._getitem_device(index)

  # -- host path: the reference's own sequence of torch ops -------------------------------------------
  def _getitem_host(self, index):
    rgb_l_path, rgb_r_path, disp_l_path, disp_r_path = self._paths(index)
    rgb_l = _to_tensor(Image.open(rgb_l_path))
    rgb_r = _to_tensor(Image.open(rgb_r_path))
    disp_l = self.load_disp_fn(disp_l_path) if self.load_disp_left else None
    disp_r = self.load_disp_fn(disp_r_path) if self.load_disp_right else None
    i, j = self._window(rgb_l.shape[-2], rgb_l.shape[-1])
    if self.do_hflip and random.random() < 0.5:
      rgb_l, rgb_r = flip_stereo_pair(rgb_l, rgb_r)
      if disp_l is not None and disp_r is not None:
        disp_l, disp_r = flip_stereo_pair(disp_l, disp_r)
    h, w = self.height, self.width
    crop = lambda t: None if t is None else t[:, i:i + h, j:j + w]
    rgb_l, rgb_r, disp_l, disp_r = crop(rgb_l), crop(rgb_r), crop(disp_l), crop(disp_r)
    out = {}
    for s in self.scales:
      if s == 0:
        continue
      size = (h // 2 ** s, w // 2 ** s)
      resize = lambda t: F.interpolate(t.unsqueeze(0), size=size, mode="bilinear", align_corners=False).squeeze(0)
      out["color_l/{}".format(s)] = resize(rgb_l)
      out["color_r/{}".format(s)] = resize(rgb_r)
      if self.load_disp_left:
        out["gt_disp_l/{}".format(s)] = resize(disp_l) / 2 ** s
      if self.load_disp_right:
        out["gt_disp_r/{}".format(s)] = resize(disp_r) / 2 ** s
    out["color_l/0"], out["color_r/0"] = rgb_l, rgb_r
    if self.load_disp_left:
      out["gt_disp_l/0"] = disp_l
    if self.load_disp_right:
      out["gt_disp_r/0"] = disp_r
    return out

  # -- device path: raw samples up, HIP gather kernels ------------------------------------------------
  def _getitem_device(self, index):
    raw = self.parse(index)
    i, j = self._window(raw[0].shape[0], raw[0].shape[1])
    flip = bool(self.do_hflip and random.random() < 0.5)
    return self.decode(raw, i, j, flip)

  def parse(self, index):
    """Host half of the device path: file parsing only (PIL / PFM / NPY) -> (rgb_l, rgb_r uint8 [H0,W0,3], disp_l,
    disp_r RawPlane or None).  No random decisions, no torch: safe to run in worker threads (datasets/prefetch.py)."""
    rgb_l_path, rgb_r_path, disp_l_path, disp_r_path = self._paths(index)
    raw_l = _rgb8(Image.open(rgb_l_path))
    raw_r = _rgb8(Image.open(rgb_r_path))
    disp_l = self.load_raw_disp_fn(disp_l_path) if self.load_disp_left else None
    disp_r = self.load_raw_disp_fn(disp_r_path) if self.load_disp_right else None
    return raw_l, raw_r, disp_l, disp_r

  def decode(self, raw, i, j, flip):
    """Device half: upload the raw samples and run crop / flip / conversion / pyramid on the current HIP stream."""
    from .. import _native as nat
    dev = self.device
    raw_l, raw_r, disp_l, disp_r = raw
    H0, W0 = raw_l.shape[0], raw_l.shape[1]
    flip_disp = flip and disp_l is not None and disp_r is not None           # stereo_dataset.py:69-70
    if flip:
      raw_l, raw_r = raw_r, raw_l                                             # mirrored in the kernel, swapped here
    if flip_disp:
      disp_l, disp_r = disp_r, disp_l
    h, w = self.height, self.width
    st = nat.stream()

    def colour(arr):
      src = torch.from_numpy(arr).to(dev, non_blocking=True)
      dst = torch.empty(3, h, w, dtype=torch.float32, device=dev)
      nat.call("as_decode_rgb8", nat.ptr(src), H0, W0, i, j, h, w, int(flip), nat.ptr(dst), st)
      return dst

    def plane(rp, mirrored):
      if rp is None:
        return None
      smp = rp.samples
      assert smp.shape[0] >= i + h and smp.shape[1] >= j + w
      src = torch.from_numpy(smp.view(np.int16) if smp.dtype == np.uint16 else smp).to(dev, non_blocking=True)
      dst = torch.empty(1, h, w, dtype=torch.float32, device=dev)
      nat.call("as_decode_plane", nat.ptr(src), rp.dtype_code, smp.shape[0], smp.shape[1], i, j, h, w, int(mirrored),
               int(rp.vflip), rp.scale, int(rp.reciprocal), nat.ptr(dst), st)
      return dst

    rgb_l, rgb_r = colour(raw_l), colour(raw_r)
    d_l, d_r = plane(disp_l, flip_disp), plane(disp_r, flip_disp)
    out = {}
    for s in self.scales:
      if s == 0:
        continue
      hs, ws = h // 2 ** s, w // 2 ** s

      def resize(t, gain):
        c = t.shape[0]
        dst = torch.empty(c, hs, ws, dtype=torch.float32, device=dev)
        nat.call("as_upsample_bilinear_fwd", nat.ptr(t), c, h, w, nat.ptr(dst), hs, ws, float(gain), st)
        return dst
      out["color_l/{}".format(s)] = resize(rgb_l, 1.0)
      out["color_r/{}".format(s)] = resize(rgb_r, 1.0)
      if self.load_disp_left:
        out["gt_disp_l/{}".format(s)] = resize(d_l, 1.0 / 2 ** s)
      if self.load_disp_right:
        out["gt_disp_r/{}".format(s)] = resize(d_r, 1.0 / 2 ** s)
    out["color_l/0"], out["color_r/0"] = rgb_l, rgb_r
    if self.load_disp_left:
      out["gt_disp_l/0"] = d_l
    if self.load_disp_right:
      out["gt_disp_r/0"] = d_r
    return out

  # -- calibration constants (stereo_dataset.py:145-186) ----------------------------------------------
  def get_baseline_meters(self):
    return BASELINE_METERS[self.dataset]

  def get_intrinsics_normalized(self):
    if self.dataset in ("KittiStereo2012", "KittiStereo2015", "KittiRaw"):
      return torch.Tensor([[0.5885, 0.0, 0.4972], [0.0, 1.9501, 0.4972], [0.0, 0.0, 1.0]])
    if "SceneFlow" in self.dataset:
      return torch.Tensor([[1.09375, 0.0, 0.5], [0, 1.94444, 0.5], [0.0, 0.0, 1.0]])
    raise NotImplementedError("no intrinsics recorded for {}".format(self.dataset))

  def get_intrinsics(self, height, width):
    K = self.get_intrinsics_normalized().clone()
    K[0] *= width
    K[1] *= height
    return K


def _rgb8(img):
  """PIL image -> uint8 [H,W,3] (what torchvision's ToTensor sees before its division by 255)."""
  if img.mode != "RGB":
    img = img.convert("RGB")
  return np.array(img, dtype=np.uint8)           # a writable, contiguous copy


def _to_tensor(img):
  """torchvision.transforms.ToTensor for 8-bit images: uint8 HWC -> float CHW / 255."""
  return torch.from_numpy(_rgb8(img)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
