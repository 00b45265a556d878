"""Split manifests and ground-truth disparity decoders (reference: adaptive_stereo/utils/dataset_utils.py:10-57).

Each decoder has two faces: ``load_disp_*`` returns the [1,H,W] float tensor the reference returns (host path), and
``raw_disp_*`` returns the file's samples untouched plus the recipe (dtype code, scale, reciprocal, vflip) that
``as_decode_plane`` applies on the device while it crops and flips (datasets/stereo_dataset.py, device path).
PNG files are read with PIL (the reference used imageio / cv2, neither of which changes the sample values)."""
import numpy as np
import torch
from PIL import Image

from .io import read_pfm_raw

VKITTI_BASELINE_M = 0.532725       # dataset_utils.py:42
VKITTI_FOCAL_PX = 725.0087         # dataset_utils.py:43


def read_lines(filename):
  with open(filename, "r") as f:
    return f.read().splitlines()


def flip_stereo_pair(l, r):
  """Mirror both images and swap them: the mirrored right view is a valid left view (dataset_utils.py:19-23)."""
  return torch.flip(r, dims=(-1,)), torch.flip(l, dims=(-1,))


class RawPlane(object):
  """One-channel samples as stored in the file + how as_decode_plane turns them into disparity."""

  def __init__(self, samples, scale, reciprocal=False, vflip=False):
    self.samples = np.ascontiguousarray(samples)
    if not self.samples.flags.writeable:            # np.frombuffer / PIL views: torch wants to own writable memory
      self.samples = self.samples.copy()
    if self.samples.dtype == np.float32:
      self.dtype_code = 0
    elif self.samples.dtype == np.uint16:
      self.dtype_code = 1
    elif self.samples.dtype == np.uint8:
      self.dtype_code = 2
    else:
      self.samples = self.samples.astype(np.float32)
      self.dtype_code = 0
    self.scale, self.reciprocal, self.vflip = float(scale), bool(reciprocal), bool(vflip)

  def to_tensor(self):
    """The reference's host-side result: float32 [1,H,W], top row first."""
    v = self.samples.astype(np.float32)
    if self.vflip:
      v = np.flipud(v)
    v = np.float32(self.scale) / v if self.reciprocal else v * np.float32(self.scale)
    return torch.from_numpy(np.ascontiguousarray(v.astype(np.float32))).unsqueeze(0)


def _png_samples(path):
  img = Image.open(path)
  arr = np.array(img)
  if arr.ndim == 3:                       # a colour PNG used as a one-channel map: first channel
    arr = arr[..., 0]
  if arr.dtype == np.int32:               # PIL mode "I": 16-bit PNG widened
    arr = arr.astype(np.uint16)
  return arr


def raw_disp_sceneflow(path):             # dataset_utils.py:26-27 (PFM, bottom-up)
  data, _ = read_pfm_raw(path)
  if data.ndim == 3:
    data = data[..., 0]
  return RawPlane(data, 1.0, vflip=True)


def raw_disp_kitti_stereo(path):          # :30-31  uint16 PNG / 256
  return RawPlane(_png_samples(path), 1.0 / 256.0)


def raw_disp_kitti_raw(path):             # :34-35  .npy / 128
  return RawPlane(np.load(path), 1.0 / 128.0)


def raw_disp_vkitti(path):                # :38-47  depth in cm -> disparity = baseline * focal / (0.01 * depth)
  return RawPlane(_png_samples(path), VKITTI_BASELINE_M * VKITTI_FOCAL_PX / 0.01, reciprocal=True)


_RAW_LOADERS = {
  "SceneFlowDriving": raw_disp_sceneflow, "SceneFlowFlying": raw_disp_sceneflow, "SceneFlowMonkaa": raw_disp_sceneflow,
  "KittiStereo2015": raw_disp_kitti_stereo, "KittiStereo2012": raw_disp_kitti_stereo,
  "KittiRaw": raw_disp_kitti_raw, "VirtualKitti": raw_disp_vkitti,
}


def get_raw_disp_loader(dataset_name):
  return _RAW_LOADERS[dataset_name]


def load_disp_sceneflow(path): return raw_disp_sceneflow(path).to_tensor()
def load_disp_kitti_stereo(path): return raw_disp_kitti_stereo(path).to_tensor()
def load_disp_kitti_raw(path): return raw_disp_kitti_raw(path).to_tensor()
def load_disp_vkitti(path): return raw_disp_vkitti(path).to_tensor()


def get_disp_loader(dataset_name):
  """dataset name -> function(path) -> float32 [1,H,W] (dataset_utils.py:50-57)."""
  raw = _RAW_LOADERS[dataset_name]
  return lambda path: raw(path).to_tensor()
