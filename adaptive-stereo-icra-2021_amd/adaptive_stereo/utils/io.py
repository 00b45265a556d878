"""PFM reader (reference: adaptive_stereo/utils/io.py:37-80, the SceneFlow release's reader).

Format: line 1 "PF" (3 channels) or "Pf" (1 channel); line 2 "<width> <height>"; line 3 a scale whose SIGN gives the
byte order (negative = little endian); then height*width(*3) float32 samples stored BOTTOM row first.
``read_pfm_raw`` returns the samples as stored (bottom-up) so that the device-side decoder can flip while it crops;
``read_pfm`` / ``read_pfm_tensor`` return the image top row first, as the reference does."""
import re

import numpy as np
import torch


def read_pfm_raw(path):
  """-> (samples as stored: float32 [H,W] or [H,W,3], bottom row first; scale)"""
  with open(path, "rb") as f:
    magic = f.readline().decode("ascii").rstrip()
    if magic not in ("PF", "Pf"):
      raise ValueError("%s: not a PFM file" % path)
    dims = re.match(r"^(\d+)\s+(\d+)\s*$", f.readline().decode("ascii"))
    if dims is None:
      raise ValueError("%s: malformed PFM header" % path)
    width, height = int(dims.group(1)), int(dims.group(2))
    scale = float(f.readline().decode("ascii").rstrip())
    order = "<" if scale < 0 else ">"
    count = height * width * (3 if magic == "PF" else 1)
    data = np.frombuffer(f.read(4 * count), dtype=order + "f4")
    if data.size != count:
      raise ValueError("%s: truncated PFM payload" % path)
  shape = (height, width, 3) if magic == "PF" else (height, width)
  return data.astype(np.float32, copy=False).reshape(shape), abs(scale)


def read_pfm(path):
  data, scale = read_pfm_raw(path)
  return np.flipud(data), scale


def read_pfm_tensor(path):
  return torch.from_numpy(np.ascontiguousarray(read_pfm(path)[0]))


def write_pfm(path, image, scale=1.0):
  """float32 [H,W] or [H,W,3], top row first (the inverse of read_pfm); little endian."""
  image = np.asarray(image)
  if image.dtype != np.float32:
    raise ValueError("PFM samples must be float32")
  if image.ndim == 3 and image.shape[2] == 3:
    magic = "PF"
  elif image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1):
    magic = "Pf"
  else:
    raise ValueError("PFM images are [H,W], [H,W,1] or [H,W,3]")
  with open(path, "wb") as f:
    f.write(("%s\n%d %d\n%f\n" % (magic, image.shape[1], image.shape[0], -abs(scale))).encode("ascii"))
    f.write(np.flipud(image.reshape(image.shape[0], image.shape[1], -1)).astype("<f4").tobytes())
