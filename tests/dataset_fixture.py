"""Writes a tiny dataset tree in each of the reference's formats (used by the CPU and the GPU dataset tests)."""
import os

import numpy as np
from PIL import Image


def write_pfm_bytes(path, image, little=True):
  """Independent of the product's writer: PFM with an explicit byte order."""
  h, w = image.shape[:2]
  with open(path, "wb") as f:
    f.write(("Pf\n%d %d\n%s\n" % (w, h, "-1.0" if little else "1.0")).encode("ascii"))
    f.write(np.flipud(image).astype("<f4" if little else ">f4").tobytes())


def make_tree(root, dataset, n=3, H0=37, W0=61, seed=0):
  """-> (dataset_path, splits_path): n samples, one manifest splits/<split>/train_lines.txt."""
  rng = np.random.RandomState(seed)
  data = os.path.join(root, "data"); os.makedirs(data, exist_ok=True)
  lines = []
  for k in range(n):
    names = []
    for side in ("l", "r"):
      name = "rgb_%s_%d.png" % (side, k)
      Image.fromarray(rng.randint(0, 256, size=(H0, W0, 3)).astype(np.uint8)).save(os.path.join(data, name))
      names.append(name)
    for side in ("l", "r"):
      if dataset.startswith("SceneFlow"):
        name = "disp_%s_%d.pfm" % (side, k)
        write_pfm_bytes(os.path.join(data, name), (rng.rand(H0, W0) * 190).astype(np.float32), little=(k % 2 == 0))
      elif dataset in ("KittiStereo2015", "KittiStereo2012"):
        name = "disp_%s_%d.png" % (side, k)
        Image.fromarray(rng.randint(0, 192 * 256, size=(H0, W0)).astype(np.uint16)).save(os.path.join(data, name))
      elif dataset == "KittiRaw":
        name = "disp_%s_%d.npy" % (side, k)
        np.save(os.path.join(data, name), rng.randint(0, 192 * 128, size=(H0, W0)).astype(np.uint16))
      else:                      # VirtualKitti: depth in centimetres, 16-bit PNG
        name = "depth_%s_%d.png" % (side, k)
        Image.fromarray(rng.randint(100, 65535, size=(H0, W0)).astype(np.uint16)).save(os.path.join(data, name))
      names.append(name)
    lines.append(" ".join(names))
  splits = os.path.join(root, "splits", "tiny"); os.makedirs(splits, exist_ok=True)
  with open(os.path.join(splits, "train_lines.txt"), "w") as f:
    f.write("\n".join(lines) + "\n")
  return data, os.path.join(root, "splits")
