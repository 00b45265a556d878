"""Pins the dataset layer against what the reference itself holds (run in the build container, CPU):

  1. /root/reference/resources/0008.pfm — a real 960x540 SceneFlow ground-truth disparity — is decoded by the REFERENCE's
     own reader (adaptive_stereo/utils/io.py:37-80, imported from /root/reference) and its output is stored as shape,
     scale, fp64 checksums, first / last row and a strided subsample in tests/golden/dataset/pfm_0008_expected.npz; the
     data file itself is copied next to it (a data file, 2 MB) so that the product's reader and the dataset oracle can be
     run on it anywhere (tests/test_dataset_cpu.py).
  2. The lengths the reference's own test asserts (test/test_stereo_dataset.py:24-97: 1540/330/330, 19031/3359/4370,
     194 x3, 200 x3; 21260 is not countable, see NOT_COUNTABLE) are re-counted from the reference's split manifests with the PRODUCT's StereoDataset /
     utils.dataset_utils.read_lines; the counts, the expected numbers and each manifest's sha256 go to
     tests/golden/dataset/split_lengths.json.  The three KITTI-2015 manifests (200 lines each, the benchmark's data set)
     are copied as fixture data so that the length test also runs where /root/reference does not exist.

The reference's dataset MODULE (datasets/stereo_dataset.py, utils/dataset_utils.py) still cannot be imported here
(torchvision, imageio, cv2 are absent), so crop / flip / pyramid logic stays pinned by the oracle's reading alone."""
import hashlib
import json
import os
import shutil
import sys

import numpy as np

REFERENCE = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "dataset")
PRODUCT = os.path.join(REPO, "adaptive-stereo-icra-2021_amd")     # put on sys.path only AFTER the reference's module is
# gone again: both packages are called adaptive_stereo, and a regular package beats the reference's namespace package

# (dataset_name, split, subsplit, length asserted by the reference's test) — test/test_stereo_dataset.py:24-97
EXPECTED = [
  ("SceneFlowDriving", "sceneflow_driving", "train", 1540), ("SceneFlowDriving", "sceneflow_driving", "val", 330),
  ("SceneFlowDriving", "sceneflow_driving", "test", 330),
  ("SceneFlowFlying", "sceneflow_flying", "train", 19031), ("SceneFlowFlying", "sceneflow_flying", "val", 3359),
  ("SceneFlowFlying", "sceneflow_flying", "test", 4370),
  ("KittiStereo2012", "kitti_stereo_2012", "train", 194), ("KittiStereo2012", "kitti_stereo_2012", "val", 194),
  ("KittiStereo2012", "kitti_stereo_2012", "test", 194),
  ("KittiStereo2015", "kitti_stereo_2015", "train", 200), ("KittiStereo2015", "kitti_stereo_2015", "val", 200),
  ("KittiStereo2015", "kitti_stereo_2015", "test", 200),
]
# test_stereo_dataset.py:94-97 also asserts 21260 for virtual_kitti_sim2real/train: that manifest is not in the reference
# tree (splits/virtual_kitti_sim2real/ holds only generate_split.py, which needs the data set on disk) — not countable.
NOT_COUNTABLE = [("VirtualKitti", "virtual_kitti_sim2real", "train", 21260)]


def reference_read_pfm(path):
  """The reference's reader, from the reference's file."""
  cwd = os.getcwd()
  os.chdir(REFERENCE); sys.path.insert(0, REFERENCE)
  try:
    import importlib
    ref_io = importlib.import_module("adaptive_stereo.utils.io")
    assert ref_io.__file__.startswith(REFERENCE), ref_io.__file__
    data, scale = ref_io.readPFM(path)
    tensor = ref_io.read_pfm_tensor(path)
  finally:
    os.chdir(cwd); sys.path.remove(REFERENCE)
    for name in [m for m in sys.modules if m == "adaptive_stereo" or m.startswith("adaptive_stereo.")]:
      del sys.modules[name]                       # the product package has the same name
  assert np.array_equal(np.asarray(data), tensor.numpy())
  return np.ascontiguousarray(data), float(scale)


def main():
  os.makedirs(OUT, exist_ok=True)
  src = os.path.join(REFERENCE, "resources", "0008.pfm")
  data, scale = reference_read_pfm(src)
  d64 = data.astype(np.float64)
  np.savez_compressed(os.path.join(OUT, "pfm_0008_expected.npz"), shape=np.array(data.shape), scale=np.array(scale),
                      dtype=np.array(str(data.dtype)), sum=np.array(d64.sum()), sumsq=np.array((d64 * d64).sum()),
                      min=np.array(data.min()), max=np.array(data.max()), first_row=data[0], last_row=data[-1],
                      sub=data.reshape(-1)[::127].copy(), sub_stride=np.array(127))
  shutil.copyfile(src, os.path.join(OUT, "0008.pfm"))
  print("0008.pfm: %s scale %g sum %.6f" % (data.shape, scale, d64.sum()))

  sys.path.insert(0, PRODUCT)
  from adaptive_stereo.datasets.stereo_dataset import StereoDataset
  from adaptive_stereo.utils.dataset_utils import read_lines
  splits = os.path.join(REFERENCE, "splits")
  rows = []
  for dataset, split, subsplit, want in EXPECTED:
    path = os.path.join(splits, split, "%s_lines.txt" % subsplit)
    ds = StereoDataset("/nonexistent", dataset, split, 320, 960, subsplit, scales=[0], splits_path=splits)
    lines = read_lines(path)
    assert len(ds) == len(lines) == want, (dataset, split, subsplit, len(ds), want)
    assert all(len(l.split()) == 4 for l in lines), path        # left, right, disp left, disp right (splits/README.md)
    rows.append({"dataset": dataset, "split": split, "subsplit": subsplit, "expected_by_reference_test": want,
                 "counted_by_product": len(ds), "sha256": hashlib.sha256(open(path, "rb").read()).hexdigest()})
    print("%-18s %-24s %-5s %6d ok" % (dataset, split, subsplit, want))
  json.dump({"source": "test/test_stereo_dataset.py:24-97 and splits/ of the reference", "rows": rows,
             "not_countable": [{"dataset": d, "split": s_, "subsplit": ss, "expected_by_reference_test": n,
                                "why": "manifest absent from the reference tree (only generate_split.py)"}
                               for d, s_, ss, n in NOT_COUNTABLE]},
            open(os.path.join(OUT, "split_lengths.json"), "w"), indent=1)
  dst = os.path.join(OUT, "splits", "kitti_stereo_2015"); os.makedirs(dst, exist_ok=True)
  for sub in ("train", "val", "test"):
    shutil.copyfile(os.path.join(splits, "kitti_stereo_2015", "%s_lines.txt" % sub), os.path.join(dst, "%s_lines.txt" % sub))


if __name__ == "__main__":
  main()
