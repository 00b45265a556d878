"""Dataset layer, host path against the oracle's restatement of the reference
(datasets/stereo_dataset.py:49-143, utils/dataset_utils.py:19-57, utils/io.py:37-80)."""
import os
import random

import numpy as np
import pytest
import torch

from conftest import REPO, PKG  # noqa: F401  (puts the package on sys.path)
from adaptive_stereo.datasets.stereo_dataset import StereoDataset
from adaptive_stereo.utils import io as as_io
from adaptive_stereo.utils.dataset_utils import flip_stereo_pair, read_lines
from oracle import dataset_oracle as dorc
from dataset_fixture import make_tree, write_pfm_bytes

DATASETS = ["SceneFlowFlying", "KittiStereo2015", "KittiRaw", "VirtualKitti"]


@pytest.mark.parametrize("little", [True, False])
def test_pfm_reader_both_byte_orders_and_round_trip(tmp_path, little):
  img = (np.random.RandomState(1).rand(9, 13) * 100).astype(np.float32)
  p = str(tmp_path / "a.pfm")
  write_pfm_bytes(p, img, little=little)
  got, scale = as_io.read_pfm(p)
  assert scale == 1.0 and np.array_equal(got, img)                  # top row first, as the reference returns it
  raw, _ = as_io.read_pfm_raw(p)
  assert np.array_equal(raw, np.flipud(img))                         # as stored: bottom row first
  assert torch.equal(as_io.read_pfm_tensor(p), torch.from_numpy(img))
  assert np.array_equal(dorc.pfm(p), img)
  q = str(tmp_path / "b.pfm")
  as_io.write_pfm(q, img)
  assert np.array_equal(as_io.read_pfm(q)[0], img)
  with open(str(tmp_path / "bad.pfm"), "wb") as f:
    f.write(b"P6\n1 1\n255\n")
  with pytest.raises(ValueError):
    as_io.read_pfm(str(tmp_path / "bad.pfm"))


def test_flip_stereo_pair_mirrors_and_swaps():
  l, r = torch.arange(6.).view(1, 2, 3), torch.arange(6., 12.).view(1, 2, 3)
  fl, fr = flip_stereo_pair(l, r)
  assert torch.equal(fl, torch.flip(r, dims=(-1,))) and torch.equal(fr, torch.flip(l, dims=(-1,)))


@pytest.mark.parametrize("dataset", DATASETS)
@pytest.mark.parametrize("do_hflip,random_crop", [(False, False), (True, True)])
def test_host_path_equals_the_oracle(tmp_path, dataset, do_hflip, random_crop):
  data, splits = make_tree(str(tmp_path), dataset, n=4)
  H, W, scales = 24, 40, [0, 1, 2]
  ds = StereoDataset(data, dataset, "tiny", H, W, "train", scales=scales, do_hflip=do_hflip, random_crop=random_crop,
                     splits_path=splits)
  assert len(ds) == 4 and len(read_lines(os.path.join(splits, "tiny", "train_lines.txt"))) == 4
  for idx in range(len(ds)):
    random.seed(100 + idx)
    got = ds[idx]
    # replay the same draws: window first, then the flip decision (stereo_dataset.py:54-66)
    random.seed(100 + idx)
    window = ds._window(37, 61)
    flip = bool(do_hflip and random.random() < 0.5)
    paths = [os.path.join(data, p) for p in ds.lines[idx].split(" ")]
    ref = dorc.sample(dataset, paths, H, W, scales, window, flip)
    assert set(got.keys()) == set(ref.keys())
    for key, exp in ref.items():
      g = got[key]
      assert g.dtype == torch.float32 and g.shape == exp.shape, key
      if key.endswith("/0") and dataset != "VirtualKitti":
        assert torch.equal(g, exp), key                    # integer decode, power-of-two scales: exact
      else:
        assert float((g - exp).abs().max()) <= 1e-6 * max(1.0, float(exp.abs().max())), key


def test_only_one_disparity_and_calibration(tmp_path):
  data, splits = make_tree(str(tmp_path), "KittiStereo2015", n=1)
  ds = StereoDataset(data, "KittiStereo2015", "tiny", 16, 32, "train", scales=[0, 1], load_disp_right=False,
                     splits_path=splits)
  s = ds[0]
  assert "gt_disp_r/0" not in s and "gt_disp_r/1" not in s and s["gt_disp_l/1"].shape == (1, 8, 16)
  assert ds.get_baseline_meters() == 0.54
  K = ds.get_intrinsics(375, 1242)
  assert abs(float(K[0, 0]) - 0.5885 * 1242) < 1e-3 and abs(float(K[1, 1]) - 1.9501 * 375) < 1e-3
  with pytest.raises(AssertionError):
    StereoDataset(data, "KittiStereo2015", "tiny", 64, 32, "train", splits_path=splits)[0]     # crop larger than image


def test_prefetcher_batching_logic_without_a_gpu():
  """Batch composition of DevicePrefetcher (order, ragged tail, drop_last, seeded shuffle) is host logic."""
  from adaptive_stereo.datasets.prefetch import DevicePrefetcher

  class Fake(object):
    device = "cuda:0"
    def __len__(self): return 10

  pf = DevicePrefetcher(Fake(), batch_size=4)
  assert list(pf._batches()) == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]] and len(pf) == 3
  pf = DevicePrefetcher(Fake(), batch_size=4, drop_last=True)
  assert list(pf._batches()) == [[0, 1, 2, 3], [4, 5, 6, 7]] and len(pf) == 2
  a = list(DevicePrefetcher(Fake(), batch_size=3, shuffle=True, seed=7)._batches())
  b = list(DevicePrefetcher(Fake(), batch_size=3, shuffle=True, seed=7)._batches())
  assert a == b and sorted(sum(a, [])) == list(range(10)) and a != [[0, 1, 2], [3, 4, 5], [6, 7, 8], [9]]
  pf = DevicePrefetcher(Fake(), batch_size=3, shuffle=True, seed=7)
  first, second = list(pf._batches()), list(pf._batches())
  assert first != second                                  # a new permutation every epoch, reproducible from the seed

  class Host(Fake):
    device = None
  with pytest.raises(ValueError):
    DevicePrefetcher(Host(), 2)


# ---- pinned against what the reference itself holds (tests/golden/make_dataset_golden.py) ----------------------------
PIN = os.path.join(REPO, "tests", "golden", "dataset")


def _pfm_expected():
  return np.load(os.path.join(PIN, "pfm_0008_expected.npz"))


def check_against_reference_pfm(img, scale):
  """img: what a reader returned for tests/golden/dataset/0008.pfm; expected: the reference's readPFM output."""
  exp = _pfm_expected()
  assert tuple(img.shape) == tuple(int(v) for v in exp["shape"]) == (540, 960) and img.dtype == np.float32
  assert float(scale) == float(exp["scale"])
  d = img.astype(np.float64)
  assert d.sum() == float(exp["sum"]) and (d * d).sum() == float(exp["sumsq"])       # same samples: exact sums
  assert np.array_equal(img[0], exp["first_row"]) and np.array_equal(img[-1], exp["last_row"])     # top row first
  assert np.array_equal(img.reshape(-1)[::int(exp["sub_stride"])], exp["sub"])
  assert float(img.min()) == float(exp["min"]) and float(img.max()) == float(exp["max"])


def test_pfm_reader_matches_the_reference_reader_on_its_own_sample():
  """resources/0008.pfm of the reference (a real 960x540 SceneFlow disparity) through the product's reader and through
  the dataset oracle: both must return, bit for bit, what the reference's readPFM returned (utils/io.py:37-80)."""
  path = os.path.join(PIN, "0008.pfm")
  img, scale = as_io.read_pfm(path)
  check_against_reference_pfm(np.ascontiguousarray(img), scale)
  check_against_reference_pfm(as_io.read_pfm_tensor(path).numpy(), scale)
  check_against_reference_pfm(np.ascontiguousarray(dorc.pfm(path)), 1.0)
  raw, _ = as_io.read_pfm_raw(path)
  assert np.array_equal(np.flipud(raw), img)
  # the SceneFlow decoder of the dataset layer returns it as [1,H,W] (utils/dataset_utils.py:26-27)
  from adaptive_stereo.utils.dataset_utils import get_disp_loader
  t = get_disp_loader("SceneFlowFlying")(path)
  assert tuple(t.shape) == (1, 540, 960)
  check_against_reference_pfm(t[0].numpy(), scale)
  assert torch.equal(dorc.load_disp("SceneFlowFlying", path), t)


def test_split_lengths_asserted_by_the_reference_test():
  """test/test_stereo_dataset.py:24-97 of the reference asserts the data set lengths; the KITTI-2015 manifests (the
  benchmark's data set) are committed as fixture data, the others are re-counted from /root/reference where it exists
  and otherwise checked against the counts recorded (with the manifests' sha256) when the fixture was made."""
  import hashlib
  import json
  rec = json.load(open(os.path.join(PIN, "split_lengths.json")))
  rows = {(r["split"], r["subsplit"]): r for r in rec["rows"]}
  assert {k: r["expected_by_reference_test"] for k, r in rows.items()} == {
      ("sceneflow_driving", "train"): 1540, ("sceneflow_driving", "val"): 330, ("sceneflow_driving", "test"): 330,
      ("sceneflow_flying", "train"): 19031, ("sceneflow_flying", "val"): 3359, ("sceneflow_flying", "test"): 4370,
      ("kitti_stereo_2012", "train"): 194, ("kitti_stereo_2012", "val"): 194, ("kitti_stereo_2012", "test"): 194,
      ("kitti_stereo_2015", "train"): 200, ("kitti_stereo_2015", "val"): 200, ("kitti_stereo_2015", "test"): 200}
  assert all(r["counted_by_product"] == r["expected_by_reference_test"] for r in rows.values())
  local = os.path.join(PIN, "splits")
  for sub in ("train", "val", "test"):
    d = StereoDataset("/nonexistent", "KittiStereo2015", "kitti_stereo_2015", 320, 960, sub, scales=[0],
                      do_hflip=False, random_crop=False, splits_path=local)
    assert len(d) == 200
    path = os.path.join(local, "kitti_stereo_2015", "%s_lines.txt" % sub)
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == rows[("kitti_stereo_2015", sub)]["sha256"]
    first = d._paths(0)
    assert len(first) == 4 and first[0].endswith("image_2/000000_10.png") and first[1].endswith("image_3/000000_10.png")
  ref_splits = "/root/reference/splits"
  if os.path.isdir(ref_splits):
    for (split, sub), r in rows.items():
      d = StereoDataset("/nonexistent", r["dataset"], split, 320, 960, sub, scales=[0], splits_path=ref_splits)
      assert len(d) == r["expected_by_reference_test"], (split, sub, len(d))
