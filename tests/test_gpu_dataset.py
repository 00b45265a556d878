"""Dataset layer, device path (as_decode_rgb8 / as_decode_plane / as_upsample_bilinear_fwd) against the host path and
the oracle: same random decisions, same sample dictionary, on the GPU."""
import os
import random

import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import REPO, PKG  # noqa: F401
from adaptive_stereo.datasets.stereo_dataset import StereoDataset
from oracle import dataset_oracle as dorc
from dataset_fixture import make_tree

DATASETS = ["SceneFlowFlying", "KittiStereo2015", "KittiRaw", "VirtualKitti"]


@pytest.mark.parametrize("dataset", DATASETS)
@pytest.mark.parametrize("do_hflip,random_crop", [(False, False), (True, True)])
def test_device_path_equals_host_path_and_oracle(tmp_path, dataset, do_hflip, random_crop):
  data, splits = make_tree(str(tmp_path), dataset, n=4, H0=45, W0=83)
  H, W, scales = 32, 64, [0, 1, 2, 3]
  kw = dict(scales=scales, do_hflip=do_hflip, random_crop=random_crop, splits_path=splits)
  host = StereoDataset(data, dataset, "tiny", H, W, "train", **kw)
  dev = StereoDataset(data, dataset, "tiny", H, W, "train", device="cuda:0", **kw)
  for idx in range(len(host)):
    random.seed(7 + idx); a = host[idx]
    random.seed(7 + idx); b = dev[idx]
    random.seed(7 + idx)
    window = host._window(45, 83)
    flip = bool(do_hflip and random.random() < 0.5)
    ref = dorc.sample(dataset, [os.path.join(data, p) for p in host.lines[idx].split(" ")], H, W, scales, window, flip)
    assert set(a.keys()) == set(b.keys()) == set(ref.keys())
    for key in a:
      g = b[key]
      assert g.is_cuda and g.dtype == torch.float32 and g.shape == a[key].shape, key
      g = g.cpu()
      if key.endswith("/0") and dataset != "VirtualKitti":
        assert torch.equal(g, a[key]) and torch.equal(g, ref[key]), key      # integer decode: exact
      else:
        tol = 2e-6 * max(1.0, float(ref[key].abs().max()))
        assert float((g - a[key]).abs().max()) <= tol and float((g - ref[key]).abs().max()) <= tol, key


def test_device_path_single_disparity_and_flip_rule(tmp_path):
  """Ground truth is flipped only when BOTH maps are loaded (stereo_dataset.py:69-70); the images always are."""
  data, splits = make_tree(str(tmp_path), "KittiStereo2012", n=2, H0=20, W0=50)
  kw = dict(scales=[0], do_hflip=True, load_disp_right=False, splits_path=splits)
  host = StereoDataset(data, "KittiStereo2012", "tiny", 16, 48, "train", **kw)
  dev = StereoDataset(data, "KittiStereo2012", "tiny", 16, 48, "train", device="cuda:0", **kw)
  for seed in range(6):
    random.seed(seed); a = host[seed % 2]
    random.seed(seed); b = dev[seed % 2]
    assert set(a) == set(b) == {"color_l/0", "color_r/0", "gt_disp_l/0"}
    for key in a:
      assert torch.equal(b[key].cpu(), a[key]), (seed, key)
