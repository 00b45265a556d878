"""Dataset layer, device path (as_decode_rgb8 / as_decode_plane / as_upsample_bilinear_fwd) against the host path and
the oracle: same random decisions, same sample dictionary, on the GPU."""
import os
import random

import numpy as np

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


def test_dataset_to_evaluate_end_to_end(tmp_path):
  """A KITTI-format tree -> StereoDataset (device path) -> torch DataLoader -> train.evaluate, as evaluate_model.py:34-70
  wires them; expected metrics from the oracle's forward on the host-path samples (train.py:98-121: per batch, then mean)."""
  import train as train_surface
  from torch.utils.data import DataLoader
  from adaptive_stereo.models.stereo_net import StereoNet, FeatureExtractorNetwork
  from adaptive_stereo.utils import synthetic as syn
  from oracle import stereo_oracle as orc
  K, MAXDISP, H, W = 3, 64, 64, 96
  data, splits = make_tree(str(tmp_path), "KittiStereo2015", n=4, H0=70, W0=110, seed=3)
  dev = StereoDataset(data, "KittiStereo2015", "tiny", H, W, "train", splits_path=splits, device="cuda:0")
  host = StereoDataset(data, "KittiStereo2015", "tiny", H, W, "train", splits_path=splits)
  loader = DataLoader(dev, batch_size=2, shuffle=False, num_workers=0)
  fnet, snet = FeatureExtractorNetwork(K), StereoNet(K, 1, 0, maxdisp=MAXDISP)
  fsd = syn.synthetic_state_dict(fnet.state_dict(), seed=123)
  ssd = syn.synthetic_state_dict(snet.state_dict(), seed=123, logit_gain=5.0)
  fnet.load_state_dict(fsd); snet.load_state_dict(ssd)
  opt = train_surface.TrainOptions().parse(["--stereonet_k", str(K)])
  m = train_surface.evaluate(fnet.to("cuda:0"), snet.to("cuda:0"), loader, opt)
  epes, d3 = [], []
  for i in (0, 2):
    a, b = host[i], host[i + 1]
    left, right = torch.stack([a["color_l/0"], b["color_l/0"]]), torch.stack([a["color_r/0"], b["color_r/0"]])
    gt = torch.stack([a["gt_disp_l/0"], b["gt_disp_l/0"]])
    out, _ = orc.forward_only(fsd, ssd, left, right, K, 0, MAXDISP)
    v = gt > 0
    err = (out["pred_disp_l/0"] - gt).abs()
    epes.append(float(err[v].mean())); d3.append(float((v * (err > 3)).sum() / float(v.sum())))
  assert abs(m["EPE"] - sum(epes) / 2) <= 1e-3 * max(1.0, sum(epes) / 2)
  assert abs(m["D1_all_3px"] - sum(d3) / 2) <= 2e-3


def test_device_prefetcher_equals_direct_indexing(tmp_path):
  """Thread-pool parsing + side-stream decoding one batch ahead: same batches, same random decisions, as indexing the
  dataset sample by sample (ragged last batch, drop_last, shuffle with a seed)."""
  from adaptive_stereo.datasets.prefetch import DevicePrefetcher
  data, splits = make_tree(str(tmp_path), "KittiStereo2015", n=7, H0=40, W0=70, seed=5)
  kw = dict(scales=[0, 1], do_hflip=True, random_crop=True, splits_path=splits, device="cuda:0")
  ds = StereoDataset(data, "KittiStereo2015", "tiny", 32, 64, "train", **kw)
  random.seed(11)
  ref = [ds[i] for i in range(len(ds))]
  random.seed(11)
  got = list(DevicePrefetcher(ds, batch_size=3, num_threads=4))
  assert [b["color_l/0"].shape[0] for b in got] == [3, 3, 1] and len(DevicePrefetcher(ds, 3)) == 3
  torch.cuda.synchronize()
  k = 0
  for b in got:
    for r in range(b["color_l/0"].shape[0]):
      for key in ref[k]:
        assert torch.equal(b[key][r], ref[k][key]), (k, key)
      k += 1
  assert len(list(DevicePrefetcher(ds, batch_size=3, drop_last=True))) == 2 == len(DevicePrefetcher(ds, 3, drop_last=True))
  random.seed(5); a = [b["color_l/0"].sum().item() for b in DevicePrefetcher(ds, 2, shuffle=True, seed=3)]
  random.seed(5); c = [b["color_l/0"].sum().item() for b in DevicePrefetcher(ds, 2, shuffle=True, seed=3)]
  assert len(a) == 4 and a == c                        # same shuffle seed and same draws: same epoch
  with pytest.raises(ValueError):
    DevicePrefetcher(StereoDataset(data, "KittiStereo2015", "tiny", 32, 64, "train", splits_path=splits), 2)


def test_prefetched_stream_drives_the_adaptation_step(tmp_path):
  """KITTI-format tree -> DevicePrefetcher (batch 1, the reference's online setting) -> OnlineAdapter.step: the losses
  equal those of stepping on the same samples taken by direct indexing (the decode stream hands over through events)."""
  from adaptive_stereo.adaptation import OnlineAdapter
  from adaptive_stereo.datasets.prefetch import DevicePrefetcher
  from adaptive_stereo.models.stereo_net import StereoNet, FeatureExtractorNetwork
  from adaptive_stereo.utils import synthetic as syn
  K, MAXDISP, H, W = 3, 64, 64, 96
  data, splits = make_tree(str(tmp_path), "KittiRaw", n=4, H0=70, W0=110, seed=9)
  ds = StereoDataset(data, "KittiRaw", "tiny", H, W, "train", splits_path=splits, device="cuda:0", load_disp_right=False)

  def run(batches):
    fnet, snet = FeatureExtractorNetwork(K), StereoNet(K, 1, 0, maxdisp=MAXDISP)
    fnet.load_state_dict(syn.synthetic_state_dict(fnet.state_dict(), seed=123))
    snet.load_state_dict(syn.synthetic_state_dict(snet.state_dict(), seed=123, logit_gain=5.0))
    adapter = OnlineAdapter(fnet.to("cuda:0"), snet.to("cuda:0"), H, W, lr=5e-5)
    return [float(adapter.step(b["color_l/0"], b["color_r/0"])["loss"]) for b in batches]

  direct = [{k: v.unsqueeze(0) for k, v in ds[i].items()} for i in range(len(ds))]
  a = run(direct)
  b = run(DevicePrefetcher(ds, batch_size=1, num_threads=2))
  assert a == b and all(x == x and 0.0 < x < 10.0 for x in a), (a, b)


def test_device_decoder_on_the_reference_sample_pfm(tmp_path):
  """The reference's own sample (resources/0008.pfm, 960x540 SceneFlow disparity) through the DEVICE path — raw samples
  uploaded bottom row first, flipped / cropped / mirrored by as_decode_plane — against the reference's readPFM output
  stored in tests/golden/dataset/pfm_0008_expected.npz (bit for bit: the decoder only moves float32 samples)."""
  import shutil
  from PIL import Image
  from conftest import REPO
  pin = os.path.join(REPO, "tests", "golden", "dataset")
  exp = np.load(os.path.join(pin, "pfm_0008_expected.npz"))
  data = str(tmp_path / "data"); os.makedirs(data)
  for name in ("l.png", "r.png"):
    Image.fromarray(np.random.RandomState(3).randint(0, 256, size=(540, 960, 3)).astype(np.uint8)).save(os.path.join(data, name))
  shutil.copyfile(os.path.join(pin, "0008.pfm"), os.path.join(data, "d.pfm"))
  splits = str(tmp_path / "splits" / "one"); os.makedirs(splits)
  with open(os.path.join(splits, "train_lines.txt"), "w") as f:
    f.write("l.png r.png d.pfm d.pfm\n")
  ds = StereoDataset(data, "SceneFlowFlying", "one", 540, 960, "train", scales=[0], splits_path=str(tmp_path / "splits"),
                     device="cuda")
  full = ds[0]["gt_disp_l/0"].cpu().numpy()[0]
  assert full.shape == (540, 960)
  assert np.array_equal(full[0], exp["first_row"]) and np.array_equal(full[-1], exp["last_row"])
  assert np.array_equal(full.reshape(-1)[::int(exp["sub_stride"])], exp["sub"])
  d = full.astype(np.float64)
  assert d.sum() == float(exp["sum"]) and (d * d).sum() == float(exp["sumsq"])
  # centre crop to the reference's training size (320x960) and a mirrored, swapped pair: rows/columns of the same image
  crop = StereoDataset(data, "SceneFlowFlying", "one", 320, 960, "train", scales=[0], splits_path=str(tmp_path / "splits"),
                       device="cuda")
  assert np.array_equal(crop[0]["gt_disp_l/0"].cpu().numpy()[0], full[110:430])
  raw = crop.parse(0)
  mirrored = crop.decode(raw, 110, 0, True)["gt_disp_l/0"].cpu().numpy()[0]
  assert np.array_equal(mirrored, full[110:430, ::-1])
