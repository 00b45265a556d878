"""The part of the reference's train.py that other scripts import (adapt.py:13, evaluate_model.py:10):
TrainOptions, process_batch, evaluate, log_scalars, log_images, save_models — SURVEY.md §8b / §8f-3.

``evaluate`` keeps the reference's definition of the metrics (train.py:74-126: per-batch EPE over gt>0,
D1-all at 2/3/4/5 px, FCS mean; then the mean over batches) but computes each batch's reductions in one
fused kernel (as_eval_metrics) and reads nothing back until the loop has finished.
Supervised training itself (train.train) and the dataset layer are out of scope.
"""
import argparse
import os

import torch

from adaptive_stereo import _native as nat
from adaptive_stereo.utils.feature_contrast import feature_contrast_mean


class TrainOptions(object):
  """Same flags and defaults as the reference (train.py:246-301)."""

  def __init__(self):
    p = argparse.ArgumentParser(description="Options for StereoNet adaptation on MI355X")
    p.add_argument("--height", type=int, default=320)
    p.add_argument("--width", type=int, default=960)
    p.add_argument("--model_name", type=str)
    p.add_argument("--stereonet_input_scale", default=0, type=int)
    p.add_argument("--stereonet_k", type=int, default=3, choices=[3, 4])
    p.add_argument("--dataset_path", type=str)
    p.add_argument("--dataset_name", type=str, default="SceneFlowDriving")
    p.add_argument("--split", type=str)
    p.add_argument("--batch_size", type=int, default=2)
    p.add_argument("--do_hflip", action="store_true", default=False)
    p.add_argument("--no_shuffle", action="store_true", default=False)
    p.add_argument("--use_grayscale", action="store_true")
    p.add_argument("--log_dir", type=str, default="/home/milo/training_logs")
    p.add_argument("--load_weights_folder", default=None, type=str)
    p.add_argument("--load_adam", action="store_true", default=False)
    p.add_argument("--scheduler_step_size", default=5, type=int)
    p.add_argument("--num_workers", type=int, default=4)
    p.add_argument("--num_epochs", type=int, default=100)
    p.add_argument("--log_frequency", type=int, default=250)
    p.add_argument("--save_freq", type=int, default=1)
    p.add_argument("--fast_eval", action="store_true", default=False)
    p.add_argument("--learning_rate", default=1e-5, type=float)
    p.add_argument("--clip_grad_norm", action="store_true", default=False)
    p.add_argument("--leftright_consistency", action="store_true", default=False)
    p.add_argument("--smoothness_weight", type=float, default=1e-3)
    p.add_argument("--consistency_weight", type=float, default=1e-3)
    p.add_argument("--num_steps", type=int, default=-1)
    p.add_argument("--ovs_buffer_size", type=int, default=10)
    p.add_argument("--skip_initial_eval", action="store_true")
    p.add_argument("--ovs_validate_hz", type=int, default=100)
    p.add_argument("--adapt_mode", choices=["NONSTOP", "VS", "ER", "VS+ER", "NONE"])
    p.add_argument("--val_improve_retries", type=int, default=1)
    p.add_argument("--eval_hz", type=int, default=1000)
    p.add_argument("--er_loss_weight", type=float, default=0.05)
    p.add_argument("--train_dataset_path", type=str)
    p.add_argument("--train_dataset_name", type=str)
    p.add_argument("--train_split", type=str)
    p.add_argument("--ood_threshold", type=float, default=15.0)
    p.add_argument("--fcs_ema_weight", type=float, default=0.999)
    self.parser = p

  def parse(self, args=None):
    self.options = self.parser.parse_args(args)
    return self.options


def process_batch(feature_net, stereo_net, left, right, opt, output_cost_volume=False):
  """Reference train.py:19-22.  In eval mode the feature extractor is stateless and batch-independent (bit for bit), so
  both images pass it as one batch (half the launches); in train mode each image keeps its own BatchNorm statistics."""
  if not feature_net.training and left.shape == right.shape:
    both = feature_net(torch.cat([left, right]))
    left_feat, right_feat = both[:left.shape[0]], both[left.shape[0]:]
  else:
    left_feat, right_feat = feature_net(left), feature_net(right)
  return stereo_net(left, left_feat, right_feat, "l", output_cost_volume=output_cost_volume)


def disparity_metrics(pred_disp, gt_disp):
  """-> device tensor [EPE, D1_all_2px, D1_all_3px, D1_all_4px, D1_all_5px] for one batch (no host sync)."""
  nat.require_gpu(pred_disp, gt_disp)
  pred, gt = nat.f32c(pred_disp), nat.f32c(gt_disp)
  n = pred.numel()
  out6 = torch.empty(6, dtype=torch.float32, device=pred.device)
  ws = torch.empty(nat.load().as_eval_metrics_workspace(n), dtype=torch.float32, device=pred.device)
  nat.call("as_eval_metrics", nat.ptr(pred), nat.ptr(gt), n, nat.ptr(out6), nat.ptr(ws), nat.stream())
  return torch.cat([out6[0:1], out6[2:6]]) / out6[1]


def evaluate(feature_net, stereo_net, val_loader, opt):
  """Metrics dict {"EPE", "FCS", "D1_all_{2,3,4,5}px"} exactly as train.py:74-126 defines them."""
  was_training = feature_net.training
  feature_net.eval(); stereo_net.eval()
  n_eval = len(val_loader) // 10 if getattr(opt, "fast_eval", False) else len(val_loader)
  if getattr(opt, "num_steps", -1) > 0:
    n_eval = min(opt.num_steps // val_loader.batch_size, len(val_loader))
  s, k = opt.stereonet_input_scale, opt.stereonet_k
  rows, fcs = [], []
  with torch.no_grad():
    for i, inputs in enumerate(val_loader):
      if i >= n_eval:
        break
      left = inputs["color_l/{}".format(s)].cuda()
      right = inputs["color_r/{}".format(s)].cuda()
      gt = inputs["gt_disp_l/{}".format(s)].cuda()
      outputs = process_batch(feature_net, stereo_net, left, right, opt, output_cost_volume=True)
      rows.append(disparity_metrics(outputs["pred_disp_l/{}".format(s)], gt))
      fcs.append(feature_contrast_mean(outputs["cost_volume_l/{}".format(s + k)]).mean())
    m = torch.stack(rows).mean(dim=0).cpu() if rows else torch.zeros(5)
    f = float(torch.stack(fcs).mean()) if fcs else 0.0
  feature_net.train(was_training); stereo_net.train(was_training)
  return {"EPE": float(m[0]), "FCS": f, "D1_all_2px": float(m[1]), "D1_all_3px": float(m[2]),
          "D1_all_4px": float(m[3]), "D1_all_5px": float(m[4])}


def log_scalars(writer, metrics, losses, examples_per_sec, epoch, step):
  """Console summary (+ scalars to `writer` when one is given; tensorboardX is not a dependency here)."""
  if writer is not None:
    for name in losses:
      writer.add_scalar(name, float(losses[name]), step)
    for name in metrics:
      writer.add_scalar(name, float(metrics[name]), step)
    writer.add_scalar("examples_per_sec", examples_per_sec, step)
  print("\n{}|{} examples/sec={:.3f}".format(epoch, step, examples_per_sec))
  if metrics:
    print("METRICS // " + " | ".join("{}={:.3f}".format(k, float(v)) for k, v in sorted(metrics.items())))
  if losses:
    print("LOSS    // " + " | ".join("{}={:.3f}".format(k, float(v)) for k, v in losses.items()))


def log_images(writer, inputs, outputs, step, skip_prefixes=("cost_volume",)):
  if writer is None:
    return
  for io in (inputs, outputs):
    for name in io:
      if any(p in name for p in skip_prefixes):
        continue
      writer.add_image(name, io[name][0].detach().float().cpu(), step)


def save_models(feature_net, stereo_net, optimizer, log_path, epoch):
  """<log_path>/models/weights_<epoch>/{feature_net,stereo_net,adam}.pth (train.py:129-137).  The state
  dicts are cloned to the CPU first: parameters here are views into a flat arena."""
  folder = os.path.join(log_path, "models", "weights_{}".format(epoch))
  os.makedirs(folder, exist_ok=True)
  for name, net in (("feature_net", feature_net), ("stereo_net", stereo_net)):
    torch.save({k: v.detach().cpu().clone() for k, v in net.state_dict().items()}, os.path.join(folder, name + ".pth"))
  if optimizer is not None:
    torch.save(optimizer.state_dict(), os.path.join(folder, "adam.pth"))
  return folder


def load_models(feature_net, stereo_net, folder, strict=True):
  """adapt.py:203-206."""
  feature_net.load_state_dict(torch.load(os.path.join(folder, "feature_net.pth"), map_location="cpu"), strict=strict)
  stereo_net.load_state_dict(torch.load(os.path.join(folder, "stereo_net.pth"), map_location="cpu"), strict=strict)
