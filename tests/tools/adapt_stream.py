"""Continuous online adaptation on a synthetic stream (BASELINE configs[4], the one-GPU leg): control.AdaptationLoop around
OnlineAdapter for --steps steps, one pair per step (the reference's online setting), mode VS+ER by default: OOD gate on the
FCS EMA, reservoir OVS with validation, experience replay with the Khamis loss.  Prints one JSON summary line (losses and FCS
at checkpoints, state-machine counters, pairs/s) — the same loop tests/test_gpu_stream.py checks against the oracle.
  python tests/tools/adapt_stream.py --steps 1000 --height 375 --width 1242"""
import argparse
import json
import os
import random
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import conftest                                           # noqa: E402,F401
from adaptive_stereo.adaptation import OnlineAdapter      # noqa: E402
from adaptive_stereo.control import AdaptationLoop, State  # noqa: E402
from adaptive_stereo.models.stereo_net import StereoNet, FeatureExtractorNetwork   # noqa: E402
from adaptive_stereo.utils import synthetic as syn        # noqa: E402


def make_pool(n, H, W, dev, seed=100):
  """n distinct pairs with known integer disparities (the ER ground truth is the constant map d0)."""
  pool = []
  disps = (3.0, 5.0, 8.0, 11.0, 14.0, 17.0)
  for i in range(n):
    d0 = disps[i % len(disps)]
    l, r = syn.stereo_pair(1, H, W, seed=seed + i, disparities=(d0,))
    pool.append((l.to(dev), r.to(dev), torch.full((1, 1, H, W), d0, device=dev)))
  return pool


def build(k, gain, dev, maxdisp=192):
  fnet, snet = FeatureExtractorNetwork(k), StereoNet(k, 1, 0, maxdisp=maxdisp)
  fnet.load_state_dict(syn.synthetic_state_dict(fnet.state_dict(), seed=123))
  snet.load_state_dict(syn.synthetic_state_dict(snet.state_dict(), seed=123, logit_gain=gain))
  return fnet.to(dev), snet.to(dev)


def run(steps, H, W, mode="VS+ER", k=4, gain=20.0, pool_size=32, lr=5e-5, dev="cuda:0", on_step=None):
  random.seed(123); torch.manual_seed(123)               # adapt.py:28-31
  fnet, snet = build(k, gain, dev)
  adapter = OnlineAdapter(fnet, snet, H, W, lr=lr)
  pool = make_pool(pool_size, H, W, dev)
  # the OOD threshold is set after the first step, 2 % above the FCS EMA it produced: the stream then reads as "novel", fills
  # the OVS (no updates meanwhile, adapt.py:381), and from there the reservoir's coin and the EMA's drift decide
  loop = AdaptationLoop(adapter, mode=mode, ovs_buffer_size=8, ovs_validate_hz=50, val_improve_retries=2,
                        ood_threshold=-1e30, er_loss_weight=0.05)
  trace = []
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for i in range(steps):
    l, r, _ = pool[i % pool_size]
    rl, rr, rgt = pool[(7 * i + 3) % pool_size]
    if on_step is not None:
      on_step(i, loop, l, r)
    res = loop.process(l, r, i, replay=(rl, rr, rgt))
    trace.append((res["loss"], res["fcs"], res["fcs_smoothed"].clone(), res["state"], res["updated"], res["added_to_ovs"]))
    if i == 0:
      loop.ood_threshold = float(res["fcs_smoothed"]) * 1.02
  torch.cuda.synchronize(); dt = time.perf_counter() - t0
  return loop, adapter, fnet, snet, trace, dt


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--steps", type=int, default=1000)
  ap.add_argument("--height", type=int, default=375)
  ap.add_argument("--width", type=int, default=1242)
  ap.add_argument("--mode", default="VS+ER")
  args = ap.parse_args()
  loop, adapter, fnet, snet, trace, dt = run(args.steps, args.height, args.width, args.mode)
  losses = torch.stack([t[0].float() for t in trace]).cpu()
  fcs = torch.stack([t[1].float() for t in trace]).cpu()
  ema = torch.stack([t[2].float() for t in trace]).cpu()
  marks = [m for m in (1, 10, 100, 500, 1000) if m <= args.steps]
  finite = bool(torch.isfinite(losses).all() and torch.isfinite(fcs).all() and torch.isfinite(adapter.arena.params).all())
  bufs_finite = all(bool(torch.isfinite(b.float()).all()) for net in (fnet, snet) for b in net.buffers())
  print(json.dumps({
      "steps": args.steps, "size": [args.height, args.width], "mode": args.mode, "seconds": round(dt, 3),
      "pairs_per_s": round(args.steps / dt, 2), "gradient_updates": loop.gradient_updates,
      "added_to_ovs": sum(1 for t in trace if t[5]), "steps_in_done_state": sum(1 for t in trace if t[3] == State.DONE),
      "ovs_size": loop.state_machine.ovs_buffer_size(), "all_finite": finite, "buffers_finite": bufs_finite,
      "loss_at": {str(m): round(float(losses[m - 1]), 6) for m in marks},
      "fcs_at": {str(m): round(float(fcs[m - 1]), 5) for m in marks},
      "fcs_ema_at": {str(m): round(float(ema[m - 1]), 5) for m in marks},
      "loss_mean_first_50": round(float(losses[:50].mean()), 6), "loss_mean_last_50": round(float(losses[-50:].mean()), 6)}))


if __name__ == "__main__":
  main()
