#!/usr/bin/env python3
"""Benchmark of the hot path named by BASELINE.json: stereo pairs/s at KITTI 1242x375,
maxdisp 192 (k=4, input_scale=0), forward + one online-adaptation step, synthetic inputs,
1..8 MI355X (one process per GPU, RCCL gradient all-reduce).

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

A "step" = feature nets on both images + StereoNet forward + LinearWarping + monodepth loss +
FCS + backward + clip_grad_norm_(stereo_net, 1.0) + Adam(lr 5e-5) on a batch of `--batch` pairs
per GPU (reference: adapt.py:304-396 NONSTOP / evaluation/stereonet_timing.py:44-72).
Prints ONE JSON line (rank 0).  `value` is the adaptation-step rate; the forward-only rate
(eval, no_grad, evaluation/stereonet_timing.py:22-41) is reported next to it.
"""
import argparse
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (REPO, os.path.join(REPO, "adaptive-stereo-icra-2021_amd")):
  if p not in sys.path:
    sys.path.insert(0, p)

import torch
import torch.distributed as dist

FP32_MFMA_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"


def parse():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=20)
  ap.add_argument("--warmup", type=int, default=5)
  ap.add_argument("--batch", type=int, default=4, help="stereo pairs per GPU per step (32/8 in BASELINE configs[3])")
  ap.add_argument("--height", type=int, default=375)
  ap.add_argument("--width", type=int, default=1242)
  ap.add_argument("--k", type=int, default=4)
  ap.add_argument("--maxdisp", type=int, default=192)
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--online", action="store_true", help=argparse.SUPPRESS)        # (the default now; accepted for older scripts)
  ap.add_argument("--no-online", action="store_true", help="skip \"online_batch1\" (the same step and forward at ONE pair per "
                  "step, the reference's own online setting: adapt_*.sh --batch_size 1), measured by default AFTER the "
                  "profiled region; tracing scripts pass this so that a kernel trace holds launches of one size")
  ap.add_argument("--no-legs", action="store_true", help="skip the side legs \"batch16\" (BASELINE.md 4: a saturating per-GPU "
                  "batch), \"sceneflow_fwd\" (960x540 forward only, evaluation/stereonet_timing.py:22-41) and \"sustained\" (the captured "
                  "step replayed for >= 2 s)")
  ap.add_argument("--no-dp-overhead", action="store_true", help="skip \"dp_path_overhead_ms\" (N=1 only: the data-parallel "
                  "step in a one-rank RCCL group minus the plain step, the one scaling-loss term one GPU can measure)")
  ap.add_argument("--one-stream", action="store_true", help="the two feature extractions of a pair back to back on one "
                  "stream instead of side by side on two (for per-kernel profiles: rocprofv3 serialises queues)")
  ap.add_argument("--sync-bn", action="store_true", help="N>1: train-mode BatchNorm over the batches of all ranks (the "
                  "reference's whole-batch semantics; 34 small collectives per step, eager launches) instead of "
                  "per-replica statistics")
  ap.add_argument("--no-graph", action="store_true", help="launch the ~130 kernels of a step eagerly instead of "
                                                          "replaying a captured hipGraph (single GPU only)")
  return ap.parse_args()


def log(msg):
  if int(os.environ.get("RANK", "0")) == 0:
    print("[bench %7.1fs] %s" % (time.perf_counter() - T0, msg), file=sys.stderr, flush=True)


T0 = time.perf_counter()


def csrc_digest():
  """sha256 over the kernel sources (csrc/*.hip, *.h, sorted by name; code lines only) — recorded in a PMC summary when its counters are taken
  (tests/tools/pmc_summarize.py) and compared here: HBM-counter figures of other kernels than the ones this run timed are not
  reported."""
  import glob
  import hashlib
  h = hashlib.sha256()
  d = os.path.join(REPO, "adaptive-stereo-icra-2021_amd", "csrc")
  for path in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
    h.update(os.path.basename(path).encode()); h.update(b"\0")
    for line in open(path, "rb").read().split(b"\n"):
      t = line.strip()
      if t and not t.startswith(b"//"):          # (whole-line comments and blank lines do not change a kernel)
        h.update(t); h.update(b"\n")
  return h.hexdigest()[:16]


def model_flops(H, W, k, maxdisp):
  """Algorithmic forward FLOPs per stereo pair (SURVEY 8 size table: 2 x output voxels x Cin x Cout x taps per
  convolution; both feature towers, the four 3-D layers + conv3d_alone, the refinement): 63.4 G at KITTI k=4."""
  def down(n):
    return (n - 1) // 2 + 1
  feat, h, w, cin = 0.0, H, W, 3
  for _ in range(k):
    h, w = down(h), down(w)
    feat += 2.0 * h * w * 25 * cin * 32
    cin = 32
  feat += 7 * 2.0 * h * w * 9 * 32 * 32
  D = (maxdisp + 1) // 2 ** k
  agg = 4 * 2.0 * D * h * w * 27 * 32 * 32 + 2.0 * D * h * w * 27 * 32
  refine = 2.0 * H * W * 9 * 4 * 32 + 6 * 2.0 * H * W * 9 * 32 * 32 + 2.0 * H * W * 9 * 32
  return {"feature_towers": 2 * feat, "aggregation": agg, "refinement": refine, "forward": 2 * feat + agg + refine}


def timed(fn, steps, world):
  """barrier + synchronize, K calls, synchronize + barrier; MAX over ranks (seconds)."""
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    fn()
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  dt = time.perf_counter() - t0
  if world > 1:
    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t)
  return dt


def cpu_share():
  """(threads for the CPU baseline, how that number was arrived at).  BASELINE.md 4 says "all host cores"; what a one-GPU
  box may use is bounded by, in this order: the scheduler affinity of this process, the cgroup CPU quota (cpu.max), and the
  pool's stated CPU share of 16 threads per GPU (the host is shared by the boxes of its 8 GPUs; its 256 hardware threads
  are visible to every one of them).  AS_CPU_THREADS overrides the last of the three."""
  try:
    aff = len(os.sched_getaffinity(0))
  except AttributeError:
    aff = os.cpu_count() or 1
  quota = None
  try:
    q, per = open("/sys/fs/cgroup/cpu.max").read().split()
    if q != "max":
      quota = max(1, int(float(q) / float(per) + 0.5))
  except (OSError, ValueError):
    pass
  share = int(os.environ.get("AS_CPU_THREADS", "16"))
  n = max(1, min(aff, quota if quota is not None else aff, share))
  why = "min(affinity %d, cgroup cpu.max %s, pool share per GPU %d)" % (aff, quota if quota is not None else "unlimited", share)
  return n, why


def host_threads():
  return cpu_share()[0]


def cpu_model_string():
  try:
    for line in open("/proc/cpuinfo"):
      if line.lower().startswith("model name"):
        return line.split(":", 1)[1].strip()
  except OSError:
    pass
  import platform
  return platform.processor() or platform.machine()


def cpu_baseline(args, fsd, ssd, budget_s=10.0):
  """The oracle (a port of the reference's CPU path) on this host's cores: all threads of the box's CPU share
  AND one thread; a bounded sample per leg — 3 warm-ups (1 for the one-thread legs, whose single repetition already takes
  seconds) then as many repetitions as fit in ~budget_s seconds (at least one, at most 10); medians reported."""
  from adaptive_stereo.utils import synthetic as syn
  from oracle import stereo_oracle as orc
  left, right = syn.stereo_pair(1, args.height, args.width, seed=1)

  def sample(fn, warmups, budget):
    for _ in range(warmups):
      fn()
    times, t_begin = [], time.perf_counter()
    while True:
      t0 = time.perf_counter(); fn(); times.append(time.perf_counter() - t0)
      if time.perf_counter() - t_begin >= budget or len(times) >= 10:
        times.sort()
        return times[len(times) // 2], len(times)

  legs = {}
  for threads, warmups, budget in ((host_threads(), 3, budget_s), (1, 1, budget_s / 2)):
    torch.set_num_threads(threads)
    fp, sp = orc.make_params(fsd, True), orc.make_params(ssd, True)
    state = {}
    t_adapt, n_adapt = sample(lambda: orc.adapt_step(fp, sp, state, left, right, args.k, 0, args.maxdisp), warmups, budget)
    t_fwd, n_fwd = sample(lambda: orc.forward_only(fsd, ssd, left, right, args.k, 0, args.maxdisp), warmups, budget)
    log("cpu baseline, %d thread(s): adapt %.3f s/step (%d reps), forward %.3f s (%d reps)" % (threads, t_adapt, n_adapt,
                                                                                             t_fwd, n_fwd))
    legs[threads] = (t_adapt, n_adapt, t_fwd, n_fwd, warmups)
  torch.set_num_threads(host_threads())
  # parity in the same run: the GPU forward of this pair against the CPU oracle's
  from adaptive_stereo.models.stereo_net import StereoNet, FeatureExtractorNetwork
  ref_out, _ = orc.forward_only(fsd, ssd, left, right, args.k, 0, args.maxdisp)
  fnet, snet = FeatureExtractorNetwork(args.k), StereoNet(args.k, 1, 0, maxdisp=args.maxdisp)
  fnet.load_state_dict(fsd); snet.load_state_dict(ssd)
  fnet, snet = fnet.cuda().eval(), snet.cuda().eval()
  with torch.no_grad():
    ld, rd = left.cuda(), right.cuda()
    out = snet(ld, fnet(ld), fnet(rd), "l", output_cost_volume=True)
  key = "cost_volume_l/%d" % args.k
  epe = float((out["pred_disp_l/0"].cpu() - ref_out["pred_disp_l/0"]).abs().mean())
  srt = torch.sort(ref_out[key], dim=1, descending=True)[0]
  gap = srt[:, 0] - srt[:, 1]
  am, ref_am = out[key]._as_argmax.cpu().long(), torch.argmax(ref_out[key], dim=1)
  bad = am != ref_am
  parity = {"disparity_epe_vs_cpu": epe, "epe_bar": 1e-3, "argmax_mismatches": int(bad.sum()),
            "argmax_pixels": int(bad.numel()), "argmax_equal": not bool(bad.any()),
            "ref_top2_gap_max_at_mismatch": float(gap[bad].max()) if bool(bad.any()) else 0.0,
            "ref_top2_gap_min": float(gap.min()),
            "logit_max_abs_err": float((out[key].cpu() - ref_out[key]).abs().max())}
  log("parity: EPE %.2e, arg-max mismatches %d of %d pixels (reference's smallest top-2 gap %.2e, largest at a mismatch "
      "%.2e)" % (epe, parity["argmax_mismatches"], parity["argmax_pixels"], parity["ref_top2_gap_min"],
                 parity["ref_top2_gap_max_at_mismatch"]))
  many = host_threads()
  t_adapt, n_adapt, t_fwd, n_fwd, wu = legs[many]
  t1_adapt, n1_adapt, t1_fwd, n1_fwd, wu1 = legs[1]
  return {"value": round(1.0 / t_adapt, 4), "unit": "stereo pairs/s (fwd+adapt-step)", "cores": many,
          "kind": "port", "fwd_value": round(1.0 / t_fwd, 4),
          "one_thread": {"value": round(1.0 / t1_adapt, 4), "fwd_value": round(1.0 / t1_fwd, 4), "cores": 1},
          "cores_source": cpu_share()[1],
          "cpu_model": cpu_model_string(), "host_cores_visible": os.cpu_count(), "torch": torch.__version__,
          "parity": parity,
          "sample": "oracle/stereo_oracle.py (PyTorch CPU fp32), batch 1 at %dx%d, medians: %d threads: %d warm-ups + %d "
                    "adapt steps, %d warm-ups + %d forwards; 1 thread: %d warm-up + %d adapt step(s), %d warm-up + %d "
                    "forward(s)" % (args.width, args.height, many, wu, n_adapt, wu, n_fwd, wu1, n1_adapt, wu1, n1_fwd)}


def dp_path_overhead(args, fsd, ssd, left, right, dev, use_graph, plain_ms):
  """N = 1 only: the data-parallel step (local loss sums, flush, ONE RCCL all-reduce of [gradients | 4 scalars] on its own
  stream, 1/N_total, clip + Adam; two hipGraphs with the collective between them) in a ONE-rank "nccl" process group, minus
  the plain single-GPU step of the same batch measured in this run.  It is the scaling-loss term one GPU can measure:
  everything the data-parallel path adds except the wire time of the collective itself."""
  import socket
  from adaptive_stereo.adaptation import OnlineAdapter
  from adaptive_stereo.models.stereo_net import StereoNet, FeatureExtractorNetwork
  try:
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    try:
      def leg(native):
        f1, s1 = FeatureExtractorNetwork(args.k), StereoNet(args.k, 1, 0, maxdisp=args.maxdisp)
        f1.load_state_dict(fsd); s1.load_state_dict(ssd)
        a1 = OnlineAdapter(f1.to(dev), s1.to(dev), args.height, args.width, lr=5e-5, clip_grad_norm=True,
                           overlap_features=not args.one_stream, force_data_parallel=True, native_collectives=native)
        try:
          for _ in range(max(2, args.warmup // 2)):
            a1.step(left, right)
          l1, r1 = left, right
          if use_graph:
            a1.capture(left, right, warmup=1)
            l1, r1 = a1.graph_inputs(); l1.copy_(left); r1.copy_(right)
            a1.step(l1, r1)
          torch.cuda.synchronize()
          t = timed(lambda: a1.step(l1, r1), args.steps, 1)
          return 1e3 * t / args.steps, a1.graph_count(), a1.comm is not None, a1.arena.grads_and_scalars.numel()
        finally:
          a1.close()                       # the communicator (and the graph holding its nodes) goes before the process group
      dp_ms, graphs, native, floats = leg(True)
      c10d_ms, c10d_graphs, _, _ = leg(False)
      log("data-parallel path in a one-rank RCCL group: %.3f ms/step in %d graph(s) (torch.distributed collectives: %.3f in %d; "
          "plain %.3f)" % (dp_ms, graphs, c10d_ms, c10d_graphs, plain_ms))
      return {"value": round(dp_ms - plain_ms, 3), "dp_one_rank_ms_per_step": round(dp_ms, 3), "plain_ms_per_step": round(plain_ms, 3),
              "graphs": graphs, "native_rccl_communicator": native,
              "torch_distributed_collectives": {"value": round(c10d_ms - plain_ms, 3), "ms_per_step": round(c10d_ms, 3), "graphs": c10d_graphs},
              "note": "one-rank RCCL group on this GPU: all-reduce of %d floats per step; everything the data-parallel path adds "
                      "except the wire time of the collective" % floats}
    finally:
      dist.destroy_process_group()
  except Exception as e:                      # a measurement leg must never take the headline line down with it
    log("dp_path_overhead failed: %r" % (e,))
    return {"value": None, "error": repr(e)[:300]}


_JSON_OUT = None


def claim_stdout():
  """The contract is ONE JSON line on stdout.  Libraries underneath write there too (librccl prints a five-line version banner
  when a communicator is created — torch.distributed's or this package's): the real stdout is kept aside for the JSON line and
  file descriptor 1 points at stderr for everything else, C and Python alike."""
  global _JSON_OUT
  sys.stdout.flush()
  _JSON_OUT = os.fdopen(os.dup(1), "w")
  os.dup2(2, 1)


def emit_json(obj):
  sys.stdout.flush()
  _JSON_OUT.write(json.dumps(obj) + "\n")
  _JSON_OUT.flush()


def main():
  args = parse()
  claim_stdout()
  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  if os.environ.get("AS_BENCH_SINGLE_DEVICE"):      # rehearsal of the N>1 code path on a one-GPU box (with gloo)
    local_rank = 0
  if world > 1:
    torch.cuda.set_device(local_rank)
    backend = os.environ.get("AS_BENCH_BACKEND", "nccl")   # nccl = RCCL over xGMI; gloo only for the rehearsal above
    if backend == "nccl":
      dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
      dist.init_process_group(backend=backend)
  assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU product path)"
  dev = torch.device("cuda", local_rank)

  from adaptive_stereo import _native as nat
  from adaptive_stereo.adaptation import OnlineAdapter
  from adaptive_stereo.models.stereo_net import StereoNet, FeatureExtractorNetwork
  from adaptive_stereo.utils import synthetic as syn

  fnet, snet = FeatureExtractorNetwork(args.k), StereoNet(args.k, 1, 0, maxdisp=args.maxdisp)
  fsd = syn.synthetic_state_dict(fnet.state_dict(), seed=123)
  ssd = syn.synthetic_state_dict(snet.state_dict(), seed=123)
  fnet.load_state_dict(fsd); snet.load_state_dict(ssd)
  fnet, snet = fnet.to(dev), snet.to(dev)
  B = args.batch
  left, right = syn.stereo_pair(B, args.height, args.width, seed=1 + rank)
  left, right = left.to(dev), right.to(dev)
  adapter = OnlineAdapter(fnet, snet, args.height, args.width, lr=5e-5, clip_grad_norm=True, sync_bn=args.sync_bn,
                          overlap_features=not args.one_stream)

  lib = nat.load()
  # cross-replica BatchNorm puts collectives inside forward/backward: capturable only with the library's own communicator
  use_graph = not args.no_graph and (adapter.bn_sync is None or adapter.comm is not None)
  log("setup done: %d pairs/GPU at %dx%d, world %d, %s" % (B, args.width, args.height, world,
                                                           "hipGraph replay" if use_graph else "eager launches"))
  # ---- forward + adaptation step ---------------------------------------------------------
  for i in range(args.warmup):
    adapter.step(left, right)
    torch.cuda.synchronize()
    log("warm-up adapt step %d done" % i)
  step_l, step_r = left, right
  if use_graph:
    # Nothing below has run on more than one GPU before the driver's scaling runs: a capture that fails on ANY rank must not
    # cost the measurement.  Ladder, agreed on by all ranks after every rung (a c10d MIN, outside any capture): one graph with
    # the library's own RCCL communicator -> two graphs around torch.distributed's all-reduce -> eager launches.
    def all_ranks(ok):
      if world == 1:
        return ok
      flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
      dist.all_reduce(flag, op=dist.ReduceOp.MIN)
      return int(flag) == 1
    def try_capture(ad):
      try:
        if os.environ.get("AS_BENCH_TEST_CAPTURE_FAILS"):       # rehearsal of the ladder itself
          raise RuntimeError("AS_BENCH_TEST_CAPTURE_FAILS is set")
        ad.capture(left, right, warmup=1)
        return True
      except Exception as e:               # noqa: BLE001 — whatever the runtime throws, the ladder goes on
        log("capture failed on rank %d: %r" % (rank, e))
        return False
    captured = all_ranks(try_capture(adapter))
    if not captured and adapter.dp and adapter.comm is not None:
      log("falling back to torch.distributed collectives between two graphs")
      torch.cuda.synchronize()
      adapter.close()
      adapter = OnlineAdapter(fnet, snet, args.height, args.width, lr=5e-5, clip_grad_norm=True, sync_bn=args.sync_bn,
                              overlap_features=not args.one_stream, native_collectives=False)
      adapter.step(left, right); torch.cuda.synchronize()
      captured = adapter.bn_sync is None and all_ranks(try_capture(adapter))
    if not captured:
      log("falling back to eager launches")
      torch.cuda.synchronize()
      if adapter.graph_count():
        adapter._graph = None
      use_graph = False
  if use_graph:
    # the pairs of the timed region live in the graph's own input buffers (resident in HBM, as the contract asks):
    # a producer decodes into them (datasets.prefetch), so a replay does not begin with two device copies
    step_l, step_r = adapter.graph_inputs()
    step_l.copy_(left); step_r.copy_(right)
    # untimed replays of the captured step (the W warm-up steps above ran eagerly, before the capture): the timed region then
    # starts at the clocks and cache state a stream of steps runs at — the 20-step region lasts 0.14 s, and its first replays
    # right behind a capture measured 1-2 % slower than the `sustained` leg's
    for _ in range(max(3, args.warmup)):
      adapter.step(step_l, step_r)
    torch.cuda.synchronize()
    log("step captured into a hipGraph")
  t_adapt = timed(lambda: adapter.step(step_l, step_r), args.steps, world)

  # Roofline leg: per-launch durations of the dominant kernels from HIP events on the launch stream.
  # Events cannot bracket nodes inside a graph replay, so with --graph the same K steps are run once
  # more eagerly, right after the timed region, with the events armed (same kernels, same shapes).
  n_graphs = adapter.graph_count()
  if use_graph:
    adapter._graph = None
  lib.as_prof_reset(); lib.as_prof_enable(1)
  t_adapt_eager = timed(lambda: adapter.step(left, right), args.steps, world)
  lib.as_prof_enable(0)
  if not use_graph:
    t_adapt = min(t_adapt, t_adapt_eager)
  prof = []
  for kid in range(28):             # AS_PROF_IDS (csrc/as_common.h)
    n, ms, fl = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
    nat.call("as_prof_read", kid, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl))
    prof.append((n.value, ms.value, fl.value))

  log("timed adapt: %.2f ms/step" % (1e3 * t_adapt / args.steps))
  # ---- forward only --------------------------------------------------------------------------
  for _ in range(max(1, args.warmup // 2)):
    adapter.infer(left, right)
  inf_l, inf_r = left, right
  if not args.no_graph:
    try:                                   # (no collective inside inference: a rank that cannot capture just stays eager)
      adapter.capture_infer(left, right)
      inf_l, inf_r = adapter.infer_inputs()
      inf_l.copy_(left); inf_r.copy_(right)
      adapter.infer(inf_l, inf_r)
    except Exception as e:                 # noqa: BLE001
      log("inference capture failed on rank %d: %r — eager inference" % (rank, e))
      adapter._infer_graph = None
      inf_l, inf_r = left, right
  torch.cuda.synchronize()
  log("warm-up forward done")
  t_fwd = timed(lambda: adapter.infer(inf_l, inf_r), args.steps, world)
  log("timed forward: %.2f ms/step" % (1e3 * t_fwd / args.steps))

  if rank != 0:
    if world > 1:
      adapter.close()
      dist.destroy_process_group()
    return

  pairs = world * B * args.steps
  # Per-kernel entries: algorithmic FLOPs per launch (direct form: 2 * voxels * 32 * 32 * taps) / mean launch duration from
  # HIP events recorded on the launch stream.  Which kernel is the dominant one is decided below, by time per step.
  def entry(i, name):
    n, ms, fl = prof[i]
    if n == 0 or ms <= 0:
      return None
    return {"kernel": name, "achieved": round(fl / (ms * 1e-3) / 1e12, 3), "launches": n,
            "avg_launch_us": round(1e3 * ms / n, 2), "flops_per_launch": fl / n}
  # the LDS convolution family: id 2 = forward flavours, id 6 = the data gradient that also carries stage 1 of the
  # following BatchNorm backward (one more tensor read and ~100 vector instructions per tile that the FLOP count
  # does not credit).  The roofline line is the whole family, as one kernel.
  fam = [prof[2], prof[6]]
  n_f, ms_f, fl_f = (sum(t[i] for t in fam) for i in range(3))
  prof_family = (n_f, ms_f, fl_f)
  def entry_family():
    if n_f == 0 or ms_f <= 0:
      return None
    return {"kernel": "conv32_lds_kernel", "achieved": round(fl_f / (ms_f * 1e-3) / 1e12, 3), "launches": n_f,
            "avg_launch_us": round(1e3 * ms_f / n_f, 2), "flops_per_launch": fl_f / n_f}
  fam_entry = entry_family()
  # The dominant kernel is the one a step spends most time in.  Candidates: the full-resolution layers' kernels — by minimal
  # filtering (ids 24-26: data gradient, weight gradient, forward; conv32_wino*.hip), in their direct form (22, 23) — and the
  # conv32_lds family.  For a minimal-filtering kernel `achieved` still counts the ALGORITHMIC FLOPs of the layer (SURVEY 8d:
  # 2 x voxels x 32 x 32 x 9, what the direct form executes), `executed` what the matrix pipe actually does (4/9 of them).
  NAMES = {
      24: ("conv32_wino_dgrad_kernel<L> (dilation 1, 2, 4) / conv32_wino_kernel<2, 3> (dilation 8)", "conv32_wino_dgrad_kernel<0>", 5,
           "data gradient of a full-resolution layer by minimal filtering F(2x2,3x3): stage 3 of the BatchNorm backward on the way "
           "in (g_z written once), skip connection from raw g_a rows kept in LDS, next BatchNorm's sums"),
      25: ("conv32_wino_wgrad_kernel<L>", "conv32_wino_wgrad_kernel<0>", 2,
           "weight / bias gradient of a full-resolution layer by minimal filtering F(3x3,2x2) from x and g_z (LDS-DMA rows)"),
      26: ("conv32_wino_kernel<0|1, L>", "conv32_wino_kernel<1, 0>", 4,
           "full-resolution training forward by minimal filtering F(2x2,3x3): previous BatchNorm + LeakyReLU + skip applied on the "
           "way in, by-product written back, raw output + moments"),
      27: ("conv32_wino_bwd_kernel<L>", "conv32_wino_bwd_kernel<0>", 5,
           "full-resolution layer backward in ONE launch by minimal filtering: stage 3 of the BatchNorm backward on the way in, data "
           "gradient F(2x2,3x3) + skip, weight gradient F(3x3,2x2) from the same g_z tiles (g_z never reaches HBM), next BatchNorm's sums"),
      22: ("conv32_bwd_fused_kernel", "conv32_bwd_fused_kernel", 5,
           "full-resolution layer backward in one launch, direct form: BatchNorm-backward apply, data gradient + skip, weight "
           "gradient, next BatchNorm's sums"),
      23: ("conv32_act_kernel", "conv32_act_kernel<true>", 4,
           "full-resolution training forward, direct form: previous BatchNorm + LeakyReLU + skip applied on the way in, by-product "
           "written back, raw output + moments"),
  }
  cands = [(prof[i][1], i) for i in NAMES if prof[i][0] > 0 and prof[i][1] > 0]
  if fam_entry is not None:
    cands.append((ms_f, -1))
  dom_id = max(cands)[1] if cands else None
  dom = None if dom_id is None else (fam_entry if dom_id == -1 else entry(dom_id, NAMES[dom_id][0]))
  roofline = None
  if dom is not None:
    # HBM bytes per launch from the committed PMC passes (profiles/r0N_pmc_by_pairs.json, keyed by pairs per launch;
    # FETCH_SIZE / WRITE_SIZE corrected as MI355X_MICROARCH.md prescribes).  `traffic` is the dominant kernel's; the other
    # full-resolution kernels and the 3-D kernels are in traffic_detail.
    traffic, traffic_detail, traffic_source = None, None, None
    rec = None
    # the newest committed PMC file wins (rocprofv3 --pmc passes cannot run inside this timed process: the counters come
    # from tests/tools/pmc_run.sh runs of the same kernels at 1, 2 and 4 pairs per launch, committed under profiles/) — but
    # only a file whose counters were taken on THESE kernel sources: each summary records the digest of csrc/ it was measured
    # on; one measured on other sources is named, with the reason, and not used
    digest_now = csrc_digest()
    for name in ("r05_pmc_by_pairs.json", "r04_pmc_by_pairs.json", "r03_pmc_by_pairs.json", "r02_pmc_by_pairs.json"):
      by_pairs = os.path.join(REPO, "profiles", name)
      if os.path.exists(by_pairs):
        table = json.load(open(by_pairs))
        keys = sorted(int(k_) for k_ in table if k_.isdigit())
        if keys:
          near = min(keys, key=lambda k_: (abs(k_ - B), -k_))
          measured_on = table[str(near)].get("csrc_digest")
          traffic_source = {"file": "profiles/" + name, "pairs_per_launch_measured": near, "exact": near == B,
                            "csrc_digest_measured_on": measured_on, "csrc_digest_now": digest_now,
                            "git_commit": table.get("git_commit"),
                            "how": "separate rocprofv3 --pmc passes (FETCH_SIZE x2 per the gfx950 correction, WRITE_SIZE), "
                                   "tests/tools/pmc_run.sh; per launch; scaled by pairs when not exact"}
          if measured_on == digest_now:
            rec = table[str(near)]
          else:
            traffic_source["stale"] = ("the counters were taken on other kernel sources (csrc digest %s, now %s): traffic is not "
                                       "reported" % (measured_on, digest_now))
          break
    if rec is not None:
      ks = rec.get("kernels", {})
      def pmc_row(k):
        v = ks.get(k)
        return None if v is None else {"hbm_bytes": v["hbm_bytes_per_launch"], "algorithmic_bytes": v["algorithmic_bytes_per_launch"],
                                       "traffic_over_algorithmic": v["traffic_over_algorithmic"],
                                       "mfma_busy": v["mfma_busy_fraction_of_simd_cycles"]}
      scale = 1.0 if traffic_source is None or traffic_source["exact"] else B / float(traffic_source["pairs_per_launch_measured"])
      if dom_id == -1:
        if "conv32_lds_kernel<0, false>" in ks and "conv32_lds_kernel<3, true>" in ks:
          traffic = int((ks["conv32_lds_kernel<0, false>"]["hbm_bytes_per_launch"] + ks["conv32_lds_kernel<3, true>"]["hbm_bytes_per_launch"]) / 2 * scale)
      elif NAMES[dom_id][1] in ks or (dom_id == 24 and "conv32_wino_kernel<2, 0>" in ks):
        key_ = NAMES[dom_id][1] if NAMES[dom_id][1] in ks else "conv32_wino_kernel<2, 0>"      # (files older than round 4)
        traffic = int(ks[key_]["hbm_bytes_per_launch"] * scale)
      traffic_detail = {"pairs_per_launch": rec.get("pairs_per_launch"),
                        "full_resolution_layers": {NAMES[i][1]: pmc_row(NAMES[i][1]) for i in NAMES if NAMES[i][1] in ks},
                        "forward": pmc_row("conv32_lds_kernel<0, false>"),
                        "dgrad_with_skip_and_bn_sums": pmc_row("conv32_lds_kernel<3, true>"),
                        "cost_aggregation_3d": {k: pmc_row(k) for k in ks if k.startswith(("agg3d", "agg_tail", "conv3d"))}}
    roofline = {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["achieved"], "peak": FP32_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(dom["achieved"] / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                "traffic_source": traffic_source,
                "launches": dom["launches"], "avg_launch_us": dom["avg_launch_us"],
                "flops_per_launch": dom["flops_per_launch"]}
    vox_bytes = float(B) * args.height * args.width * 128.0
    def wino_views(i, e):
      """What a minimal-filtering kernel executes on the matrix pipe and moves through HBM, beside its algorithmic rate."""
      t = e["avg_launch_us"] * 1e-6
      ex = e["flops_per_launch"] * 4.0 / 9.0
      alg_bytes = NAMES[i][2] * vox_bytes
      return {"algorithm": "minimal filtering (Winograd) F(2x2,3x3) / F(3x3,2x2): 16 multiplications per 2x2 tile and channel pair "
                           "where the direct form has 36 — the matrix pipe executes 4/9 of the algorithmic FLOPs",
              "executed_mfma": {"flops_per_launch": ex, "achieved": round(ex / t / 1e12, 2), "frac": round(ex / t / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)},
              "hbm": {"algorithmic_bytes_per_launch": alg_bytes, "achieved": round(alg_bytes / t / 1e9, 1), "peak": 8000.0,
                      "unit": "GB/s", "frac": round(alg_bytes / t / 8e12, 4)}}
    def binding(e, views):
      """A minimal-filtering kernel executes 4/9 of the layer's direct-form multiplications: its arithmetic intensity
      (executed FLOPs / algorithmic bytes = 12.8 FLOP/B for the data gradient) is below the machine balance (157.3 TFLOP/s /
      8 TB/s = 19.7), so what binds is the larger of (algorithmic bytes / HBM peak) and (executed FLOPs / matrix peak), and
      `frac` is the fraction of THAT bound — never the direct-form FLOP count, which the kernel does not execute."""
      f_hbm, f_mfma = views["hbm"]["frac"], views["executed_mfma"]["frac"]
      e["effective_direct_form"] = {"achieved": e["achieved"], "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": round(e["achieved"] / FP32_MFMA_PEAK_TFLOPS, 4),
                                    "note": "algorithmic FLOPs of the layer in its direct form (SURVEY 8d) / time / fp32 matrix peak: a "
                                            "speed-up figure (it may exceed 1), not a distance from a bound"}
      if f_hbm >= f_mfma:
        e.update({"bound": "hbm", "achieved": views["hbm"]["achieved"], "peak": 8000.0, "unit": "GB/s", "frac": f_hbm})
      else:
        e.update({"bound": "mfma", "achieved": views["executed_mfma"]["achieved"], "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                  "frac": f_mfma})
      return e
    if dom_id in (24, 25, 26, 27):
      views = wino_views(dom_id, dom)
      roofline.update(views)
      binding(roofline, views)
      if traffic is not None:
        roofline["hbm_counter_frac"] = round(traffic / (dom["avg_launch_us"] * 1e-6) / 8e12, 4)
      roofline["note"] = ("frac = the binding roofline: max(algorithmic bytes / 8 TB/s, EXECUTED matrix FLOPs / 157.3 TFLOP/s) / time; "
                          "hbm_counter_frac = measured HBM bytes (traffic) / time / 8 TB/s; effective_direct_form = the layer's direct-form "
                          "FLOPs / time / matrix peak, kept for comparison with the direct kernels")
    roofline["traffic_detail"] = traffic_detail
    flav = []
    for i in (27, 24, 25, 26, 22, 23):
      e = entry(i, "%s (%s)" % (NAMES[i][0], NAMES[i][3]))
      if e is not None:
        if i in (24, 25, 26, 27):
          views = wino_views(i, e)
          e.update({k_: v for k_, v in views.items() if k_ != "algorithm"})
          binding(e, views)
        else:
          e["frac"] = round(e["achieved"] / FP32_MFMA_PEAK_TFLOPS, 4)
        flav.append(e)
    for i, nm in ((2, "conv32_lds_kernel<0,false> (training forward: raw output + BatchNorm moments)"),
                  (6, "conv32_lds_kernel<3,true> (data gradient + skip + stage 1 of the next BatchNorm backward)")):
      e = entry(i, nm)
      if e is not None:
        flav.append(e)
    roofline["flavours"] = flav
    roofline["other_mfma_kernels"] = [e for e in (entry(3, "conv32_wgrad_lds_kernel"),
                                                   entry(7, "agg3d_kernel (a3: rolling-window 3-D aggregation layers and their data gradients)"),
                                                   entry(9, "conv3d_wgrad_lds_kernel (a3 weight gradient)"),
                                                   entry(0, "conv32_fwd_kernel<taps> (strided, small 2-D)"),
                                                   entry(1, "conv32_wgrad_kernel<taps>")) if e is not None]
    # the HBM-bound passes of the step, against the 8 TB/s HBM3E peak (algorithmic bytes / HIP-event time)
    def hbm_entry(i, name):
      n, ms, by = prof[i]
      if n == 0 or ms <= 0:
        return None
      return {"kernel": name, "achieved": round(by / (ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
              "frac": round(by / (ms * 1e-3) / 8e12, 4), "launches": n, "avg_launch_us": round(1e3 * ms / n, 2)}
    roofline["hbm_bound_kernels"] = [e for e in (hbm_entry(4, "bn_act_fwd_kernel"),
                                                 hbm_entry(5, "bn_bwd_reduce + finalize + apply (what is left of "
                                                               "them: the 1/16-resolution and cost-volume layers, "
                                                               "launch-latency bound; the full-resolution passes ride "
                                                               "on the matrix-core kernels)")) if e is not None]
    # the HBM-bound star rows of SURVEY 8 (a2, a4, a5+a8, a6, a9, a10), forward and backward: algorithmic bytes (every
    # tensor read once + written once) / HIP-event time of the launches of this very run, against 8 TB/s
    star = [(10, "a2 cost volume fwd"), (11, "a2 cost volume bwd"),
            (8, "a4+a5+a8 conv3d_alone + soft-argmax + arg-max + FCS fwd (one launch, with layer 4's BatchNorm + LeakyReLU)"),
            (20, "a4 conv3d_alone fwd (separate launch)"), (21, "a5+a8 soft-argmax + FCS fwd (separate launch)"),
            (12, "a4+a5 soft-argmax + conv3d_alone bwd (data + weight gradient; one launch + slab reduction since round 5)"), (13, "a5 soft-argmax bwd (separate launch)"),
            (14, "a6/a7 bilinear up-sampling fwd"), (15, "a6/a7 bilinear up-sampling bwd"),
            (16, "a9 LinearWarping fwd"), (17, "a9 LinearWarping bwd"),
            (18, "a9+a10 LinearWarping + monodepth loss + masked sum fwd (image mean + one row-strip pass)"),
            (19, "a9+a10 backward (one row-strip pass + the per-image mean term)")]
    roofline["hbm_rows"] = [dict(e, row=nm.split(" ")[0]) for e, nm in ((hbm_entry(i, nm_), nm_) for i, nm_ in star) if e is not None]

  mf = model_flops(args.height, args.width, args.k, args.maxdisp)
  if roofline is not None:
    # the whole step / the whole forward against the fp32 matrix peak: algorithmic FLOPs of the model (forward as in SURVEY
    # 8's table; a training step = forward + data gradient + weight gradient = 3 x forward) / measured time / peak
    step_tf = 3.0 * mf["forward"] * B * world / (t_adapt / args.steps) / 1e12
    fwd_tf = mf["forward"] * B * world / (t_fwd / args.steps) / 1e12
    roofline["whole_step"] = {"flops_per_pair": 3.0 * mf["forward"], "achieved": round(step_tf / world, 2), "unit": "TFLOP/s per GPU",
                              "frac": round(step_tf / world / FP32_MFMA_PEAK_TFLOPS, 4)}
    roofline["whole_forward"] = {"flops_per_pair": mf["forward"], "achieved": round(fwd_tf / world, 2), "unit": "TFLOP/s per GPU",
                                 "frac": round(fwd_tf / world / FP32_MFMA_PEAK_TFLOPS, 4),
                                 "breakdown_gflop_per_pair": {k_: round(v / 1e9, 3) for k_, v in mf.items()}}
  out = {
    "metric": "stereo pairs/sec (fwd+adapt-step), KITTI 1242x375 D=192",
    "value": round(pairs / t_adapt, 3),
    "unit": "stereo pairs/s",
    "n_gpus": world,
    "steps": args.steps,
    "warmup": args.warmup,
    "ms_per_step": round(1e3 * t_adapt / args.steps, 3),
    "higher_is_better": True,
    "scaling": "weak",
    "vs_baseline": None,
    "dtype": "f32",
    "data": "synthetic",
    "config": {"workload": "KITTI-2015 %dx%d, maxdisp %d, k=%d (Dc=%d, %dx%d cost volume), forward + one online-adapt "
                           "step (train-mode BN, monodepth loss, clip 1.0, Adam lr 5e-5)" % (
                               args.width, args.height, args.maxdisp, args.k, (args.maxdisp + 1) // 2 ** args.k,
                               -(-args.height // 2 ** args.k), -(-args.width // 2 ** args.k)),
               "pairs_per_gpu": B, "global_batch": world * B,
               "parallelism": "dp%d (%s BatchNorm statistics, one flat RCCL gradient all-reduce)" % (
                   world, "cross-replica" if adapter.bn_sync is not None else "per-replica"),
               "kernels": "all hand-written HIP (no MIOpen/rocBLAS on the path)"},
    "launch_mode": ("hipGraph replay of the captured step (%d graph%s)" % (n_graphs, "" if n_graphs == 1 else "s"))
                   if use_graph else "eager",
    "collectives": None if world == 1 else (
        ("RCCL through the library's own communicator" if adapter.comm is not None else "torch.distributed") +
        (", eager" if not use_graph else (", captured in the step's graph" if adapter.comm is not None else ", between two graphs"))),
    "eager_ms_per_step": round(1e3 * t_adapt_eager / args.steps, 3),
    "fwd_pairs_per_s": round(pairs / t_fwd, 3),
    "fwd_ms_per_step": round(1e3 * t_fwd / args.steps, 3),
    "roofline": roofline,
  }
  if world > 1 and adapter.comm is not None:
    # what the communicator's first collectives returned (adaptive_stereo/rccl.py, stage "probe"): an all-reduce of ones = the
    # number of ranks that took part, an all-gather of each rank's HIP device, the library's version
    out["rccl"] = getattr(adapter.comm, "evidence", None)
  if world == 1 and use_graph and not args.no_legs:
    # "sustained": the captured step replayed back to back for >= 2 s (the 20-step region above lasts ~0.15 s)
    adapter.capture(left, right, warmup=1)
    sl, sr = adapter.graph_inputs(); sl.copy_(left); sr.copy_(right)
    adapter.step(sl, sr); torch.cuda.synchronize()
    n_sus, t_sus = 0, 0.0
    while t_sus < 2.0:
      t_sus += timed(lambda: adapter.step(sl, sr), 50, 1); n_sus += 50
    out["sustained"] = {"ms_per_step": round(1e3 * t_sus / n_sus, 3), "pairs_per_s": round(B * n_sus / t_sus, 3), "steps": n_sus,
                        "seconds": round(t_sus, 3), "note": "the same captured step, replayed back to back in blocks of 50"}
    log("sustained: %.3f ms/step over %d steps" % (1e3 * t_sus / n_sus, n_sus))
    adapter._graph = None

  def side_leg(Bl, Hl, Wl, forward_only, seed):
    """A fresh pair of networks and an adapter of their own at another batch / image size: (seconds per step, per forward)."""
    f1, s1 = FeatureExtractorNetwork(args.k), StereoNet(args.k, 1, 0, maxdisp=args.maxdisp)
    f1.load_state_dict(fsd); s1.load_state_dict(ssd)
    a1 = OnlineAdapter(f1.to(dev), s1.to(dev), Hl, Wl, lr=5e-5, clip_grad_norm=True, overlap_features=not args.one_stream)
    l1, r1 = syn.stereo_pair(Bl, Hl, Wl, seed=seed)
    l1, r1 = l1.to(dev), r1.to(dev)
    t_step = None
    if not forward_only:
      l1s, r1s = l1, r1
      for _ in range(max(2, args.warmup // 2)):
        a1.step(l1, r1)
      if use_graph:
        a1.capture(l1, r1, warmup=1)
        g1l, g1r = a1.graph_inputs(); g1l.copy_(l1); g1r.copy_(r1); l1s, r1s = g1l, g1r
        a1.step(l1s, r1s)
      torch.cuda.synchronize()
      t_step = timed(lambda: a1.step(l1s, r1s), args.steps, 1) / args.steps
    for _ in range(2):
      a1.infer(l1, r1)
    li, ri = l1, r1
    if use_graph:
      a1.capture_infer(l1, r1)
      li, ri = a1.infer_inputs(); li.copy_(l1); ri.copy_(r1)
      a1.infer(li, ri)
    torch.cuda.synchronize()
    t_inf = timed(lambda: a1.infer(li, ri), args.steps, 1) / args.steps
    del a1, f1, s1
    return t_step, t_inf

  if world == 1 and not args.no_legs:
    try:
      if B != 16:
        t16, t16f = side_leg(16, args.height, args.width, False, 11)
        out["batch16"] = {"pairs_per_s": round(16 / t16, 3), "ms_per_step": round(1e3 * t16, 3), "fwd_pairs_per_s": round(16 / t16f, 3),
                          "fwd_ms_per_step": round(1e3 * t16f, 3),
                          "note": "16 pairs per GPU and step (BASELINE.md 4: a saturating per-GPU batch), same kernels and graphs"}
        log("batch 16: %.2f ms/step, forward %.2f ms" % (1e3 * t16, 1e3 * t16f))
      _, tsf = side_leg(B, 540, 960, True, 13)
      out["sceneflow_fwd"] = {"pairs_per_s": round(B / tsf, 3), "ms_per_forward": round(1e3 * tsf, 3), "pairs_per_gpu": B,
                              "workload": "SceneFlow Flying 960x540, maxdisp %d, k=%d, forward only (eval, no_grad; BASELINE.json configs[1], "
                                          "evaluation/stereonet_timing.py:22-41)" % (args.maxdisp, args.k)}
      log("SceneFlow 960x540 forward: %.2f ms for %d pairs" % (1e3 * tsf, B))
    except Exception as e:                 # noqa: BLE001 — a side leg must never take the headline line down with it
      log("side legs failed: %r" % (e,))
      out["side_legs_error"] = repr(e)[:300]
  if world == 1 and B != 1 and not args.no_online:
    # The reference adapts online, one pair per step (experiments/adaptation/adapt_*.sh: --batch_size 1): the same
    # step at batch 1 next to the headline configuration (fresh networks, its own captured graphs).
    f1, s1 = FeatureExtractorNetwork(args.k), StereoNet(args.k, 1, 0, maxdisp=args.maxdisp)
    f1.load_state_dict(fsd); s1.load_state_dict(ssd)
    a1 = OnlineAdapter(f1.to(dev), s1.to(dev), args.height, args.width, lr=5e-5, clip_grad_norm=True,
                       overlap_features=not args.one_stream)
    l1, r1 = left[:1].contiguous(), right[:1].contiguous()
    l1s, r1s = l1, r1
    for _ in range(max(2, args.warmup // 2)):
      a1.step(l1, r1)
    if use_graph:
      a1.capture(l1, r1, warmup=1)
      g1l, g1r = a1.graph_inputs(); g1l.copy_(l1); g1r.copy_(r1); l1s, r1s = g1l, g1r
      a1.step(l1s, r1s)
    torch.cuda.synchronize()
    t1 = timed(lambda: a1.step(l1s, r1s), args.steps, 1)
    for _ in range(2):
      a1.infer(l1, r1)
    if use_graph:
      a1.capture_infer(l1, r1)
      a1.infer(l1, r1)
    torch.cuda.synchronize()
    t1f = timed(lambda: a1.infer(l1, r1), args.steps, 1)
    out["online_batch1"] = {"pairs_per_s": round(args.steps / t1, 3), "ms_per_step": round(1e3 * t1 / args.steps, 3),
                            "fwd_pairs_per_s": round(args.steps / t1f, 3), "fwd_ms_per_step": round(1e3 * t1f / args.steps, 3),
                            "note": "one pair per step (the reference's online setting), same kernels and graphs"}
    log("online (batch 1): %.2f ms/step, forward %.2f ms" % (1e3 * t1 / args.steps, 1e3 * t1f / args.steps))
    del a1, f1, s1
  if world == 1 and not args.no_dp_overhead:
    out["dp_path_overhead_ms"] = dp_path_overhead(args, fsd, ssd, left, right, dev, use_graph, 1e3 * t_adapt / args.steps)
  if world == 1 and not args.no_cpu_baseline:
    out["cpu_baseline"] = cpu_baseline(args, fsd, ssd)
    log("cpu baseline done")
  emit_json(out)
  if world > 1:
    adapter.close()
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
