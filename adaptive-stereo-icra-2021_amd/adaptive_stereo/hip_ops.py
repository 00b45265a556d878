"""torch.autograd.Function wrappers over the C ABI (include/adaptive_stereo_hip.h).

Each Function is a hand-written forward AND backward: autograd only stitches them
together.  Nothing here computes on the CPU; every call goes through
``_native.call`` on the current HIP stream with raw device pointers.
"""
import os
import torch

from . import _native as nat
from ._native import Pcl, ConvShape, call, ptr, stream, f32c

LEAKY_SLOPE = 0.2
BN_EPS = 1e-5
BN_MOMENTUM = 0.1

CONV3D_333 = ConvShape(3, 3, 3, 1, 1, 1, 1, 1)

# Direct gradient accumulation: for parameters that adaptation.FlatArena has re-homed (it tags them with
# ``_as_grad_sink``; their .grad tensors are views of one flat buffer that is zeroed at the start of a step)
# the backward kernels ADD parameter gradients straight into p.grad and the Functions return None for them —
# autograd then issues no "grad += dW" kernel per tensor (~120 launches per step).  Parameters without the tag
# (plain torch.optim / torch.autograd.grad use) keep ordinary autograd semantics.  The switch below turns the
# mechanism off globally (parity tests compare both routes).
_DIRECT_GRADS = True


def set_direct_grad_accumulation(flag: bool):
  global _DIRECT_GRADS
  _DIRECT_GRADS = bool(flag)


def grad_sinks(params):
  """Per-parameter accumulation targets (or None where a parameter has no usable .grad)."""
  if not (_DIRECT_GRADS and torch.is_grad_enabled()):
    return None
  out = []
  for p in params:
    g = getattr(p, "grad", None)
    ok = (g is not None and getattr(p, "_as_grad_sink", False) and p.requires_grad and
          g.dtype == torch.float32 and g.is_contiguous() and g.device == p.device and g.shape == p.shape)
    out.append(g if ok else None)
  return out


def _sink(sinks, i):
  return sinks[i] if sinks is not None else None


# ----------------------------------------------------------------------------------------
# PCL helpers (allocation + layout conversion; conversion is host plumbing used at the
# boundary with NCHW tensors and by tests)
# ----------------------------------------------------------------------------------------
def pcl_zeros(g: Pcl, device):
  return torch.zeros(g.numel(), dtype=torch.float32, device=device)


def pcl_view(buf, g: Pcl):
  return buf.view(g.B, g.D + 2 * g.pd, g.H + 2 * g.ph, g.W + 2 * g.pw, 32)


def pcl_interior(buf, g: Pcl):
  v = pcl_view(buf, g)
  return v[:, g.pd:g.pd + g.D, g.ph:g.ph + g.H, g.pw:g.pw + g.W, :]


def pcl_to_ncdhw(buf, g: Pcl):
  return pcl_interior(buf, g).permute(0, 4, 1, 2, 3).contiguous()


def ncdhw_to_pcl(t, g: Pcl):
  buf = pcl_zeros(g, t.device)
  pcl_interior(buf, g).copy_(t.permute(0, 2, 3, 4, 1))
  return buf


def _empty(n, device, dtype=torch.float32):
  return torch.empty(int(n), dtype=dtype, device=device)


class PclPool(object):
  """Recycles PCL buffers.  Kernels only ever write the interior of a PCL tensor, so a buffer
  that was allocated zero-filled keeps a zero halo for life and can be handed out again without
  a memset (a full-resolution activation is 60 MB per pair: clearing ~30 of them per step would
  cost more HBM traffic than the soft-argmax path moves in total).
  Free lists are per HIP stream: a buffer returned while stream S is current is handed out again only to work
  issued on S, which the stream orders after everything that touched the buffer before (other streams' uses
  reached S through the event wait of their hand-over).  No cross-stream synchronisation is ever needed."""

  def __init__(self):
    self.free = {}

  @staticmethod
  def _key(device, g, channels):
    return (str(device), torch.cuda.current_stream(device).cuda_stream, g.key(), channels)

  def get(self, g: Pcl, device, channels=32):
    lst = self.free.get(self._key(device, g, channels))
    if lst:
      return lst.pop()
    return torch.zeros(g.numel() // 32 * channels, dtype=torch.float32, device=device)

  def put(self, buf, g: Pcl, channels=32):
    if buf is None:
      return
    self.free.setdefault(self._key(buf.device, g, channels), []).append(buf)

  def clear(self):
    self.free.clear()


POOL = PclPool()


# ----------------------------------------------------------------------------------------
# Read-modify-write ordering across HIP streams
# ----------------------------------------------------------------------------------------
class _RmwOrder(object):
  """When independent parts of a step run on different HIP streams (the two feature extractions of a pair), the few
  buffers both parts UPDATE IN PLACE — gradient sinks in the flat arena, BatchNorm running statistics — must be
  touched in the order a single stream would: whoever is issued second waits for an event the first recorded after
  its kernel.  Issue order is Python order, which is deterministic, so results stay bit-identical to the one-stream
  step.  Inside a graph capture the events become edges between the parallel branches.  Disabled (no events, no
  cost) unless a caller opens a multi-stream region."""
  enabled = False
  last = {}          # data_ptr -> (stream handle, event recorded after the last read-modify-write)
  streams = {}       # stream handle -> torch stream, every stream that recorded such an event in this region


def rmw_order_reset(enabled):
  """Opens (True) or closes (False) a multi-stream region; forgets events of the previous region (an event recorded
  outside a graph capture must not be waited on inside it).  Closing runs the deferred weight-gradient reductions."""
  if not enabled:
    flush_deferred_reductions()
  _RmwOrder.enabled = bool(enabled)
  _RmwOrder.last = {}
  _RmwOrder.streams = {}
  if enabled and _DEFER_REDUCE and torch.cuda.is_available():
    lib = nat.load()
    if lib.as_wgrad_defer_pending() != 0 or _DeferredReduce.keep:
      raise RuntimeError("adaptive_stereo: a multi-stream region opens while weight-gradient reductions of an earlier one "
                         "are still pending (a region was left without rmw_order_reset(False))")
    _DeferredReduce.on = True
    lib.as_wgrad_defer(1)


class _DeferredReduce(object):
  """Inside a step's multi-stream region the ~30 slab reductions behind the weight-gradient kernels (4.5 us each, nothing
  reads their result before the optimizer) are recorded by the library and run in one launch when the region closes
  (as_wgrad_defer / as_wgrad_defer_flush).  `keep` holds the workspaces until then."""
  on = False
  keep = []


_DEFER_REDUCE = True       # set_defer_reduce(False): every reduction right behind its kernel (the parity test compares both)


def set_defer_reduce(enabled):
  global _DEFER_REDUCE
  prev, _DEFER_REDUCE = _DEFER_REDUCE, bool(enabled)
  return prev


def _keep_for_deferred_reduce(ws):
  if _DeferredReduce.on:
    _DeferredReduce.keep.append(ws)


def flush_deferred_reductions():
  """Runs the recorded weight-gradient reductions (one launch) on the current stream and ends the deferral.  Called when the
  multi-stream region closes and, under data parallelism, before the gradient all-reduce.  Everything that produced a slab
  must already be ordered before the current stream: join_region_streams() makes that explicit for every stream that
  issued work inside the region (autograd itself only joins streams through defined gradients, and the Functions here
  return None for sunk parameters)."""
  join_region_streams()
  if not _DeferredReduce.on:
    return
  lib = nat.load()
  lib.as_wgrad_defer(0)
  _DeferredReduce.on = False
  call("as_wgrad_defer_flush", stream())
  _DeferredReduce.keep = []


_TAIL_BNSUMS = True        # conv2d_out's data gradient also sums for the last block's BatchNorm backward
_HEAD_PROJ = True          # conv2d_feature backward: per-tap projections instead of g_z
_FWD_ACT = True            # full-resolution training forward: previous BatchNorm + LeakyReLU applied on the way in
_WINOGRAD = os.environ.get("AS_DIAG_DIRECT_FORM") != "1"   # ... and its 3x3 convolution by the minimal-filtering algorithm
                           # F(2x2, 3x3) (csrc/conv32_wino.hip); the variable is for A/B diagnostics, set_winograd() the API
# (module switches, each with a setter: tests/test_gpu_end_to_end.py runs both routes and compares them; timing A/B of a
# route is a tool's business — tests/tools/ab_switch.py — not the environment's)


def set_tail_bnsums(enabled):
  global _TAIL_BNSUMS
  prev, _TAIL_BNSUMS = _TAIL_BNSUMS, bool(enabled)
  return prev


def set_head_proj(enabled):
  global _HEAD_PROJ
  prev, _HEAD_PROJ = _HEAD_PROJ, bool(enabled)
  return prev


_REFINE_OUT = True       # the refinement's output layer on csrc/refine_out.hip (False: bn_act_fwd + conv32to1_2d_fwd_kernel)


def set_refine_out(enabled):
  """Output layer of the refinement (with the last block's BatchNorm + LeakyReLU + skip on the way in) as one launch, or as
  the two launches of rounds 1-3 (parity tests compare both); returns the previous setting."""
  global _REFINE_OUT
  prev, _REFINE_OUT = _REFINE_OUT, bool(enabled)
  return prev


def set_head_staged(enabled):
  """The strided head of the feature towers on the staged-row kernels (csrc/conv32_s2.hip, conv4_s2_* in csrc/conv4_mfma.hip) or
  on the generic gather kernels: the library's two switches together (the parity tests compare both routes kernel by kernel;
  tests/tools/ab_switch.py times them); returns the previous setting."""
  lib = nat.load()
  prev = lib.as_conv32_s2_enable(2) == 1 and lib.as_conv4_s2_enable(2) == 1
  lib.as_conv32_s2_enable(1 if enabled else 0)
  lib.as_conv4_s2_enable(1 if enabled else 0)
  return prev


def set_fwd_act(enabled):
  global _FWD_ACT
  prev, _FWD_ACT = _FWD_ACT, bool(enabled)
  return prev


_WINOGRAD_BWD = True       # (diagnostics: False keeps the direct fused backward while the forward uses minimal filtering)
_WINOGRAD_BWD_ONE_LAUNCH = True   # both gradients of a layer in ONE launch (csrc/conv32_wino_bwd.hip); False: data gradient, then
                                  # weight gradient, g_z through HBM in between (as_conv32_wino_bwd) — A/B runs and parity tests


def set_winograd_bwd_one_launch(flag: bool):
  """Switches the minimal-filtering backward of the full-resolution layers between the one-launch kernel and the two launches
  it replaced (same g_x bit for bit, dW to summation order); returns the previous setting."""
  global _WINOGRAD_BWD_ONE_LAUNCH
  prev, _WINOGRAD_BWD_ONE_LAUNCH = _WINOGRAD_BWD_ONE_LAUNCH, bool(flag)
  return prev


def set_winograd(enabled, backward=None):
  """False: the full-resolution layers keep the direct form (as_conv32_act_fwd, as_conv32_bwd_fused, as_conv32_fwd).
  ``backward`` (default: same as ``enabled``) switches the two backward launches separately.  Returns the previous setting."""
  global _WINOGRAD, _WINOGRAD_BWD
  prev, _WINOGRAD = _WINOGRAD, bool(enabled)
  _WINOGRAD_BWD = bool(enabled if backward is None else backward)
  return prev


# ----------------------------------------------------------------------------------------
# Weight-gradient kernels beside the data-gradient chain.  In the backward pass of a layer the weight gradient (reads the
# layer's input and its output gradient, writes private slabs) and the data gradient (reads the output gradient, writes the
# input gradient the NEXT layer's backward needs) are independent; where neither fills the chip on its own — the strided head
# (0.23 / 0.56 of the matrix peak), the 3-D aggregation layers (52 + 50 us on 90 K voxels), the 32->1 output layer — the
# weight gradient CAN be launched on a side stream at the point the output gradient is ready and joined right behind the data
# gradient: the same kernels, the same bits, in a captured step two parallel branches of the graph.
# OFF by default: measured on one box, interleaved (profiles/r05_l_ab_wgrad_beside*.txt), the step is not faster with it —
# 6.91-7.02 ms against 6.80-6.98 at four pairs, 2.52-2.54 against 2.45-2.46 at one: two kernels that each already occupy every
# CU slow each other down by what the overlap gains, and the fork / join edges add launch latency.  AS_WGRAD_BESIDE=1 or
# set_wgrad_beside(True) turns it on (tests/test_gpu_end_to_end.py holds it to the one-stream bits).
# ----------------------------------------------------------------------------------------
class _Beside(object):
  enabled = os.environ.get("AS_WGRAD_BESIDE", "0") == "1"
  stream = None
  origin = None          # inside a capture: the handle of the stream that may fork (the capture's origin stream)
  forks = 0


def set_wgrad_beside(flag: bool):
  prev, _Beside.enabled = _Beside.enabled, bool(flag)
  return prev


def set_fork_origin(handle):
  """OnlineAdapter tells which stream a capture it opened runs on: a fork from an already forked stream inside a capture
  crashes hipStreamEndCapture on ROCm 7.2, so inside a capture only that stream forks (anywhere else the body runs inline)."""
  prev, _Beside.origin = _Beside.origin, handle
  return prev


def fork_beside(fn):
  """Runs fn() — launches that READ what the current stream has produced so far and WRITE only buffers nobody else touches
  before join_beside() — on the side stream; returns the handle join_beside() takes (None: fn ran inline)."""
  main = torch.cuda.current_stream()
  if (not _Beside.enabled or _BN_SYNC is not None or
      (torch.cuda.is_current_stream_capturing() and main.cuda_stream != _Beside.origin)):
    fn()
    return None
  if _Beside.stream is None:
    _Beside.stream = torch.cuda.Stream()
  side = _Beside.stream
  side.wait_stream(main)
  with torch.cuda.stream(side):
    fn()
  ev = torch.cuda.Event()
  ev.record(side)
  _Beside.forks += 1
  return ev


def join_beside(ev):
  if ev is not None:
    torch.cuda.current_stream().wait_event(ev)


def _rmw_wait(t):
  if _RmwOrder.enabled and t is not None:
    last = _RmwOrder.last.get(t.data_ptr())
    cur = torch.cuda.current_stream()
    if last is not None and last[0] != cur.cuda_stream:
      cur.wait_event(last[1])


def _rmw_done(t):
  if _RmwOrder.enabled and t is not None:
    cur = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(cur)
    _RmwOrder.last[t.data_ptr()] = (cur.cuda_stream, ev)
    _RmwOrder.streams[cur.cuda_stream] = cur


def join_region_streams():
  """The current stream waits for every other stream that updated a shared buffer inside the open multi-stream region
  (weight-gradient slabs and gradient sinks written on the right tower's stream are read by the flush and the optimizer
  on this one).  Until round 3 this edge was implicit (the left tower's _rmw_wait chain happened to order it)."""
  if not _RmwOrder.enabled:
    return
  cur = torch.cuda.current_stream()
  for handle, st in list(_RmwOrder.streams.items()):
    if handle != cur.cuda_stream:
      cur.wait_stream(st)


# ----------------------------------------------------------------------------------------
# Thin call helpers
# ----------------------------------------------------------------------------------------
def pack_weights(w, shape: ConvShape, transpose_flip: bool):
  w = f32c(w)
  plan = _ACTIVE_PLAN
  if plan is not None:
    hit = plan.packed(w, shape.taps(), int(bool(transpose_flip)), shape.taps() * 1024)
    if hit is not None:
      return hit
  packed = _empty(shape.taps() * 1024, w.device)
  call("as_conv32_pack_weights", ptr(w), ptr(packed), shape, int(transpose_flip), stream())
  return packed


# kinds of include/adaptive_stereo_hip.h (as_pack_job.transpose_flip)
PACK_S2_DGRAD, PACK_CONV4, PACK_MIRROR_TAP, PACK_MIRROR_CH, PACK_WINO, PACK_WINO_T = 2, 16, 32, 33, 34, 35


def pack_special(w, kind, taps, size, direct):
  """A weight-derived buffer other than the standard packing (the strided data gradient's phase-major taps, the 4-channel
  layers' packing, the mirrored channel-0 taps): from the StepPlan's one batch launch when a plan is active, else through
  ``direct(out)`` (the stand-alone entry point)."""
  w = f32c(w)
  plan = _ACTIVE_PLAN
  if plan is not None:
    hit = plan.packed(w, taps, kind, size)
    if hit is not None:
      return hit
  out = _empty(size, w.device)
  direct(w, out)
  return out


class StepPlan(object):
  """Batches a step's many tiny launches.  The first step it sees runs normally and is RECORDED (which weights
  get packed in which orientation, which BatchNorm layers count a batch, how often); from then on
    * begin()  packs every recorded weight with ONE kernel (as_conv32_pack_weights_batch) into persistent
               buffers that pack_weights() hands out, and
    * end()    bumps all num_batches_tracked counters with ONE add on an int64 arena
  instead of ~55 + ~23 launches.  Valid only while the weights do not change between begin() and end()
  (adaptation.OnlineAdapter: they change in the optimizer step, after end()).  Anything not seen while
  recording falls back to the ordinary per-call path, so a different control flow stays correct."""

  def __init__(self, enabled=True):
    self.enabled = enabled     # False: begin()/end() do nothing (every call takes the ordinary path)
    self.ready = False
    self._rec_pack = {}        # (data_ptr, taps, kind) -> (weight tensor, floats of the packed buffer)
    self._rec_bn = {}          # id(bn module) -> [bn module, count]
    self._rec_affine = {}      # gamma.data_ptr() -> (gamma, beta, running_mean, running_var)   (eval-mode layers)
    self._affine_states = {}
    self._affine_jobs = None
    self._views = {}
    self._jobs = None
    self._bn_index = {}
    self._pending = None

  # -- weights ---------------------------------------------------------------------------------------
  def packed(self, w, taps, kind, size):
    key = (w.data_ptr(), int(taps), int(kind))
    if not self.ready:
      self._rec_pack[key] = (w, int(size))
      return None
    return self._views.get(key)

  # -- eval-mode BatchNorm affines -------------------------------------------------------------------
  def eval_affine(self, gamma, beta, running_mean, running_var):
    key = gamma.data_ptr()
    if not self.ready:
      self._rec_affine[key] = (gamma, beta, running_mean, running_var)
      return None
    return self._affine_states.get(key)

  # -- batch counters --------------------------------------------------------------------------------
  def count(self, bn):
    """Returns True when the plan takes care of this layer's counter."""
    if not self.ready:
      self._rec_bn.setdefault(id(bn), [bn, 0])[1] += 1
      return False
    i = self._bn_index.get(id(bn))
    if i is None:
      return False
    self._pending[i] += 1
    return True

  def begin(self, resume=False):
    """``resume=True``: the backward section of a step whose forward section ended with end(final=False)."""
    global _ACTIVE_PLAN
    if not self.enabled:
      return
    _ACTIVE_PLAN = self
    if not self.ready and not resume:
      for rec in self._rec_bn.values():
        rec[1] = 0               # counts are per step: an earlier forward-only recording does not add up
    if self.ready:
      if self._jobs is not None:
        call("as_conv32_pack_weights_batch", ptr(self._jobs), self._njobs, self._max_taps, stream())
      if self._affine_jobs is not None:
        call("as_bn_eval_affine_batch", ptr(self._affine_jobs), len(self._affine_states), BN_EPS, stream())
      self._pending = [0] * len(self._bn_index)

  def end(self, final=True):
    """``final=False``: a forward-only section ends, the recording (if any) goes on into the backward section."""
    global _ACTIVE_PLAN
    if not self.enabled:
      return
    _ACTIVE_PLAN = None
    if not self.ready:
      if final:
        self._build()
      else:
        self._flush_recorded_counts()
      return
    if self._pending and any(self._pending):
      if self._pending == self._inc_host:
        self._nbt.add_(self._inc)
      else:
        self._nbt.add_(torch.tensor(self._pending, dtype=torch.int64).to(self._nbt.device))
    self._pending = None

  def _flush_recorded_counts(self):
    pass   # while recording, count_batch() already bumped the counters the ordinary way

  def _build(self):
    import struct
    if self._rec_pack:
      dev = next(iter(self._rec_pack.values()))[0].device
      total = sum((size + 3) // 4 * 4 for _, size in self._rec_pack.values())
      self._buf = torch.empty(total, dtype=torch.float32, device=dev)
      blob, off = b"", 0
      for key, (w, size) in self._rec_pack.items():
        view = self._buf[off:off + size]
        self._views[key] = view
        blob += struct.pack("<QQii", w.data_ptr(), view.data_ptr(), key[1], int(key[2]))
        off += (size + 3) // 4 * 4             # every packed buffer 16-byte aligned
      self._weights = [w for w, _ in self._rec_pack.values()]     # keep the storages alive
      self._jobs = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
      self._njobs = len(self._rec_pack)
      self._max_taps = max(k[1] for k in self._rec_pack)
    if self._rec_affine:
      dev = next(iter(self._rec_affine.values()))[0].device
      self._affine_buf = torch.empty(len(self._rec_affine), 4, 32, dtype=torch.float32, device=dev)
      blob = b""
      for i, (key, (ga, be, rm, rv)) in enumerate(self._rec_affine.items()):
        self._affine_states[key] = BnState(dev, self._affine_buf[i])
        blob += struct.pack("<QQQQQ", ga.data_ptr(), be.data_ptr(), rm.data_ptr(), rv.data_ptr(), self._affine_buf[i].data_ptr())
      self._affine_keep = list(self._rec_affine.values())        # keep the storages alive
      self._affine_jobs = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    if self._rec_bn:
      mods = [m for m, _ in self._rec_bn.values()]
      dev = mods[0].num_batches_tracked.device
      self._nbt = torch.stack([m.num_batches_tracked.to(torch.int64).reshape(()) for m in mods]).to(dev)
      for i, m in enumerate(mods):
        m._buffers["num_batches_tracked"] = self._nbt[i]           # same state_dict key, now an arena view
        self._bn_index[id(m)] = i
      self._inc_host = [c for _, c in self._rec_bn.values()]
      self._inc = torch.tensor(self._inc_host, dtype=torch.int64).to(dev)
    self.ready = True


_ACTIVE_PLAN = None


def count_batch(bn):
  """``bn.num_batches_tracked += 1`` (nn.BatchNorm in training mode), batched when a StepPlan is active."""
  plan = _ACTIVE_PLAN
  if plan is not None and plan.count(bn):
    return
  bn.num_batches_tracked += 1


def conv32(x, gin: Pcl, packed_w, bias, gout: Pcl, shape: ConvShape, out=None, epilogue=0, scale=None,
           shift=None, residual=None, stats=None):
  """Returns the PCL output buffer. ``stats`` = StatParts to fill (train-mode BatchNorm)."""
  z = out if out is not None else POOL.get(gout, x.device)
  sm, s2, sc = (stats.mean, stats.m2, stats.cnt) if stats is not None else (None, None, None)
  call("as_conv32_fwd", ptr(x), gin, ptr(packed_w), ptr(bias), ptr(z), gout, shape, int(epilogue),
       ptr(scale), ptr(shift), LEAKY_SLOPE, ptr(residual), ptr(sm), ptr(s2), ptr(sc), stream())
  return z


class StatParts(object):
  """BatchNorm partial moments written by a convolution epilogue: (count, mean, M2) per workgroup."""

  def __init__(self, nparts, device):
    self.nparts = int(nparts)
    buf = torch.empty(self.nparts * 65, dtype=torch.float32, device=device)
    self.buf = buf
    self.mean = buf[:self.nparts * 32]
    self.m2 = buf[self.nparts * 32:self.nparts * 64]
    self.cnt = buf[self.nparts * 64:]


def conv32_stat_parts(gin: Pcl, gout: Pcl, shape: ConvShape, device):
  return StatParts(nat.load().as_conv32_stat_parts(gin, gout, shape), device)


def conv32_wgrad(x, gin: Pcl, gz, gout: Pcl, shape: ConvShape, want_bias=True, sink_w=None, sink_b=None):
  """Returns (dW, db); an entry is None when it was accumulated into its sink instead."""
  lib = nat.load()
  dev = x.device
  taps = shape.taps()
  if sink_w is not None and (sink_b is not None or not want_bias):
    ws = _empty(lib.as_conv32_wgrad_workspace(gin, gout, shape), dev)
    _keep_for_deferred_reduce(ws)
    _rmw_wait(sink_w)
    call("as_conv32_wgrad", ptr(x), gin, ptr(gz), gout, shape, ptr(sink_w), ptr(sink_b), 1, ptr(ws), stream())
    _rmw_done(sink_w)
    return None, None
  ws = _empty(lib.as_conv32_wgrad_workspace(gin, gout, shape), dev)
  if shape.kd > 1:
    dW = _empty(32 * 32 * taps, dev).view(32, 32, shape.kd, shape.kh, shape.kw)
  else:
    dW = _empty(32 * 32 * taps, dev).view(32, 32, shape.kh, shape.kw)
  db = _empty(32, dev) if want_bias else None
  # (accumulate = 0: the library reduces right away even inside a deferral region — autograd reads dW on return)
  call("as_conv32_wgrad", ptr(x), gin, ptr(gz), gout, shape, ptr(dW), ptr(db), 0, ptr(ws), stream())
  return dW, db


class BnState(object):
  """Per-layer BatchNorm scalars living on the device: scale/shift (the affine the
  activation pass applies) and mean/invstd (what backward needs)."""

  def __init__(self, device, buf=None):
    if buf is None:
      buf = torch.empty(4, 32, dtype=torch.float32, device=device)
    self.mean, self.invstd, self.scale, self.shift = buf[0], buf[1], buf[2], buf[3]


class BnSync(object):
  """Cross-replica BatchNorm for data-parallel adaptation: train-mode statistics and the backward's
  per-channel sums span the batches of ALL ranks of ``group``, as the reference's single-process BatchNorm over the
  whole batch does (stereo_net.py:17,29).  Every rank merges the same partials in the same (rank-major) order, so
  the replicas stay bit-identical.  ``comm`` = this library's own RCCL communicator (rccl.RcclComm): its collectives are
  enqueued on the current stream and can be captured into a hipGraph; without it the collectives go through
  torch.distributed and stepping is eager while this is on."""

  def __init__(self, group=None, comm=None):
    import torch.distributed as dist
    self.dist, self.group, self.comm = dist, group, comm
    self.world = dist.get_world_size(group)


_BN_SYNC = None


def set_bn_sync(sync):
  """BnSync or None; returns the previous setting."""
  global _BN_SYNC
  prev, _BN_SYNC = _BN_SYNC, sync
  return prev


def _gathered_stats(stats: StatParts, sync: BnSync):
  """The (count, mean, M2) partials of all ranks, rank-major, in one StatParts."""
  n, w = stats.nparts, sync.world
  every = torch.empty(w, n * 65, dtype=torch.float32, device=stats.buf.device)
  if sync.comm is not None:
    sync.comm.all_gather(every, stats.buf)
  else:
    sync.dist.all_gather(list(every.unbind(0)), stats.buf, group=sync.group)
  out = StatParts(n * w, stats.buf.device)
  out.mean.view(w, n * 32).copy_(every[:, :n * 32])
  out.m2.view(w, n * 32).copy_(every[:, n * 32:n * 64])
  out.cnt.view(w, n).copy_(every[:, n * 64:])
  return out


def bn_train_stats(stats: StatParts, gamma, beta, running_mean, running_var):
  st = BnState(gamma.device)
  if _BN_SYNC is not None:
    stats = _gathered_stats(stats, _BN_SYNC)
  _rmw_wait(running_mean)
  call("as_bn_finalize", ptr(stats.mean), ptr(stats.m2), ptr(stats.cnt), stats.nparts, ptr(gamma), ptr(beta),
       ptr(running_mean), ptr(running_var), BN_MOMENTUM, BN_EPS, ptr(st.mean), ptr(st.invstd),
       ptr(st.scale), ptr(st.shift), stream())
  _rmw_done(running_mean)
  return st


def bn_eval_stats(gamma, beta, running_mean, running_var):
  plan = _ACTIVE_PLAN
  if plan is not None:
    hit = plan.eval_affine(gamma, beta, running_mean, running_var)
    if hit is not None:
      return hit
  st = BnState(gamma.device)
  call("as_bn_eval_affine", ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), BN_EPS,
       ptr(st.mean), ptr(st.invstd), ptr(st.scale), ptr(st.shift), stream())
  return st


def bn_act(z, st: BnState, g: Pcl, residual=None, out=None):
  a = out if out is not None else POOL.get(g, z.device)
  call("as_bn_act_fwd", ptr(z), ptr(st.scale), ptr(st.shift), LEAKY_SLOPE, ptr(residual), ptr(a), g, stream())
  return a


class BnBwdSums(object):
  """Stage 1 of a BatchNorm backward (per-channel sums) that a fused producer has already left in a workspace."""

  def __init__(self, workspace, nparts):
    self.workspace, self.nparts = workspace, nparts


def bn_bwd_coefs(g_a, z, st: BnState, gamma, g: Pcl, train: bool, g_gamma, g_beta, accumulate, ws, sums=None):
  """Stages 1-2 of the BatchNorm backward (stage 1 may already be in ``sums``): parameter gradients into g_gamma /
  g_beta, stage-3 coefficients left in ``ws`` at as_bn_bwd_coef_offset()."""
  sync = _BN_SYNC if train else None
  if accumulate:
    _rmw_wait(g_gamma)
  try:
    _bn_bwd_coefs(g_a, z, st, gamma, g, train, g_gamma, g_beta, accumulate, ws, sums, sync)
  finally:
    if accumulate:
      _rmw_done(g_gamma)


def _bn_bwd_coefs(g_a, z, st, gamma, g, train, g_gamma, g_beta, accumulate, ws, sums, sync):
  if sync is None:
    if sums is not None:
      call("as_bn_act_bwd_given", ptr(g_a), ptr(z), ptr(st.scale), ptr(st.shift), ptr(st.mean), ptr(st.invstd), ptr(gamma),
           LEAKY_SLOPE, int(train), None, ptr(g_gamma), ptr(g_beta), int(accumulate), ptr(ws), g, sums.nparts, stream())
    else:
      call("as_bn_act_bwd", ptr(g_a), ptr(z), ptr(st.scale), ptr(st.shift), ptr(st.mean), ptr(st.invstd), ptr(gamma),
           LEAKY_SLOPE, int(train), None, ptr(g_gamma), ptr(g_beta), int(accumulate), ptr(ws), g, stream())
    return
  local = torch.empty(65, dtype=torch.float64, device=z.device)
  call("as_bn_bwd_sums", ptr(g_a), ptr(z), ptr(st.scale), ptr(st.shift), ptr(st.mean), LEAKY_SLOPE, ptr(ws), g,
       sums.nparts if sums is not None else 0, ptr(local), stream())
  everyone = local.clone()
  if sync.comm is not None:
    sync.comm.all_reduce(everyone)
  else:
    sync.dist.all_reduce(everyone, op=sync.dist.ReduceOp.SUM, group=sync.group)
  call("as_bn_bwd_finalize_synced", ptr(local), ptr(everyone), ptr(st.invstd), ptr(gamma), ptr(g_gamma), ptr(g_beta),
       int(accumulate), ptr(ws), stream())


def bn_act_bwd(g_a, z, st: BnState, gamma, g: Pcl, train: bool, sink_gamma=None, sink_beta=None, sums=None):
  lib = nat.load()
  dev = z.device
  g_z = POOL.get(g, dev)
  ws = sums.workspace if sums is not None else _empty(lib.as_bn_bwd_workspace(g), dev)
  if train and _BN_SYNC is not None:
    sunk = sink_gamma is not None and sink_beta is not None
    g_gamma, g_beta = (sink_gamma, sink_beta) if sunk else (_empty(32, dev), _empty(32, dev))
    bn_bwd_coefs(g_a, z, st, gamma, g, train, g_gamma, g_beta, sunk, ws, sums)
    call("as_bn_bwd_apply", ptr(g_a), ptr(z), ptr(st.scale), ptr(st.shift), ptr(st.mean), LEAKY_SLOPE, ptr(g_z), ptr(ws),
         g, stream())
    return (g_z, None, None) if sunk else (g_z, g_gamma, g_beta)
  fn = "as_bn_act_bwd_given" if sums is not None else "as_bn_act_bwd"
  tail = (ptr(ws), g, sums.nparts, stream()) if sums is not None else (ptr(ws), g, stream())
  if sink_gamma is not None and sink_beta is not None:
    _rmw_wait(sink_gamma)
    call(fn, ptr(g_a), ptr(z), ptr(st.scale), ptr(st.shift), ptr(st.mean), ptr(st.invstd), ptr(gamma),
         LEAKY_SLOPE, int(train), ptr(g_z), ptr(sink_gamma), ptr(sink_beta), 1, *tail)
    _rmw_done(sink_gamma)
    return g_z, None, None
  g_gamma, g_beta = _empty(32, dev), _empty(32, dev)
  call(fn, ptr(g_a), ptr(z), ptr(st.scale), ptr(st.shift), ptr(st.mean), ptr(st.invstd), ptr(gamma),
       LEAKY_SLOPE, int(train), ptr(g_z), ptr(g_gamma), ptr(g_beta), 0, *tail)
  return g_z, g_gamma, g_beta


def conv32_dgrad_bnbwd(g_z, g: Pcl, wp_t, shape: ConvShape, residual, next_z, next_st: BnState):
  """g_x = dgrad(g_z) (+ residual) fused with stage 1 of the BatchNorm backward of the layer whose output gradient
  g_x is (pre-activation next_z, statistics next_st).  Returns (g_x, BnBwdSums) or None when the configuration has no
  fused kernel."""
  lib = nat.load()
  nparts = lib.as_conv32_bnbwd_parts(g, g, shape)
  if nparts <= 0:
    return None
  dev = g_z.device
  g_x = POOL.get(g, dev)
  ws = _empty(lib.as_bn_bwd_workspace(g), dev)
  call("as_conv32_fwd_bnbwd", ptr(g_z), g, ptr(wp_t), ptr(g_x), g, shape, ptr(residual), ptr(next_z), ptr(next_st.scale),
       ptr(next_st.shift), ptr(next_st.mean), LEAKY_SLOPE, ptr(ws), stream())
  return g_x, BnBwdSums(ws, nparts)


# ----------------------------------------------------------------------------------------
# conv -> BatchNorm -> LeakyReLU (-> + skip) blocks on PCL tensors: shared by the 3-D aggregation
# (stereo_net.py:155-161), the feature extractor's and the refinement's BasicBlocks (:33-51)
# ----------------------------------------------------------------------------------------
def conv_shape_2d(dilation=1):
  pad = dilation if dilation > 1 else 1
  return ConvShape(1, 3, 3, 0, pad, pad, dilation, 1)


def block_forward(x, g: Pcl, shape: ConvShape, w, b, gamma, beta, rm, rv, train, skip, keep_z):
  """Returns (z or None, a, BnState).  train: batch statistics (+ running-stat update);
  otherwise running statistics, fused into the convolution epilogue when z is not needed."""
  dev = x.device
  wino_eval = (not train) and (not keep_z) and _WINOGRAD and nat.load().as_conv32_wino_ok(g, g, shape) == 1
  wp = None if wino_eval else pack_weights(w, shape, False)
  if train:
    stats = conv32_stat_parts(g, g, shape, dev)
    z = conv32(x, g, wp, b, g, shape, stats=stats)
    st = bn_train_stats(stats, gamma, beta, rm, rv)
    a = bn_act(z, st, g, residual=x if skip else None)
  else:
    st = bn_eval_stats(gamma, beta, rm, rv)
    if keep_z:
      z = conv32(x, g, wp, b, g, shape)
      a = bn_act(z, st, g, residual=x if skip else None)
    elif wino_eval:
      # inference: the block in one launch by minimal filtering (csrc/conv32_wino.hip MODE 3) — 4 matrix products per pixel
      # instead of 9, x read once (the skip connection comes out of the staged rows)
      z = None
      ww = pack_special(w, PACK_WINO, 16, 16 * 1024,
                        lambda w_, o_: call("as_conv32_wino_pack_weights", ptr(w_), ptr(o_), 0, stream()))
      a = POOL.get(g, x.device)
      call("as_conv32_wino_eval", ptr(x), g, shape, ptr(ww), ptr(b), ptr(st.scale), ptr(st.shift), LEAKY_SLOPE, 1 if skip else 0,
           ptr(a), stream())
    else:
      z = None
      a = conv32(x, g, wp, b, g, shape, epilogue=1, scale=st.scale, shift=st.shift, residual=x if skip else None)
  if not keep_z and z is not None:
    POOL.put(z, g); z = None
  return z, a, st


def block_forward_act(z_prev, a_prevprev, st_prev: BnState, g: Pcl, shape: ConvShape, w, b, gamma, beta, rm, rv):
  """Train-mode forward of a full-resolution layer whose operand a_prev = lrelu(BN(z_prev)) (+ a_prevprev) is formed on
  the way in (csrc/conv32_act.hip): returns (a_prev, z, BnState of this layer).  The standalone bn_act pass of the
  previous layer (read z_prev, read a_prevprev, write a_prev) disappears."""
  dev = z_prev.device
  lib = nat.load()
  a_prev, z = POOL.get(g, dev), POOL.get(g, dev)
  if _WINOGRAD and lib.as_conv32_wino_ok(g, g, shape) == 1:
    # F(2x2, 3x3): 4 matrix products per output pixel instead of 9; the transformed filters come out of the step's one
    # packing launch like every other weight-derived buffer
    ww = pack_special(w, PACK_WINO, 16, 16 * 1024,
                      lambda w_, o_: call("as_conv32_wino_pack_weights", ptr(w_), ptr(o_), 0, stream()))
    stats = StatParts(lib.as_conv32_wino_parts(), dev)
    call("as_conv32_wino_fwd", ptr(z_prev), ptr(a_prevprev), ptr(st_prev.scale), ptr(st_prev.shift), ptr(a_prev), g, ptr(ww),
         ptr(b), LEAKY_SLOPE, ptr(z), g, shape, ptr(stats.mean), ptr(stats.m2), ptr(stats.cnt), stream())
    return a_prev, z, bn_train_stats(stats, gamma, beta, rm, rv)
  wp = pack_weights(w, shape, False)
  stats = StatParts(lib.as_conv32_act_parts(), dev)
  call("as_conv32_act_fwd", ptr(z_prev), ptr(a_prevprev), ptr(st_prev.scale), ptr(st_prev.shift), ptr(a_prev), g, ptr(wp), ptr(b),
       LEAKY_SLOPE, ptr(z), g, shape, ptr(stats.mean), ptr(stats.m2), ptr(stats.cnt), stream())
  st = bn_train_stats(stats, gamma, beta, rm, rv)
  return a_prev, z, st


def block_backward(g_out, x, z, st, w, gamma, g: Pcl, shape: ConvShape, train, skip, need_dx, sinks=None,
                   sums=None, next_bn=None):
  """g_out: gradient w.r.t. the block output (PCL).  Returns (g_x or None, dW, db, g_gamma, g_beta, next_sums).
  With a skip connection g_x = g_out + dgrad(...) — the add is fused into the dgrad epilogue.
  ``sinks`` = (w, b, gamma, beta) accumulation targets; sunk gradients come back as None.
  ``sums``: stage 1 of this block's BatchNorm backward, if the producer of g_out already computed it;
  ``next_bn`` = (z, BnState) of the layer whose output gradient g_x is: its stage 1 is then fused into this block's
  data gradient and returned as next_sums (None when not available)."""
  sw, sb, sg, sbeta = sinks if sinks is not None else (None, None, None, None)
  lib = nat.load()
  beside = None
  all_sunk = sw is not None and sb is not None and sg is not None and sbeta is not None
  if _BWD_FUSED and all_sunk and skip and need_dx and train and next_bn is not None and _BN_SYNC is None and \
      lib.as_conv32_bwd_fused_ok(g, g, shape) == 1:
    # the whole backward of the layer in ONE launch (csrc/conv32_bwd.hip): stages 1-2 of its BatchNorm backward first
    # (stage 1 usually already in `sums`), then stage 3 + data gradient + skip + weight/bias gradient + stage 1 of the
    # next BatchNorm backward; g_z never leaves the chip
    dev = z.device
    ws = sums.workspace if sums is not None else _empty(lib.as_bn_bwd_workspace(g), dev)
    bn_bwd_coefs(g_out, z, st, gamma, g, train, sg, sbeta, True, ws, sums)
    coef = ws[lib.as_bn_bwd_coef_offset():]
    g_x = POOL.get(g, dev)
    nws = _empty(lib.as_bn_bwd_workspace(g), dev)
    next_z, next_st = next_bn
    if _WINOGRAD and _WINOGRAD_BWD and lib.as_conv32_wino_ok(g, g, shape) == 1:
      # both gradients by minimal filtering (csrc/conv32_wino.hip MODE 2, csrc/conv32_wino_wgrad.hip): 4 matrix products per
      # pixel and gradient instead of 9; g_z makes one round trip through HBM between the two launches
      ww_t = pack_special(w, PACK_WINO_T, 16, 16 * 1024,
                          lambda w_, o_: call("as_conv32_wino_pack_weights", ptr(w_), ptr(o_), 1, stream()))
      if _WINOGRAD_BWD_ONE_LAUNCH:
        # ... in one launch (csrc/conv32_wino_bwd.hip): g_z stays in LDS between the data-gradient and the weight-gradient waves
        wws = _empty(lib.as_conv32_wino_bwd_fused_workspace(), dev)
        _keep_for_deferred_reduce(wws)
        _rmw_wait(sw)
        call("as_conv32_wino_bwd_fused", ptr(x), g, ptr(g_out), ptr(z), g, shape, ptr(ww_t), ptr(st.scale), ptr(st.shift),
             ptr(st.mean), ptr(coef), LEAKY_SLOPE, ptr(next_z), ptr(next_st.scale), ptr(next_st.shift), ptr(next_st.mean),
             ptr(g_x), ptr(sw), ptr(sb), 1, ptr(nws), ptr(wws), stream())
        _rmw_done(sw)
        return g_x, None, None, None, None, BnBwdSums(nws, lib.as_conv32_wino_bwd_fused_parts())
      wws = _empty(lib.as_conv32_wino_bwd_workspace(), dev)
      _keep_for_deferred_reduce(wws)
      g_z = POOL.get(g, dev)
      _rmw_wait(sw)
      call("as_conv32_wino_bwd", ptr(x), g, ptr(g_out), ptr(z), g, shape, ptr(ww_t), ptr(st.scale), ptr(st.shift), ptr(st.mean),
           ptr(coef), LEAKY_SLOPE, ptr(next_z), ptr(next_st.scale), ptr(next_st.shift), ptr(next_st.mean), ptr(g_z), ptr(g_x),
           ptr(sw), ptr(sb), 1, ptr(nws), ptr(wws), stream())
      _rmw_done(sw)
      POOL.put(g_z, g)
      return g_x, None, None, None, None, BnBwdSums(nws, lib.as_conv32_wino_bwd_parts())
    wp_t = pack_weights(w, shape, True)
    wws = _empty(lib.as_conv32_bwd_fused_workspace(), dev)
    _keep_for_deferred_reduce(wws)
    _rmw_wait(sw)
    call("as_conv32_bwd_fused", ptr(x), g, ptr(g_out), ptr(z), g, shape, ptr(wp_t), ptr(st.scale), ptr(st.shift), ptr(st.mean),
         ptr(coef), LEAKY_SLOPE, ptr(next_z), ptr(next_st.scale), ptr(next_st.shift), ptr(next_st.mean), ptr(g_x), ptr(sw),
         ptr(sb), 1, ptr(nws), ptr(wws), stream())
    _rmw_done(sw)
    return g_x, None, None, None, None, BnBwdSums(nws, lib.as_conv32_bwd_fused_parts())
  if all_sunk and \
      lib.as_conv32_wgrad_bnapply_ok(g, g, shape) == 1:
    # stages 1-2 of the BatchNorm backward only (stage 1 may already be in `sums`); stage 3 rides on the weight gradient,
    # which stages g_out and z rows, applies it in LDS, accumulates dW / db and writes g_z for the data gradient
    dev = z.device
    ws = sums.workspace if sums is not None else _empty(lib.as_bn_bwd_workspace(g), dev)
    bn_bwd_coefs(g_out, z, st, gamma, g, train, sg, sbeta, True, ws, sums)
    coef = ws[lib.as_bn_bwd_coef_offset():]
    g_z = POOL.get(g, dev)
    wws = _empty(lib.as_conv32_wgrad_workspace(g, g, shape), dev)
    _keep_for_deferred_reduce(wws)
    _rmw_wait(sw)
    call("as_conv32_wgrad_bnapply", ptr(x), g, ptr(g_out), ptr(z), g, shape, ptr(st.scale), ptr(st.shift), ptr(st.mean),
         ptr(coef), LEAKY_SLOPE, ptr(g_z), ptr(sw), ptr(sb), 1, ptr(wws), stream())
    _rmw_done(sw)
    g_gamma = g_beta = dW = db = None
  else:
    g_z, g_gamma, g_beta = bn_act_bwd(g_out, z, st, gamma, g, train, sg, sbeta, sums)
    if sw is not None and sb is not None and need_dx:
      # (sunk gradients: nothing comes back to autograd) beside the data gradient below
      dW = db = None
      beside = fork_beside(lambda: conv32_wgrad(x, g, g_z, g, shape, True, sw, sb))
    else:
      dW, db = conv32_wgrad(x, g, g_z, g, shape, True, sw, sb)
  g_x, next_sums = None, None
  if need_dx:
    wp_t = pack_weights(w, shape, True)
    fused = None
    if next_bn is not None and train:
      fused = conv32_dgrad_bnbwd(g_z, g, wp_t, shape, g_out if skip else None, next_bn[0], next_bn[1])
    if fused is not None:
      g_x, next_sums = fused
    elif shape.kd == 3 and not skip and agg3d_ok(g):
      g_x = agg3d(g_z, g, wp_t, None, epilogue=2)            # rolling-window kernel: the data gradient of a 3-D layer
    else:
      g_x = conv32(g_z, g, wp_t, None, g, shape, residual=g_out if skip else None)
  join_beside(beside)
  POOL.put(g_z, g)
  return g_x, dW, db, g_gamma, g_beta, next_sums


# ----------------------------------------------------------------------------------------
# a3, second generation (csrc/agg3d.hip): rolling-window 3-D aggregation layer with the previous layer's BatchNorm +
# LeakyReLU applied to the operand in LDS, that BatchNorm merged from the previous launch's partials by the consumer
# ----------------------------------------------------------------------------------------
_AGG3D = True            # False: the first-generation path (conv3d_lds + finalize + element-wise pass per layer)
_AGG_TAIL = True         # False: conv3d_alone and the soft-argmax as two launches (conv32to1_fwd + softargmax_fwd)
_BWD_FUSED = True        # False: a full-resolution layer's backward as two launches (wgrad + BN stage 3 | dgrad + BN stage 1)


def set_agg3d(flag: bool):
  """Switches the cost-aggregation forward between the rolling-window kernels and the first-generation path
  (A/B measurements and parity tests); returns the previous setting."""
  global _AGG3D
  prev, _AGG3D = _AGG3D, bool(flag)
  return prev


def set_bwd_fused(flag: bool):
  global _BWD_FUSED
  prev, _BWD_FUSED = _BWD_FUSED, bool(flag)
  return prev


# Who merges a train-mode BatchNorm's per-workgroup partials: the consuming kernel (up to this many partials) or a finalize
# launch in front of it.  Measured inside a step at 4 pairs, same box, interleaved (tests/tools/ab_env_step.sh,
# profiles/r05_u_ab_*_merge.txt):
#   the tail kernel (one wave per SIMD, 234 workgroups, each merging on its own): 38.9 us merging itself against 7.6 + 22.4 us
#     with a finalize launch -> the launch wins by 9 us: 0;
#   aggregation layers 2-4: 71.0 us merging themselves against 7.8 + 62.0 us -> a wash, and a launch boundary more in the replayed
#     graph: they keep merging (512; beyond that the reads through L2 grow quadratically).
_TAIL_MERGE_MAX = int(os.environ.get("AS_TAIL_MERGE_MAX", "0"))
_AGG_MERGE_MAX = int(os.environ.get("AS_AGG_MERGE_MAX", "512"))
_TAIL_BWD = True         # False: the tail's backward as three launches (as_softargmax_bwd, then as_conv3d_out_bwd's two)


def set_tail_bwd(flag: bool):
  """Test switch: the one-launch backward of the aggregation tail (csrc/agg_tail_bwd.hip) on / off."""
  global _TAIL_BWD
  prev, _TAIL_BWD = _TAIL_BWD, bool(flag)
  return prev


def set_agg_tail(flag: bool):
  global _AGG_TAIL
  prev, _AGG_TAIL = _AGG_TAIL, bool(flag)
  return prev


def agg3d_ok(g: Pcl):
  return _AGG3D and nat.load().as_agg3d_ok(g) == 1


class PendingBn(object):
  """A train-mode BatchNorm whose statistics are still per-workgroup partials (written by the producing convolution):
  the kernel that consumes the layer's output merges them itself (csrc/bn_merge.h) and writes ``state`` (a BnState: the
  backward pass reads it) and the running statistics.  Keeps every tensor it points to alive."""

  def __init__(self, stats: StatParts, gamma, beta, running_mean, running_var):
    self.stats, self.gamma, self.beta, self.rm, self.rv = stats, gamma, beta, running_mean, running_var
    self.state = BnState(gamma.device)
    st = self.state
    self.block = nat.BnMerge(ptr(stats.mean), ptr(stats.m2), ptr(stats.cnt), ptr(gamma), ptr(beta), ptr(running_mean),
                             ptr(running_var), ptr(st.mean), ptr(st.invstd), ptr(st.scale), ptr(st.shift), stats.nparts,
                             BN_MOMENTUM, BN_EPS)

  def finalize(self):
    """The same result through the stand-alone finalize launch (no fused consumer available)."""
    _rmw_wait(self.rm)
    st = self.state
    call("as_bn_finalize", ptr(self.stats.mean), ptr(self.stats.m2), ptr(self.stats.cnt), self.stats.nparts, ptr(self.gamma),
         ptr(self.beta), ptr(self.rm), ptr(self.rv), BN_MOMENTUM, BN_EPS, ptr(st.mean), ptr(st.invstd), ptr(st.scale),
         ptr(st.shift), stream())
    _rmw_done(self.rm)
    return st


def agg3d(x, g: Pcl, packed_w, bias, z=None, in_state=None, in_bn=None, a_out=None, epilogue=0, ep_state=None, stats=None):
  """One aggregation layer (as_agg3d_fwd).  x is the previous layer's raw output when ``in_state`` (BnState: finalized
  affine) or ``in_bn`` (PendingBn: partials, merged by the kernel) is given: its BatchNorm + LeakyReLU is applied on the
  fly, and a_out (optional) receives the activated tensor.  ``stats`` (StatParts) receives this layer's BatchNorm
  partials.  Returns z."""
  z = z if z is not None else POOL.get(g, x.device)
  sm, s2, sc = (stats.mean, stats.m2, stats.cnt) if stats is not None else (None, None, None)
  if in_bn is not None:
    _rmw_wait(in_bn.rm)
  call("as_agg3d_fwd", ptr(x), g, ptr(packed_w), ptr(bias), ptr(in_state.scale) if in_state is not None else None,
       ptr(in_state.shift) if in_state is not None else None, in_bn.block if in_bn is not None else None, ptr(a_out), ptr(z),
       int(epilogue), ptr(ep_state.scale) if ep_state is not None else None,
       ptr(ep_state.shift) if ep_state is not None else None, LEAKY_SLOPE, ptr(sm), ptr(s2), ptr(sc), stream())
  if in_bn is not None:
    _rmw_done(in_bn.rm)
  return z


# ----------------------------------------------------------------------------------------
# a2-a5 (+a8): cost volume -> 4x(conv3d+BN+LeakyReLU) -> conv3d 32->1 -> soft-argmax
# Reference: StereoNet.forward, adaptive_stereo/models/stereo_net.py:173-192
# ----------------------------------------------------------------------------------------
class CostAggregationFn(torch.autograd.Function):
  """inputs : fl, fr [B,32,H,W]; 4 x (conv weight, conv bias, bn weight, bn bias); out conv weight, bias
     outputs: logits [B,D,H,W], pred [B,H,W], argmax int32 [B,H,W], fcs [B,H,W]"""

  @staticmethod
  def forward(ctx, fl, fr, num_disp, train, grad_on, bn_buffers, sinks, *params):
    assert len(params) == 18
    fl, fr = f32c(fl), f32c(fr)
    params = [f32c(p) for p in params]
    B, C, H, W = fl.shape
    if C != 32 or fr.shape != fl.shape:
      raise RuntimeError("CostAggregationFn: features must be [B,32,H,W] and equal in shape")
    dev = fl.device
    D = int(num_disp)
    g = Pcl(B, D, H, W, 1, 1, 1)
    need_bwd = grad_on and any(ctx.needs_input_grad)   # grad_on: the caller's grad mode (forward() itself always runs
    # with grad disabled, and needs_input_grad stays True under torch.no_grad())

    vol = POOL.get(g, dev)
    call("as_cost_volume_fwd", ptr(fl), ptr(fr), ptr(vol), g, stream())
    xs, zs, sts = [vol], [], []
    tail_in = None
    if agg3d_ok(g) and ((train and _BN_SYNC is None) or (not train and not need_bwd)):
      # rolling-window layers: in train mode layer l+1 applies layer l's BatchNorm + LeakyReLU to its operand in LDS (and
      # writes the activated tensor back only when a backward pass will need it); each launch finalizes its own BatchNorm
      lib = nat.load()
      nparts = lib.as_agg3d_parts(g)
      # every workgroup of the consumer merges ALL of the producer's partials: fine for a few hundred (240 at 4 KITTI pairs),
      # quadratic beyond (k = 3 volumes: 1,632 workgroups x 1,632 partials = 0.7 GB through L2 per layer, measured 2x the
      # layer's time) — there one finalize launch per layer is the cheaper form
      merge_in_consumer = nparts <= _AGG_MERGE_MAX
      x, prev = vol, None
      for l in range(4):
        w, b, gamma, beta = params[4 * l:4 * l + 4]
        rm, rv = bn_buffers[l]
        wp = pack_weights(w, CONV3D_333, False)
        if train:
          a_prev = POOL.get(g, dev) if (prev is not None and need_bwd) else None
          pending = PendingBn(StatParts(nparts, dev), gamma, beta, rm, rv)
          if prev is not None and not merge_in_consumer:
            z = agg3d(x, g, wp, b, in_state=prev.finalize(), a_out=a_prev, stats=pending.stats)
          else:
            z = agg3d(x, g, wp, b, in_bn=prev, a_out=a_prev, stats=pending.stats)
          if prev is not None:
            xs.append(a_prev)                   # a_l (None when nothing will read it)
            if not need_bwd:
              POOL.put(x, g)                    # z_l was only this layer's operand
          zs.append(z); sts.append(pending.state)
          x, prev = z, pending
        else:
          st = bn_eval_stats(gamma, beta, rm, rv)
          a = agg3d(x, g, wp, b, epilogue=1, ep_state=st)          # BatchNorm folded into the epilogue
          zs.append(None); sts.append(st); xs.append(a)
          x = a
      if train:
        tail_in = (x, prev)                     # z4 + its pending BatchNorm: finalized and applied by the tail kernel
    else:
      for l in range(4):
        w, b, gamma, beta = params[4 * l:4 * l + 4]
        rm, rv = bn_buffers[l]
        z, a, st = block_forward(xs[-1], g, CONV3D_333, w, b, gamma, beta, rm, rv, train, False, need_bwd)
        zs.append(z); sts.append(st); xs.append(a)

    w_out, b_out = params[16], params[17]
    logits = torch.empty(B, D, H, W, dtype=torch.float32, device=dev)
    pred = torch.empty(B, H, W, dtype=torch.float32, device=dev)
    argmax = torch.empty(B, H, W, dtype=torch.int32, device=dev)
    fcs = torch.empty(B, H, W, dtype=torch.float32, device=dev)
    fused_tail = _AGG_TAIL and nat.load().as_agg_tail_ok(g) == 1
    if tail_in is not None:
      z4, bn4 = tail_in
      if fused_tail:
        # one launch: layer 4's BatchNorm (merged from its partials) + LeakyReLU on the way in, 32->1 convolution,
        # soft-argmax, arg-max, FCS
        a4 = POOL.get(g, dev) if need_bwd else None
        if bn4.stats.nparts <= _TAIL_MERGE_MAX:
          _rmw_wait(bn4.rm)
          call("as_agg_tail_fwd", ptr(z4), g, None, None, bn4.block, ptr(a4), ptr(w_out), ptr(b_out), LEAKY_SLOPE,
               ptr(logits), ptr(pred), ptr(argmax), ptr(fcs), stream())
          _rmw_done(bn4.rm)
        else:
          st4 = bn4.finalize()
          call("as_agg_tail_fwd", ptr(z4), g, ptr(st4.scale), ptr(st4.shift), None, ptr(a4), ptr(w_out), ptr(b_out), LEAKY_SLOPE,
               ptr(logits), ptr(pred), ptr(argmax), ptr(fcs), stream())
        keep_alive = bn4                        # (its tensors are referenced by the launch just issued)
      else:
        a4 = bn_act(z4, bn4.finalize(), g)
      xs.append(a4)
      if not need_bwd:
        POOL.put(z4, g)
    if fused_tail:
      if tail_in is None:
        call("as_agg_tail_fwd", ptr(xs[4]), g, None, None, None, None, ptr(w_out), ptr(b_out), LEAKY_SLOPE,
             ptr(logits), ptr(pred), ptr(argmax), ptr(fcs), stream())
    else:
      call("as_conv3d_out_fwd", ptr(xs[4]), g, ptr(w_out), ptr(b_out), ptr(logits), stream())
      call("as_softargmax_fwd", ptr(logits), B, D, H, W, ptr(pred), ptr(argmax), ptr(fcs), stream())

    if need_bwd:
      ctx.g = g
      ctx.train = bool(train)
      ctx.sinks = sinks
      ctx.xs, ctx.zs, ctx.sts = xs, zs, sts
      ctx.save_for_backward(logits, *params)
    else:
      for buf in xs:
        POOL.put(buf, g)
    ctx.mark_non_differentiable(argmax, fcs)
    ctx.set_materialize_grads(False)       # an output nobody differentiates arrives as None, not as a zero-filled tensor
    return logits, pred, argmax, fcs

  @staticmethod
  def backward(ctx, g_logits_in, g_pred, _g_argmax, _g_fcs):
    logits, *params = ctx.saved_tensors
    g = ctx.g
    dev = logits.device
    B, D, H, W = logits.shape
    lib = nat.load()
    xs, zs, sts = ctx.xs, ctx.zs, ctx.sts

    g_pred, g_logits_in = f32c(g_pred), f32c(g_logits_in)      # keep any contiguous copies alive past the launch (None = zero)
    grads = [None] * 18
    w_out = params[16]
    sinks = ctx.sinks
    g_a = POOL.get(g, dev)
    into_sinks = _sink(sinks, 16) is not None and _sink(sinks, 17) is not None
    if into_sinks:
      g_wout, g_bout = sinks[16], sinks[17]
    else:
      g_wout, g_bout = torch.empty_like(w_out), _empty(1, dev)
      grads[16], grads[17] = g_wout, g_bout
    if _TAIL_BWD and lib.as_agg_tail_bwd_ok(g) == 1:
      # soft-argmax backward + both gradients of conv3d_alone in one launch: the logits gradient stays in LDS
      ws = _empty(lib.as_agg_tail_bwd_workspace(g), dev)
      call("as_agg_tail_bwd", ptr(logits), ptr(g_pred), ptr(g_logits_in), ptr(xs[4]), g, ptr(w_out), ptr(g_a), ptr(g_wout),
           ptr(g_bout), 1 if into_sinks else 0, ptr(ws), stream())
    else:
      g_logits = torch.empty_like(logits)
      call("as_softargmax_bwd", ptr(logits), ptr(g_pred), ptr(g_logits_in), B, D, H, W, ptr(g_logits), stream())
      ws = _empty(lib.as_conv3d_out_bwd_workspace(g), dev)
      call("as_conv3d_out_bwd", ptr(g_logits), ptr(xs[4]), g, ptr(w_out), ptr(g_a), ptr(g_wout), ptr(g_bout),
           1 if into_sinks else 0, ptr(ws), stream())

    need_feat = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
    for l in range(3, -1, -1):
      w, b, gamma, beta = params[4 * l:4 * l + 4]
      g_x, dW, db, g_gamma, g_beta, _ = block_backward(g_a, xs[l], zs[l], sts[l], w, gamma, g, CONV3D_333, ctx.train,
                                                       False, l > 0 or need_feat,
                                                       sinks[4 * l:4 * l + 4] if sinks is not None else None)
      grads[4 * l:4 * l + 4] = [dW, db, g_gamma, g_beta]
      POOL.put(g_a, g); POOL.put(zs[l], g); POOL.put(xs[l + 1], g)
      g_a = g_x
    g_fl = g_fr = None
    if need_feat:
      g_fl = torch.empty(B, 32, H, W, dtype=torch.float32, device=dev)
      g_fr = torch.empty_like(g_fl)
      call("as_cost_volume_bwd", ptr(g_a), ptr(g_fl), ptr(g_fr), g, stream())
      POOL.put(g_a, g)
    POOL.put(xs[0], g)
    ctx.xs = ctx.zs = ctx.sts = None
    return (g_fl, g_fr, None, None, None, None, None) + tuple(grads)


# ----------------------------------------------------------------------------------------
# a1: FeatureExtractorNetwork (stereo_net.py:54-85): k strided 5x5 convolutions (no activation in
# between), six BasicBlocks, conv_alone — whole module in PCL, forward and hand-written backward
# ----------------------------------------------------------------------------------------
CONV5_S2 = ConvShape(1, 5, 5, 0, 2, 2, 1, 2)


def _down(n):
  return (n - 1) // 2 + 1


_TRUNK = True             # False: the trunk on the generic kernels (conv + finalize + activation launches per block)


def set_trunk(flag: bool):
  """Train-mode trunk of the feature extractor on the one-launch-per-layer kernels (csrc/trunk.hip) or on the generic
  per-operation kernels (parity tests compare both); returns the previous setting."""
  global _TRUNK
  prev, _TRUNK = _TRUNK, bool(flag)
  return prev


def trunk_enabled():
  return _TRUNK and _BN_SYNC is None


def adjacent_or_cat(left, right):
  """[left; right] along the batch: a view when the two already sit back to back in one buffer, else a copy."""
  if (left.is_contiguous() and right.is_contiguous() and left.shape == right.shape and left.dtype == right.dtype
      and left.untyped_storage().data_ptr() == right.untyped_storage().data_ptr()
      and right.data_ptr() == left.data_ptr() + left.numel() * left.element_size()):
    return torch.as_strided(left, (2 * left.shape[0],) + tuple(left.shape[1:]), left.stride(), left.storage_offset())
  return torch.cat([left, right])


def _host_ptrs(tensors):
  return (nat.c_vp * len(tensors))(*[t.data_ptr() for t in tensors])


class FeatureExtractorFn(torch.autograd.Function):
  """rgb [G*B,3,H,W] -> features [G*B,32,Hc,Wc] (NCHW, the public layout).  ``groups`` = G statistics groups: G = 2 is the
  two images of a pair (left batch, then right batch) through ONE pass — the reference calls feature_net twice
  (adapt.py:72), so train-mode BatchNorm moments, running-statistics updates and gradients are taken per group.
  params: k x (downsample w, b), 6 x (conv w, conv b, bn w, bn b), conv_alone (w, b)."""

  @staticmethod
  def forward(ctx, rgb, k, groups, train, grad_on, bn_buffers, sinks, *params):
    k, groups = int(k), int(groups)
    assert len(params) == 2 * k + 26
    rgb = f32c(rgb)
    params = [f32c(p) for p in params]
    B, C, H, W = rgb.shape
    if C != 3:
      raise RuntimeError("FeatureExtractorFn: expected an RGB image [B,3,H,W]")
    if groups < 1 or B % groups != 0:
      raise RuntimeError("FeatureExtractorFn: %d images do not split into %d statistics groups" % (B, groups))
    dev = rgb.device
    lib = nat.load()
    need_bwd = grad_on and any(ctx.needs_input_grad)   # grad_on: the caller's grad mode (forward() itself always runs
    # with grad disabled, and needs_input_grad stays True under torch.no_grad())

    # head: level 0 = image (PCL4, halo 2); level i = output of downsample[i-1].  No BatchNorm: batch-independent
    g4 = Pcl(B, 1, H, W, 0, 2, 2)
    in4 = POOL.get(g4, dev, channels=4)
    call("as_pack_in4", None, ptr(rgb), 3, ptr(in4), g4, stream())
    geoms, levels = [], []
    h, w = H, W
    for i in range(k):
      h, w = _down(h), _down(w)
      halo = 2 if i + 1 < k else 1            # feeds another 5x5 stride-2 conv, or the 3x3 trunk
      gi = Pcl(B, 1, h, w, 0, halo, halo)
      out = POOL.get(gi, dev)
      wd, bd = params[2 * i], params[2 * i + 1]
      if i == 0:
        wp = pack_special(wd, PACK_CONV4 + 3, 25, 25 * 128,
                          lambda w_, o_: call("as_conv4_pack_weights", ptr(w_), 3, ptr(o_), CONV5_S2, stream()))
        call("as_conv4_fwd", ptr(in4), g4, ptr(wp), ptr(bd), ptr(out), gi, CONV5_S2, 0, None, None, LEAKY_SLOPE,
             None, None, None, stream())
      else:
        conv32(levels[-1], geoms[-1], pack_weights(wd, CONV5_S2, False), bd, gi, CONV5_S2, out=out)
      geoms.append(gi); levels.append(out)

    g = geoms[-1]
    shape = conv_shape_2d(1)
    tp = params[2 * k:]
    trunk = bool(train) and _TRUNK and _BN_SYNC is None
    if not trunk and groups != 1 and train:
      raise RuntimeError("FeatureExtractorFn: several statistics groups in train mode need the trunk kernels "
                         "(set_trunk(True), no cross-replica BatchNorm)")
    states = None
    if trunk:
      # one launch per layer for all groups: layer l forms its operand a_{l-1} = lrelu(BN(z_{l-1})) + a_{l-2} while staging,
      # merging BN_{l-1} from the partials launch l-1 left behind, and leaves a_{l-1} as a by-product (csrc/trunk.hip)
      parts = lib.as_trunk_parts(g, groups)
      states = torch.empty(6, groups, 5, 32, dtype=torch.float32, device=dev)
      xs, zs, sts = [levels[-1]], [], []
      prev_bn, keep = None, []
      for l in range(7):
        wl, bl = tp[4 * l], tp[4 * l + 1]
        z = POOL.get(g, dev)
        stats = StatParts(groups * parts, dev) if l < 6 else None
        sm, s2, sc = (stats.mean, stats.m2, stats.cnt) if stats is not None else (None, None, None)
        if l == 0:
          call("as_trunk_fwd", ptr(xs[0]), None, None, None, g, groups, ptr(pack_weights(wl, shape, False)), ptr(bl), LEAKY_SLOPE,
               ptr(z), ptr(sm), ptr(s2), ptr(sc), stream())
        else:
          a_prev = POOL.get(g, dev)
          call("as_trunk_fwd", ptr(zs[-1]), ptr(xs[-1]), prev_bn, ptr(a_prev), g, groups, ptr(pack_weights(wl, shape, False)),
               ptr(bl), LEAKY_SLOPE, ptr(z), ptr(sm), ptr(s2), ptr(sc), stream())
          xs.append(a_prev)
        if l < 6:
          gamma, beta = tp[4 * l + 2], tp[4 * l + 3]
          prev_bn = nat.TrunkBn(ptr(sm), ptr(s2), ptr(sc), ptr(gamma), ptr(beta), ptr(states[l]), parts, BN_EPS)
          keep.append(stats)
          zs.append(z)
        else:
          out = z
      feats = torch.empty(B, 32, g.H, g.W, dtype=torch.float32, device=dev)
      rms = [bn_buffers[l][0] for l in range(6)]
      rvs = [bn_buffers[l][1] for l in range(6)]
      _rmw_wait(rms[0])
      call("as_trunk_finish_fwd", ptr(out), g, ptr(feats), ptr(states), 6, groups, _host_ptrs(rms), _host_ptrs(rvs), BN_MOMENTUM,
           stream())
      _rmw_done(rms[0])
      POOL.put(out, g)
      if not need_bwd:
        for buf in xs[1:] + zs:
          POOL.put(buf, g)
    else:
      xs, zs, sts = [levels[-1]], [], []
      for l in range(6):
        wl, bl, gamma, beta = tp[4 * l:4 * l + 4]
        rm, rv = bn_buffers[l]
        z, a, st = block_forward(xs[-1], g, shape, wl, bl, gamma, beta, rm, rv, train, True, need_bwd)
        zs.append(z); sts.append(st); xs.append(a)
        if not need_bwd:
          POOL.put(xs[-2], g)
      out = conv32(xs[6], g, pack_weights(tp[24], shape, False), tp[25], g, shape)
      feats = pcl_interior(out, g)[:, 0].permute(0, 3, 1, 2).contiguous()
      POOL.put(out, g)
    ctx.pair_out = groups == 2
    if need_bwd:
      ctx.k, ctx.g4, ctx.geoms, ctx.train, ctx.groups = k, g4, geoms, bool(train), groups
      ctx.sinks = sinks
      ctx.in4, ctx.levels = in4, levels
      ctx.xs, ctx.zs, ctx.sts, ctx.states = xs, zs, sts, states
      ctx.save_for_backward(*params)
    else:
      if not trunk:
        POOL.put(xs[-1], g)
      POOL.put(in4, g4, channels=4)
      for buf, gi in zip(levels[:-1] if not trunk else levels, geoms[:-1] if not trunk else geoms):
        POOL.put(buf, gi)
    if ctx.pair_out:
      # the two groups as two outputs: autograd hands their gradients over separately (slicing one output afterwards costs a
      # zero-fill and a copy per slice and an add in backward)
      return feats[:B // 2], feats[B // 2:]
    return feats

  @staticmethod
  def backward(ctx, *g_outs):
    if ctx.needs_input_grad[0]:
      raise NotImplementedError("FeatureExtractorFn: gradient w.r.t. the image is not part of the adaptation path")
    params = ctx.saved_tensors
    k, g4, geoms = ctx.k, ctx.g4, ctx.geoms
    g = geoms[-1]
    xs, zs, sts, levels = ctx.xs, ctx.zs, ctx.sts, ctx.levels
    dev = g_outs[0].device
    lib = nat.load()
    shape = conv_shape_2d(1)
    tp = params[2 * k:]
    grads = [None] * len(params)

    sinks = ctx.sinks
    g_out = POOL.get(g, dev)
    g_outs = [f32c(t) for t in g_outs]
    call("as_trunk_begin_bwd", ptr(g_outs[0]), ptr(g_outs[1]) if len(g_outs) > 1 else None, g_outs[0].shape[0], g, ptr(g_out),
         stream())
    if ctx.states is not None:
      g_a = FeatureExtractorFn._trunk_backward(ctx, g_out, tp, grads, sinks, dev)
    else:
      dW, db = conv32_wgrad(xs[6], g, g_out, g, shape, True, _sink(sinks, 2 * k + 24), _sink(sinks, 2 * k + 25))
      grads[2 * k + 24], grads[2 * k + 25] = dW, db
      g_a = conv32(g_out, g, pack_weights(tp[24], shape, True), None, g, shape)
      POOL.put(g_out, g)
      for l in range(5, -1, -1):
        wl, bl, gamma, beta = tp[4 * l:4 * l + 4]
        g_x, dW, db, g_gamma, g_beta, _ = block_backward(
            g_a, xs[l], zs[l], sts[l], wl, gamma, g, shape, ctx.train, True, True,
            sinks[2 * k + 4 * l:2 * k + 4 * l + 4] if sinks is not None else None)
        grads[2 * k + 4 * l:2 * k + 4 * l + 4] = [dW, db, g_gamma, g_beta]
        POOL.put(g_a, g); POOL.put(zs[l], g); POOL.put(xs[l + 1], g)
        g_a = g_x

    # head, last strided convolution first.  g_a is the gradient w.r.t. levels[k-1] (geometry geoms[k-1]).
    for i in range(k - 1, -1, -1):
      wd = params[2 * i]
      gi = geoms[i]
      if i == 0:
        ws = _empty(lib.as_conv4_wgrad_workspace(gi, CONV5_S2), dev)
        if _sink(sinks, 0) is not None and _sink(sinks, 1) is not None:
          _rmw_wait(sinks[0])
          call("as_conv4_wgrad", ptr(ctx.in4), g4, ptr(g_a), gi, CONV5_S2, 3, ptr(sinks[0]), ptr(sinks[1]), 1, ptr(ws),
               stream())
          _rmw_done(sinks[0])
        else:
          dW = torch.empty_like(wd); db = _empty(32, dev)
          call("as_conv4_wgrad", ptr(ctx.in4), g4, ptr(g_a), gi, CONV5_S2, 3, ptr(dW), ptr(db), 0, ptr(ws), stream())
          grads[0], grads[1] = dW, db
      else:
        gprev = geoms[i - 1]
        sw_, sb_ = _sink(sinks, 2 * i), _sink(sinks, 2 * i + 1)

        def head_wgrad(i=i, gprev=gprev, gi=gi, g_a=g_a, sw_=sw_, sb_=sb_):
          dW, db = conv32_wgrad(levels[i - 1], gprev, g_a, gi, CONV5_S2, True, sw_, sb_)
          grads[2 * i], grads[2 * i + 1] = dW, db
        # sunk gradients only: a dW handed back to autograd would be consumed on the main stream
        ev = fork_beside(head_wgrad) if (sw_ is not None and sb_ is not None) else head_wgrad()
        g_prev = POOL.get(gprev, dev)
        wps = pack_special(wd, PACK_S2_DGRAD, 25, 25 * 1024,
                           lambda w_, o_: call("as_conv32_dgrad_s2_pack", ptr(w_), ptr(o_), stream()))
        call("as_conv32_dgrad_s2_packed", ptr(g_a), gi, ptr(wps), ptr(g_prev), gprev, stream())
        join_beside(ev)
        POOL.put(g_a, gi)
        g_a = g_prev
    POOL.put(g_a, geoms[0])
    POOL.put(ctx.in4, g4, channels=4)
    for buf, gi in zip(levels, geoms):
      POOL.put(buf, gi)
    ctx.xs = ctx.zs = ctx.sts = ctx.levels = ctx.in4 = ctx.states = None
    return (None, None, None, None, None, None, None) + tuple(grads)

  @staticmethod
  def _trunk_backward(ctx, g_out, tp, grads, sinks, dev):
    """The trunk's backward on csrc/trunk.hip: conv_alone, then blocks 6..1, one launch each for all groups; returns the
    gradient w.r.t. the head's output.  Parameter gradients go straight to their sinks, or come back through ``grads``."""
    lib = nat.load()
    k, groups, g = ctx.k, ctx.groups, ctx.geoms[-1]
    xs, zs, states = ctx.xs, ctx.zs, ctx.states
    shape = conv_shape_2d(1)
    parts = lib.as_trunk_parts(g, groups)
    base = 2 * k

    def dest(i, like):
      s_ = _sink(sinks, base + i)
      if s_ is not None:
        return s_, True
      return torch.empty_like(like), False

    bn_grads = torch.empty(6, groups, 2, 32, dtype=torch.float32, device=dev)
    sums = [torch.empty(groups * parts * 64, dtype=torch.float64, device=dev) for _ in range(6)]
    g_a = g_out
    for l in range(6, -1, -1):
      wl, bl = tp[4 * l], tp[4 * l + 1]
      dW, sunk_w = dest(4 * l, wl)
      db, sunk_b = dest(4 * l + 1, bl)
      if sunk_w != sunk_b:                       # one flag for both: fall back to fresh tensors
        dW, db, sunk_w = torch.empty_like(wl), torch.empty_like(bl), False
      ws = _empty(lib.as_trunk_bwd_workspace(g, groups), dev)
      if sunk_w:
        _keep_for_deferred_reduce(ws)
        _rmw_wait(dW)
      g_x = POOL.get(g, dev)
      zn, stn, sn = (zs[l - 1], states[l - 1], sums[l - 1]) if l >= 1 else (None, None, None)
      if l == 6:
        call("as_trunk_bwd", ptr(g_a), None, None, None, 0, None, None, ptr(xs[6]), ptr(pack_weights(wl, shape, True)), ptr(g_x),
             ptr(zn), ptr(stn), ptr(sn), g, groups, LEAKY_SLOPE, ptr(dW), ptr(db), int(sunk_w), ptr(ws), stream())
      else:
        call("as_trunk_bwd", ptr(g_a), ptr(zs[l]), ptr(states[l]), ptr(sums[l]), parts, ptr(tp[4 * l + 2]), ptr(bn_grads[l]),
             ptr(xs[l]), ptr(pack_weights(wl, shape, True)), ptr(g_x), ptr(zn), ptr(stn), ptr(sn), g, groups, LEAKY_SLOPE,
             ptr(dW), ptr(db), int(sunk_w), ptr(ws), stream())
      if sunk_w:
        _rmw_done(dW)
      else:
        grads[base + 4 * l], grads[base + 4 * l + 1] = dW, db
      POOL.put(g_a, g)
      if l < 6:
        POOL.put(zs[l], g)
      POOL.put(xs[l + 1] if l < 6 else None, g)
      g_a = g_x
    gg, gb, all_sunk = [], [], True
    for l in range(6):
      d1, s1 = dest(4 * l + 2, tp[4 * l + 2])
      d2, s2 = dest(4 * l + 3, tp[4 * l + 3])
      if not (s1 and s2):
        all_sunk = False
      gg.append(d1); gb.append(d2)
    if not all_sunk:
      gg = [torch.empty_like(tp[4 * l + 2]) for l in range(6)]
      gb = [torch.empty_like(tp[4 * l + 3]) for l in range(6)]
    else:
      _rmw_wait(gg[0])
    call("as_trunk_finish_bwd", ptr(bn_grads), 6, groups, _host_ptrs(gg), _host_ptrs(gb), int(all_sunk), stream())
    if all_sunk:
      _rmw_done(gg[0])
    else:
      for l in range(6):
        grads[base + 4 * l + 2], grads[base + 4 * l + 3] = gg[l], gb[l]
    return g_a


# ----------------------------------------------------------------------------------------
# a7: EdgeAwareRefinement (stereo_net.py:88-121), whole module in PCL, forward and backward
# ----------------------------------------------------------------------------------------
REFINE_DILATIONS = (1, 2, 4, 8, 1, 1)
REFINE_HALO = 8


class EdgeRefineFn(torch.autograd.Function):
  """coarse [B,h,w], guidance rgb [B,3,H,W] -> refined disparity [B,1,H,W].
  params: conv2d_feature (w[32,4,3,3], b, bn w, bn b), 6 x (w[32,32,3,3], b, bn w, bn b), conv2d_out (w[1,32,3,3], b)."""

  @staticmethod
  def forward(ctx, coarse, rgb, train, grad_on, bn_buffers, sinks, *params):
    assert len(params) == 30
    coarse, rgb = f32c(coarse), f32c(rgb)
    params = [f32c(p) for p in params]
    B, h, w = coarse.shape
    _, C, H, W = rgb.shape
    if C != 3 or rgb.shape[0] != B:
      raise RuntimeError("EdgeRefineFn: guidance image must be [B,3,H,W]")
    dev = rgb.device
    lib = nat.load()
    need_bwd = grad_on and any(ctx.needs_input_grad)   # grad_on: the caller's grad mode (forward() itself always runs
    # with grad disabled, and needs_input_grad stays True under torch.no_grad())
    gain = W / w                                     # float ratio (stereo_net.py:113)
    g = Pcl(B, 1, H, W, 0, REFINE_HALO, REFINE_HALO)
    g4 = Pcl(B, 1, H, W, 0, 1, 1)
    s33 = conv_shape_2d(1)

    up = torch.empty(B, 1, H, W, dtype=torch.float32, device=dev)
    call("as_upsample_bilinear_fwd", ptr(coarse), B, h, w, ptr(up), H, W, float(gain), stream())
    in4 = POOL.get(g4, dev, channels=4)
    call("as_pack_in4", ptr(up), ptr(rgb), 3, ptr(in4), g4, stream())

    # conv2d_feature: 4 -> 32, BatchNorm, LeakyReLU
    w0, b0, gamma0, beta0 = params[0:4]
    rm0, rv0 = bn_buffers[0]
    wp4 = pack_special(w0, PACK_CONV4 + 4, 9, 9 * 128,
                       lambda w_, o_: call("as_conv4_pack_weights", ptr(w_), 4, ptr(o_), s33, stream()))
    if train:
      stats = StatParts(lib.as_conv4_stat_parts(g4, g, s33), dev)
      z0 = POOL.get(g, dev)
      call("as_conv4_fwd", ptr(in4), g4, ptr(wp4), ptr(b0), ptr(z0), g, s33, 0, None, None, LEAKY_SLOPE,
           ptr(stats.mean), ptr(stats.m2), ptr(stats.cnt), stream())
      st0 = bn_train_stats(stats, gamma0, beta0, rm0, rv0)
      fused_act = _FWD_ACT and need_bwd and all(lib.as_conv32_act_ok(g, g, conv_shape_2d(d)) == 1 for d in REFINE_DILATIONS)
      a0 = None if fused_act else bn_act(z0, st0, g)
    else:
      fused_act = False
      st0 = bn_eval_stats(gamma0, beta0, rm0, rv0)
      if need_bwd:
        z0 = POOL.get(g, dev)
        call("as_conv4_fwd", ptr(in4), g4, ptr(wp4), ptr(b0), ptr(z0), g, s33, 0, None, None, LEAKY_SLOPE, None, None,
             None, stream())
        a0 = bn_act(z0, st0, g)
      else:
        z0 = None
        a0 = POOL.get(g, dev)
        call("as_conv4_fwd", ptr(in4), g4, ptr(wp4), ptr(b0), ptr(a0), g, s33, 1, ptr(st0.scale), ptr(st0.shift),
             LEAKY_SLOPE, None, None, None, stream())
    if not need_bwd and z0 is not None:
      POOL.put(z0, g); z0 = None

    xs, zs, sts = [a0], [], []
    tail_fused = None
    if fused_act:
      # every layer forms its own operand: layer l reads z_{l-1} (+ a_{l-2}), leaves a_{l-1} behind as a by-product and
      # writes z_l; only the last layer's output needs a BatchNorm + LeakyReLU pass of its own
      xs = []
      z_prev, st_prev = z0, st0
      for l, dil in enumerate(REFINE_DILATIONS):
        wl, bl, gamma, beta = params[4 + 4 * l:8 + 4 * l]
        rm, rv = bn_buffers[1 + l]
        a_prev, z, st = block_forward_act(z_prev, xs[-1] if xs else None, st_prev, g, conv_shape_2d(dil), wl, bl, gamma, beta,
                                          rm, rv)
        xs.append(a_prev); zs.append(z); sts.append(st)
        z_prev, st_prev = z, st
      if _REFINE_OUT and lib.as_refine_out_ok(g) == 1:
        tail_fused = (z_prev, xs[-1], st_prev)            # the last block's activation rides on the output layer (below)
      else:
        xs.append(bn_act(z_prev, st_prev, g, residual=xs[-1]))
    else:
      for l, dil in enumerate(REFINE_DILATIONS):
        wl, bl, gamma, beta = params[4 + 4 * l:8 + 4 * l]
        rm, rv = bn_buffers[1 + l]
        z, a, st = block_forward(xs[-1], g, conv_shape_2d(dil), wl, bl, gamma, beta, rm, rv, train, True, need_bwd)
        zs.append(z); sts.append(st); xs.append(a)
        if not need_bwd:
          POOL.put(xs[-2], g)

    w_out, b_out = params[28], params[29]
    out = torch.empty(B, 1, H, W, dtype=torch.float32, device=dev)
    if tail_fused is not None:
      # a6 = lrelu(BN(z6)) + a5 formed while conv2d_out stages its rows, written once for the backward pass (csrc/refine_out.hip)
      z6, a5, st6 = tail_fused
      a6 = POOL.get(g, dev)
      call("as_refine_out_fwd", ptr(z6), ptr(a5), ptr(st6.scale), ptr(st6.shift), LEAKY_SLOPE, ptr(a6), g, ptr(w_out), ptr(b_out),
           ptr(up), 1, ptr(out), stream())
      xs.append(a6)
    elif _REFINE_OUT and lib.as_refine_out_ok(g) == 1:
      call("as_refine_out_fwd", ptr(xs[-1]), None, None, None, LEAKY_SLOPE, None, g, ptr(w_out), ptr(b_out), ptr(up), 1, ptr(out),
           stream())
    else:
      call("as_conv32to1_fwd", ptr(xs[-1]), g, s33, ptr(w_out), ptr(b_out), ptr(up), 1, ptr(out), stream())

    if need_bwd:
      ctx.g, ctx.g4, ctx.train = g, g4, bool(train)
      ctx.sinks = sinks
      ctx.dims = (B, h, w, H, W, float(gain))
      ctx.in4, ctx.z0, ctx.st0 = in4, z0, st0
      ctx.xs, ctx.zs, ctx.sts = xs, zs, sts
      ctx.save_for_backward(out, *params)
    else:
      POOL.put(xs[-1], g); POOL.put(in4, g4, channels=4)
    return out

  @staticmethod
  def backward(ctx, g_out):
    out, *params = ctx.saved_tensors
    g, g4 = ctx.g, ctx.g4
    B, h, w, H, W, gain = ctx.dims
    dev = out.device
    lib = nat.load()
    s33 = conv_shape_2d(1)
    xs, zs, sts = ctx.xs, ctx.zs, ctx.sts
    grads = [None] * 30

    g_out = f32c(g_out)
    g_pre = torch.empty_like(out)                             # through the final ReLU (stereo_net.py:121)
    call("as_relu_bwd", ptr(g_out), ptr(out), out.numel(), ptr(g_pre), stream())
    w_out = params[28]
    sinks = ctx.sinks
    g_a = POOL.get(g, dev)
    ws = _empty(lib.as_conv32to1_bwd_workspace(g, s33), dev)
    sums = None
    if _sink(sinks, 28) is not None and _sink(sinks, 29) is not None:
      if _TAIL_BNSUMS and ctx.train and _BN_SYNC is None and lib.as_conv32to1_bnsums_ok(g, s33) == 1:
        # the data gradient also leaves stage 1 of the last block's BatchNorm backward behind (its own pass otherwise)
        nws = _empty(lib.as_bn_bwd_workspace(g), dev)
        # the weight gradient (reads the last activation and g_pre, adds into its sinks) beside the data gradient
        ev = fork_beside(lambda: call("as_conv32to1_bwd", ptr(g_pre), ptr(xs[6]), g, s33, ptr(w_out), None, ptr(sinks[28]),
                                      ptr(sinks[29]), 1, ptr(ws), stream()))
        call("as_conv32to1_dgrad_bnsums", ptr(g_pre), g, s33, ptr(w_out), ptr(g_a), ptr(zs[5]), ptr(sts[5].scale),
             ptr(sts[5].shift), ptr(sts[5].mean), LEAKY_SLOPE, ptr(nws), stream())
        join_beside(ev)
        sums = BnBwdSums(nws, lib.as_conv32to1_bnsums_parts(g))
      else:
        call("as_conv32to1_bwd", ptr(g_pre), ptr(xs[6]), g, s33, ptr(w_out), ptr(g_a), ptr(sinks[28]), ptr(sinks[29]), 1,
             ptr(ws), stream())
    else:
      g_wout, g_bout = torch.empty_like(w_out), _empty(1, dev)
      call("as_conv32to1_bwd", ptr(g_pre), ptr(xs[6]), g, s33, ptr(w_out), ptr(g_a), ptr(g_wout), ptr(g_bout), 0,
           ptr(ws), stream())
      grads[28], grads[29] = g_wout, g_bout

    for l in range(5, -1, -1):
      wl, bl, gamma, beta = params[4 + 4 * l:8 + 4 * l]
      # the data gradient of block l is the output gradient of block l-1 (or of conv2d_feature): stage 1 of that
      # layer's BatchNorm backward rides on the data-gradient kernel
      next_bn = (zs[l - 1], sts[l - 1]) if l > 0 else (ctx.z0, ctx.st0)
      g_x, dW, db, g_gamma, g_beta, sums = block_backward(g_a, xs[l], zs[l], sts[l], wl, gamma, g,
                                                          conv_shape_2d(REFINE_DILATIONS[l]), ctx.train, True, True,
                                                          sinks[4 + 4 * l:8 + 4 * l] if sinks is not None else None,
                                                          sums, next_bn)
      grads[4 + 4 * l:8 + 4 * l] = [dW, db, g_gamma, g_beta]
      POOL.put(g_a, g); POOL.put(zs[l], g); POOL.put(xs[l + 1], g)
      g_a = g_x

    # conv2d_feature backward
    w0, b0, gamma0, beta0 = params[0:4]
    ws4 = _empty(lib.as_conv4_wgrad_workspace(g, s33), dev)
    all_sinks = all(_sink(sinks, i) is not None for i in range(4))
    if all_sinks and lib.as_conv4_wgrad_bnapply_ok(g4, g, s33) == 1:
      # stages 1-2 of the BatchNorm backward (stage 1 usually already in `sums`); stage 3 rides on the weight gradient
      bws = sums.workspace if sums is not None else _empty(lib.as_bn_bwd_workspace(g), dev)
      st0 = ctx.st0
      bn_bwd_coefs(g_a, ctx.z0, st0, gamma0, g, ctx.train, sinks[2], sinks[3], True, bws, sums)
      coef = bws[lib.as_bn_bwd_coef_offset():]
      if _HEAD_PROJ and ctx.needs_input_grad[0] and not ctx.needs_input_grad[1]:
        # g_z0 itself is needed by nobody: only its 3x3 32->1 data gradient towards the disparity channel is.  The weight
        # gradient kernel writes the nine per-tap projections of g_z0 (36 B per pixel instead of 128) and a gather sums
        # the nine shifted planes — no g_z0 write, no 32-channel re-read
        w_proj = pack_special(w0, PACK_MIRROR_TAP, 9, 288,                      # [9][32]: channel 0 of w0, taps mirrored
                              lambda w_, o_: call("as_mirror_taps_ch0", ptr(w_), 4, ptr(o_), None, stream()))
        h_proj = _empty(B * 9 * H * W, dev)
        call("as_conv4_wgrad_bnapply_proj", ptr(ctx.in4), g4, ptr(g_a), ptr(ctx.z0), g, s33, 4, ptr(st0.scale), ptr(st0.shift),
             ptr(st0.mean), ptr(coef), LEAKY_SLOPE, ptr(w_proj), ptr(h_proj), ptr(sinks[0]), ptr(sinks[1]), 1, ptr(ws4),
             stream())
        g_z0 = None
      else:
        h_proj = None
        g_z0 = POOL.get(g, dev)
        call("as_conv4_wgrad_bnapply", ptr(ctx.in4), g4, ptr(g_a), ptr(ctx.z0), g, s33, 4, ptr(st0.scale), ptr(st0.shift),
             ptr(st0.mean), ptr(coef), LEAKY_SLOPE, ptr(g_z0), ptr(sinks[0]), ptr(sinks[1]), 1, ptr(ws4), stream())
      dW0 = db0 = g_gamma0 = g_beta0 = None
    else:
      h_proj = None
      g_z0, g_gamma0, g_beta0 = bn_act_bwd(g_a, ctx.z0, ctx.st0, gamma0, g, ctx.train, _sink(sinks, 2), _sink(sinks, 3), sums)
      if _sink(sinks, 0) is not None and _sink(sinks, 1) is not None:
        call("as_conv4_wgrad", ptr(ctx.in4), g4, ptr(g_z0), g, s33, 4, ptr(sinks[0]), ptr(sinks[1]), 1, ptr(ws4), stream())
        dW0 = db0 = None
      else:
        dW0 = torch.empty_like(w0); db0 = _empty(32, dev)
        call("as_conv4_wgrad", ptr(ctx.in4), g4, ptr(g_z0), g, s33, 4, ptr(dW0), ptr(db0), 0, ptr(ws4), stream())
    grads[0:4] = [dW0, db0, g_gamma0, g_beta0]

    g_coarse = None
    if ctx.needs_input_grad[0]:
      # d/d(up-sampled disparity) = direct path (g_pre) + conv2d_feature's data gradient for input
      # channel 0, which is a 32->1 convolution of g_z0 with mirrored taps; the add is fused.
      g_up = torch.empty(B, 1, H, W, dtype=torch.float32, device=dev)
      if h_proj is not None:
        call("as_tap_gather", ptr(h_proj), ptr(g_pre), ptr(g_up), B, H, W, stream())
      else:
        w_ch0 = pack_special(w0, PACK_MIRROR_CH, 9, 288,                        # [32][9]
                             lambda w_, o_: call("as_mirror_taps_ch0", ptr(w_), 4, None, ptr(o_), stream()))
        call("as_conv32to1_fwd", ptr(g_z0), g, s33, ptr(w_ch0), None, ptr(g_pre), 0, ptr(g_up), stream())
      g_coarse = torch.empty(B, h, w, dtype=torch.float32, device=dev)
      call("as_upsample_bilinear_bwd", ptr(g_up), B, H, W, ptr(g_coarse), h, w, gain, stream())
    if ctx.needs_input_grad[1]:
      raise NotImplementedError("EdgeRefineFn: gradient w.r.t. the guidance image is not part of the adaptation path")
    POOL.put(g_z0, g); POOL.put(g_a, g); POOL.put(ctx.z0, g); POOL.put(xs[0], g); POOL.put(ctx.in4, g4, channels=4)
    ctx.xs = ctx.zs = ctx.sts = ctx.in4 = ctx.z0 = None
    return (g_coarse, None, None, None, None, None) + tuple(grads)


# ----------------------------------------------------------------------------------------
# a6: bilinear up-sampling (stereo_net.py:106-114, 201-202)
# ----------------------------------------------------------------------------------------
class UpsampleBilinearFn(torch.autograd.Function):
  """src [B,h,w] -> [B,1,H,W] * gain, align_corners=False."""

  @staticmethod
  def forward(ctx, src, H, W, gain):
    src = f32c(src)
    B, h, w = src.shape
    dst = torch.empty(B, 1, H, W, dtype=torch.float32, device=src.device)
    call("as_upsample_bilinear_fwd", ptr(src), B, h, w, ptr(dst), int(H), int(W), float(gain), stream())
    ctx.dims = (B, h, w, int(H), int(W), float(gain))
    return dst

  @staticmethod
  def backward(ctx, g_dst):
    B, h, w, H, W, gain = ctx.dims
    g_dst = f32c(g_dst)
    g_src = torch.empty(B, h, w, dtype=torch.float32, device=g_dst.device)
    call("as_upsample_bilinear_bwd", ptr(g_dst), B, H, W, ptr(g_src), h, w, gain, stream())
    return g_src, None, None, None


# ----------------------------------------------------------------------------------------
# a9: LinearWarping (models/linear_warping.py:18-57)
# ----------------------------------------------------------------------------------------
class LinearWarpFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, img, disp, right_to_left):
    img, disp = f32c(img), f32c(disp)
    B, C, H, W = img.shape
    if tuple(disp.shape) != (B, 1, H, W):
      raise RuntimeError("LinearWarpFn: disparity must be [B,1,H,W]")
    warped = torch.empty_like(img)
    mask = torch.empty(B, 1, H, W, dtype=torch.uint8, device=img.device)
    call("as_warp_fwd", ptr(img), ptr(disp), B, C, H, W, int(bool(right_to_left)), ptr(warped), ptr(mask), stream())
    ctx.save_for_backward(img, disp)
    ctx.r2l = int(bool(right_to_left))
    mask = mask.bool()
    ctx.mark_non_differentiable(mask)
    ctx.set_materialize_grads(False)
    return warped, mask

  @staticmethod
  def backward(ctx, g_warped, _g_mask):
    img, disp = ctx.saved_tensors
    if ctx.needs_input_grad[0]:
      raise NotImplementedError("LinearWarpFn: gradient w.r.t. the image is not part of the adaptation path")
    B, C, H, W = img.shape
    if g_warped is None:
      return None, None, None
    g_disp = torch.empty_like(disp)
    g_warped = f32c(g_warped)
    call("as_warp_bwd", ptr(g_warped), ptr(img), ptr(disp), B, C, H, W, ctx.r2l, ptr(g_disp), stream())
    return None, g_disp, None


class LinearWarpNearestFn(torch.autograd.Function):
  """LinearWarping.forward(mode="nearest"): the nearest tap instead of the bilinear blend.  The output is piecewise constant
  in the disparity: autograd's gradient w.r.t. it is zero (what F.grid_sample returns for the grid in this mode)."""

  @staticmethod
  def forward(ctx, img, disp, right_to_left):
    img, disp = f32c(img), f32c(disp)
    B, C, H, W = img.shape
    if tuple(disp.shape) != (B, 1, H, W):
      raise RuntimeError("LinearWarpNearestFn: disparity must be [B,1,H,W]")
    warped = torch.empty_like(img)
    mask = torch.empty(B, 1, H, W, dtype=torch.uint8, device=img.device)
    call("as_warp_nearest_fwd", ptr(img), ptr(disp), B, C, H, W, int(bool(right_to_left)), ptr(warped), ptr(mask), stream())
    ctx.disp_shape = disp.shape
    mask = mask.bool()
    ctx.mark_non_differentiable(mask)
    return warped, mask

  @staticmethod
  def backward(ctx, g_warped, _g_mask):
    if ctx.needs_input_grad[0]:
      raise NotImplementedError("LinearWarpNearestFn: gradient w.r.t. the image is not part of the adaptation path")
    return None, torch.zeros(ctx.disp_shape, dtype=g_warped.dtype, device=g_warped.device), None


# ----------------------------------------------------------------------------------------
# a10: monodepth photometric loss (utils/loss_functions.py:106-138)
# ----------------------------------------------------------------------------------------
class MonodepthLossFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, pred, img, warped, smoothness_weight):
    pred, img, warped = f32c(pred), f32c(img), f32c(warped)
    B, C, H, W = img.shape
    if C != 3 or tuple(pred.shape) != (B, 1, H, W) or warped.shape != img.shape:
      raise RuntimeError("MonodepthLossFn: expected pred [B,1,H,W], img/warped [B,3,H,W]")
    dev = img.device
    outs = [torch.empty(B, 1, H, W, dtype=torch.float32, device=dev) for _ in range(4)]
    ws = _empty(nat.load().as_monodepth_workspace(B, H, W), dev)
    call("as_monodepth_loss_fwd", ptr(pred), ptr(img), ptr(warped), B, H, W, float(smoothness_weight),
         ptr(outs[0]), ptr(outs[1]), ptr(outs[2]), ptr(outs[3]), ptr(ws), stream())
    ctx.save_for_backward(pred, img, warped)
    ctx.sw = float(smoothness_weight)
    ctx.set_materialize_grads(False)       # adapt.py:81 uses the first map only: the other three gradients stay None (the
    return tuple(outs)                     # kernels take NULL for them) instead of three zero-filled planes per step

  @staticmethod
  def backward(ctx, g_total, g_l1, g_ssim, g_smooth):
    pred, img, warped = ctx.saved_tensors
    if ctx.needs_input_grad[1]:
      raise NotImplementedError("MonodepthLossFn: gradient w.r.t. the true image is not part of the adaptation path")
    B, C, H, W = img.shape
    dev = img.device
    g_pred = torch.empty_like(pred) if ctx.needs_input_grad[0] else None
    g_warped = torch.empty_like(warped) if ctx.needs_input_grad[2] else None
    ws = _empty(nat.load().as_monodepth_workspace(B, H, W), dev)
    if g_total is None and g_l1 is None and g_ssim is None and g_smooth is None:
      return None, None, None, None
    g_total, g_l1, g_ssim, g_smooth = f32c(g_total), f32c(g_l1), f32c(g_ssim), f32c(g_smooth)
    call("as_monodepth_loss_bwd", ptr(g_total), ptr(g_l1), ptr(g_ssim), ptr(g_smooth),
         ptr(pred), ptr(img), ptr(warped), B, H, W, ctx.sw, ptr(g_pred), ptr(g_warped), ptr(ws), stream())
    return g_pred, None, g_warped, None


# ----------------------------------------------------------------------------------------
# The adaptation step's whole loss tail in one Function (adapt.py:78-86: monodepth_single_loss): warp the right image with the
# predicted disparity, photometric + smoothness loss map, mean over the valid pixels.  One row-walking pass each way
# (csrc/photometric_rows.hip) that gives the bits of LinearWarpFn -> MonodepthLossFn -> MaskedMeanFn: no loss map, no dense
# gradient map mask * (g / N), no coefficient planes and no gradient of the warped image ever reach HBM.
# ----------------------------------------------------------------------------------------
class MaskedPhotometricFn(torch.autograd.Function):
  """pred [B,1,H,W], left, right [B,3,H,W] -> (mean, sum, count, warped, valid mask uint8 [B,1,H,W]).
  mean = total[mask].mean(), sum = total[mask].sum() (data-parallel ranks back-propagate the sum); only mean and sum are
  differentiable, and only w.r.t. pred."""

  @staticmethod
  def forward(ctx, pred, left, right, smoothness_weight):
    pred, left, right = f32c(pred), f32c(left), f32c(right)
    B, C, H, W = left.shape
    if C != 3 or tuple(pred.shape) != (B, 1, H, W) or right.shape != left.shape:
      raise RuntimeError("MaskedPhotometricFn: expected pred [B,1,H,W], left/right [B,3,H,W]")
    dev = left.device
    lib = nat.load()
    warped = torch.empty_like(right)
    mask = torch.empty(B, 1, H, W, dtype=torch.uint8, device=dev)
    out3 = _empty(4, dev)                          # sum, count, mean, count (a second copy: the non-differentiable output)
    ws = _empty(lib.as_photometric_chain_workspace(B, H, W), dev)
    call("as_photometric_chain_fwd", ptr(pred), ptr(left), ptr(right), B, H, W, float(smoothness_weight), ptr(warped), ptr(mask),
         ptr(out3), ptr(ws), stream())
    ctx.save_for_backward(pred, left, right, out3)
    ctx.sw = float(smoothness_weight)
    ctx.fwd_ws = ws                          # the forward workspace: its per-image mean disparity is reused by backward
    count = out3[3]
    ctx.mark_non_differentiable(count, warped, mask)
    ctx.set_materialize_grads(False)
    return out3[2], out3[0], count, warped, mask

  @staticmethod
  def backward(ctx, g_mean, g_sum, _g_count, _g_warped, _g_mask):
    pred, left, right, out3 = ctx.saved_tensors
    if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
      raise NotImplementedError("MaskedPhotometricFn: gradients w.r.t. the images are not part of the adaptation path")
    if g_mean is None and g_sum is None:
      return None, None, None, None
    B, C, H, W = left.shape
    dev = left.device
    g_mean = f32c(g_mean).reshape(1) if g_mean is not None else None
    g_sum = f32c(g_sum).reshape(1) if g_sum is not None else None
    g_pred = torch.empty_like(pred)
    ws = _empty(nat.load().as_photometric_chain_workspace(B, H, W), dev)
    call("as_photometric_chain_bwd", ptr(g_sum), ptr(g_mean), ptr(out3), ptr(pred), ptr(left), ptr(right), B, H, W, ctx.sw,
         ptr(g_pred), ptr(ws), ptr(ctx.fwd_ws), stream())
    ctx.fwd_ws = None
    return g_pred, None, None, None


# ----------------------------------------------------------------------------------------
# loss[mask].mean() without the boolean-index host sync (adapt.py:81-83)
# ----------------------------------------------------------------------------------------
class MaskedMeanFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, values, mask):
    values = f32c(values)
    m8 = mask.to(torch.uint8).contiguous()
    n = values.numel()
    out2 = _empty(2, values.device)
    ws = _empty(nat.load().as_masked_sum_workspace(n), values.device)
    call("as_masked_sum", ptr(values), ptr(m8), n, ptr(out2), ptr(ws), stream())
    ctx.save_for_backward(m8, out2)
    ctx.shape = values.shape
    return out2[0] / out2[1]

  @staticmethod
  def backward(ctx, g):
    m8, out2 = ctx.saved_tensors
    return (m8.to(torch.float32) * (g / out2[1])).view(ctx.shape), None


def masked_mean(values, mask):
  return MaskedMeanFn.apply(values, mask)


# ----------------------------------------------------------------------------------------
# a13: khamis_robust_loss (utils/loss_functions.py:6-15), the supervised term of the ER modes (adapt.py:339-349)
# ----------------------------------------------------------------------------------------
class KhamisLossFn(torch.autograd.Function):
  """pred, gt (same shape) -> scalar sum_{gt>0}(sqrt((gt-pred)^2+4)/2 - 1) / max(count(gt>0), 1)."""

  @staticmethod
  def forward(ctx, pred, gt):
    pred, gt = f32c(pred), f32c(gt)
    if pred.shape != gt.shape:
      raise RuntimeError("KhamisLossFn: pred %s and gt %s differ in shape" % (tuple(pred.shape), tuple(gt.shape)))
    n = pred.numel()
    out2 = _empty(2, pred.device)
    ws = _empty(nat.load().as_khamis_workspace(n), pred.device)
    call("as_khamis_fwd", ptr(pred), ptr(gt), n, ptr(out2), ptr(ws), stream())
    ctx.save_for_backward(pred, gt, out2)
    return out2[0].clone()

  @staticmethod
  def backward(ctx, g):
    pred, gt, out2 = ctx.saved_tensors
    if ctx.needs_input_grad[1]:
      raise NotImplementedError("KhamisLossFn: gradient w.r.t. the ground truth is not part of any path")
    g = f32c(g).reshape(1)
    g_pred = torch.empty_like(pred)
    call("as_khamis_bwd", ptr(pred), ptr(gt), ptr(g), ptr(out2), pred.numel(), ptr(g_pred), stream())
    return g_pred, None
