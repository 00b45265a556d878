"""RCCL driven directly: a communicator of this library's own, through ctypes on the librccl.so that PyTorch-ROCm ships.

Why not ``torch.distributed`` for the collectives of the step itself: on this build (PyTorch 2.10 / ROCm 7.x) a c10d "nccl"
collective cannot be issued while a hipGraph capture is open — ProcessGroupNCCL's watchdog thread polls the work's end
event, the event belongs to a capturing stream, the poll fails with ``hipErrorCapturedEvent`` and the watchdog takes the
process down (tests/tools/rccl_capture_probe.py).  RCCL itself captures fine (tests/tools/rccl_native_probe.py): with its
own communicator the data-parallel adaptation step — forward, backward, the gradient all-reduce over xGMI, clip, Adam — is
ONE hipGraph, and the small collectives of cross-replica BatchNorm become graph nodes as well.

``torch.distributed`` still does what it is good at: rendezvous (the unique id travels through the process group) and the
barrier / max-over-ranks timing of bench.py.  Collectives are enqueued on the CURRENT torch stream, in place, no host sync.
"""
import contextlib
import ctypes
import glob
import os
import sys
import threading

import torch
import torch.distributed as dist

_DTYPES = {torch.float32: 7, torch.float64: 8, torch.int32: 2, torch.int64: 4, torch.uint8: 1}
_SUM = 0


class _UniqueId(ctypes.Structure):
  _fields_ = [("internal", ctypes.c_char * 128)]


_lib = None


def _load():
  global _lib
  if _lib is None:
    libs = sorted(glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*")))
    if not libs:
      raise RuntimeError("adaptive_stereo.rccl: no librccl.so next to torch")
    lib = ctypes.CDLL(libs[0])
    vp, ci, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(vp), ci, _UniqueId, ci]
    lib.ncclAllReduce.argtypes = [vp, vp, sz, ci, ci, vp, vp]
    lib.ncclAllGather.argtypes = [vp, vp, sz, ci, vp, vp]
    lib.ncclCommDestroy.argtypes = [vp]
    lib.ncclCommAbort.argtypes = [vp]
    lib.ncclGetVersion.argtypes = [ctypes.POINTER(ci)]
    lib.ncclGetErrorString.restype = ctypes.c_char_p
    lib.ncclGetErrorString.argtypes = [ci]
    _lib = lib
  return _lib


_FD_LOCK = threading.Lock()


@contextlib.contextmanager
def _c_stdout_to_stderr():
  """librccl prints a version banner (five lines) to the C stdout when a communicator is created; a library must not write to
  its host's stdout (bench.py's contract is ONE JSON line there): file descriptor 1 points at stderr for the duration.
  The swap is process-wide — another thread's stdout writes go to stderr meanwhile — so it is serialised by a lock (two
  adapters built concurrently cannot restore each other's descriptor) and an application that owns its stdout already
  (bench.py's claim_stdout) or does not mind the banner turns it off with AS_RCCL_KEEP_STDOUT=1."""
  if os.environ.get("AS_RCCL_KEEP_STDOUT"):
    yield
    return
  with _FD_LOCK:
    libc = ctypes.CDLL(None)
    try:
      sys.stdout.flush()
    except Exception:
      pass
    libc.fflush(None)
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
      yield
    finally:
      libc.fflush(None)
      os.dup2(saved, 1)
      os.close(saved)


def _check(rc, what):
  if rc != 0:
    raise RuntimeError("RCCL %s failed: %s" % (what, _load().ncclGetErrorString(rc).decode()))


class RcclComm(object):
  """One RCCL communicator over the ranks of ``group``, on the CURRENT HIP device.  Built by ``try_create`` (the staged,
  agreed-on construction below); the constructor only wraps an initialised ``ncclComm_t``."""

  def __init__(self, handle, group, rank, world):
    self._comm, self.group, self.rank, self.world = handle, group, rank, world
    self.device = torch.cuda.current_device()
    self.evidence = None        # filled by the probe stage: what the first collective of this communicator returned

  def describe(self):
    """Rank-count evidence for logs and bench.py's JSON line: the result of the probe all-reduce (a sum of ones = the number of
    ranks that took part), the HIP device of every rank as all-gathered through THIS communicator, the library's version."""
    return dict(self.evidence or {}, world=self.world, rank=self.rank)

  def _stream(self):
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

  @staticmethod
  def _ok(t):
    if not (t.is_cuda and t.is_contiguous() and t.dtype in _DTYPES):
      raise RuntimeError("adaptive_stereo.rccl: contiguous GPU tensors of %s only (got %s, %s)" % (
          sorted(str(k) for k in _DTYPES), t.dtype, t.device))

  def all_reduce(self, t):
    """In-place sum over the ranks, enqueued on the current stream (capturable)."""
    self._ok(t)
    _check(_load().ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), _DTYPES[t.dtype], _SUM, self._comm, self._stream()),
           "ncclAllReduce")
    return t

  def all_gather(self, out, inp):
    """out [world * inp.numel()] (rank-major) <- every rank's inp, enqueued on the current stream (capturable)."""
    self._ok(out); self._ok(inp)
    if out.numel() != self.world * inp.numel() or out.dtype != inp.dtype:
      raise RuntimeError("adaptive_stereo.rccl: all_gather needs out.numel() == world * inp.numel() of one dtype")
    _check(_load().ncclAllGather(inp.data_ptr(), out.data_ptr(), inp.numel(), _DTYPES[inp.dtype], self._comm, self._stream()),
           "ncclAllGather")
    return out

  def destroy(self):
    """Orderly release (a collective: every rank calls it, with the communicator's work done)."""
    if self._comm is not None and self._comm.value:
      _load().ncclCommDestroy(self._comm)
    self._comm = None

  def abort(self):
    """Non-blocking teardown for paths on which the peers may be gone or at another point of the program (garbage collection,
    interpreter exit after a crash elsewhere): ncclCommAbort does not wait for outstanding work or for the other ranks."""
    if self._comm is not None and self._comm.value:
      _load().ncclCommAbort(self._comm)
    self._comm = None


class _RcclStages(object):
  """What the staged construction does at each stage, for real.  (tests/test_distributed_cpu.py drives the same protocol over
  a gloo group with stand-ins for these five methods.)"""
  agree_device = "cuda"

  def load(self):
    _load()

  def unique_id(self):
    uid = _UniqueId()
    _check(_load().ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
    return ctypes.string_at(ctypes.addressof(uid), 128)     # (all 128 bytes: the c_char array field would stop at a NUL)

  def prepare(self):
    return torch.ones(8, dtype=torch.float32, device="cuda")    # a live HIP context on this device + the probe's buffer

  def agree_flag(self):
    return torch.zeros(1, dtype=torch.int32, device="cuda")     # the agreements' one buffer, allocated before the first of them

  def init(self, uid_bytes, group, rank, world):
    uid = _UniqueId()
    ctypes.memmove(ctypes.addressof(uid), uid_bytes, 128)
    comm = ctypes.c_void_p()
    with _c_stdout_to_stderr():
      _check(_load().ncclCommInitRank(ctypes.byref(comm), world, uid, rank), "ncclCommInitRank")
    return RcclComm(comm, group, rank, world)

  def probe(self, comm, probe):
    devs = torch.full((comm.world,), -1, dtype=torch.int32, device="cuda")
    mine = torch.tensor([torch.cuda.current_device()], dtype=torch.int32, device="cuda")
    with _c_stdout_to_stderr():
      comm.all_reduce(probe)
      comm.all_gather(devs, mine)
      torch.cuda.synchronize()
    ver = ctypes.c_int(0)
    _load().ncclGetVersion(ctypes.byref(ver))
    comm.evidence = {"ranks": float(probe[0]), "device_ids": [int(v) for v in devs.cpu()], "nccl_version": int(ver.value),
                     "how": "ranks = the communicator's first all-reduce of ones; device_ids = an all-gather of each rank's HIP device"}
    if float(probe[0]) != float(comm.world):
      raise RuntimeError("probe all-reduce returned %r over %d ranks" % (float(probe[0]), comm.world))


last_error = None          # why the most recent try_create() on this rank gave up (for logs and tests)


def _injected_failure(stage, rank):
  """AS_RCCL_FAIL_AT="stage:rank" (stages: load, unique_id, receive, prepare, init, probe) makes that stage fail on that rank
  — how the tests walk every exit of the protocol.  The local stages fail before they run; the two COLLECTIVE stages (init =
  ncclCommInitRank, probe = an all-reduce of the new communicator) fail after the collective has returned, i.e. the case
  "the call came back with an error on this rank": a rank that never ENTERS a collective the others are in is beyond what
  any agreement can repair, which is why everything fallible and local (library, id, device context, probe buffer) is done
  and agreed on before the first of them."""
  spec = os.environ.get("AS_RCCL_FAIL_AT")
  if spec and spec == "%s:%d" % (stage, rank):
    raise RuntimeError("injected failure at stage %s on rank %d (AS_RCCL_FAIL_AT)" % (stage, rank))


def _all_agree(ok, group, flag):
  """MIN over the ranks of "I am fine", in the buffer every rank allocated BEFORE the first agreement (no allocation, hence
  nothing that can fail locally, stands between a rank and a collective its peers are entering)."""
  flag.fill_(1 if ok else 0)
  dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
  return int(flag) == 1


def _create_staged(group, stages):
  """The construction protocol: every rank runs the SAME sequence of process-group collectives whatever fails locally —
    load | agree | rank 0 draws the unique id (None on failure) | broadcast | agree on "id received" | HIP context | agree |
    ncclCommInitRank | agree | probe all-reduce | agree
  — so a rank that fails at any stage never leaves the others inside a collective it does not enter itself: a failure is
  carried to the next agreement and all ranks leave together with None.  (Round 3 raised out of the constructor when
  ncclGetUniqueId failed on rank 0, i.e. BEFORE the broadcast the other ranks were already waiting in.)
  The agreements' flag buffer is allocated FIRST, before any collective (round 4 allocated it inside the first agreement,
  outside any try: a rank whose HIP context could not be created raised there while its peers sat in the all-reduce).  The one
  residual case: a rank that cannot allocate those 4 bytes on its device cannot take part in ANY collective of an nccl group,
  the agreement included — it raises, and its peers leave through the process group's timeout; nothing in a library can do
  better for a rank without a device."""
  global last_error
  rank, world = dist.get_rank(group), dist.get_world_size(group)
  err = None
  _injected_failure("agree_tensor", rank)      # (tests: the residual case raises out of try_create, before any collective)
  flag = stages.agree_flag()

  def attempt(stage, fn, collective=False):
    nonlocal err
    if err is not None:
      return None
    try:
      if not collective:
        _injected_failure(stage, rank)
      out = fn()
      if collective:
        _injected_failure(stage, rank)
      return out
    except Exception as e:                 # noqa: BLE001 — whatever fails, the protocol goes on to the next agreement
      err = "%s: %r" % (stage, e)
      return None

  def agreed():
    return _all_agree(err is None, group, flag)

  def give_up(comm=None):
    global last_error
    last_error = err if err is not None else "another rank could not build its communicator"
    if comm is not None:
      try:
        comm.destroy()
      except Exception:                    # noqa: BLE001
        pass
    return None

  attempt("load", stages.load)
  if not agreed():
    return give_up()
  uid = attempt("unique_id", stages.unique_id) if rank == 0 else None
  box = [uid if rank == 0 else None]       # None travels when rank 0 could not draw an id: the broadcast is ALWAYS reached
  src = dist.get_global_rank(group, 0) if group is not None else 0
  dist.broadcast_object_list(box, src=src, group=group)

  def received():
    if not isinstance(box[0], (bytes, bytearray)) or len(box[0]) != 128:
      raise RuntimeError("no unique id arrived from rank 0 (%r)" % (type(box[0]).__name__,))
    return bytes(box[0])
  uid = attempt("receive", received)
  if not agreed():
    return give_up()
  probe_buf = attempt("prepare", stages.prepare)
  if not agreed():                         # everyone holds the id, a device context and its probe buffer: only now is the
                                           # collective init entered
    return give_up()
  made = []

  def init():
    made.append(stages.init(uid, group, rank, world))
    return made[0]
  attempt("init", init, collective=True)
  comm = made[0] if made else None
  if not agreed():
    return give_up(comm)
  attempt("probe", lambda: stages.probe(comm, probe_buf), collective=True)
  if not agreed():
    return give_up(comm)
  return comm


def try_create(group=None):
  """A communicator for ``group`` if EVERY rank can build one, else None on every rank (the caller then keeps the c10d
  collectives, outside any capture).  See ``_create_staged`` for the protocol."""
  global last_error
  last_error = None
  if not torch.cuda.is_available() or dist.get_backend(group) != "nccl":
    last_error = "not an nccl process group on a GPU"
    return None
  return _create_staged(group, _RcclStages())
