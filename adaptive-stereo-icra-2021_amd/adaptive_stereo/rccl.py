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
    lib.ncclGetErrorString.restype = ctypes.c_char_p
    lib.ncclGetErrorString.argtypes = [ci]
    _lib = lib
  return _lib


@contextlib.contextmanager
def _c_stdout_to_stderr():
  """librccl prints a version banner (five lines) to the C stdout when a communicator is created; a library must not write to
  its host's stdout (bench.py's contract is ONE JSON line there): file descriptor 1 points at stderr for the duration."""
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
  """One RCCL communicator over the ranks of ``group`` (default: the world), created on the CURRENT HIP device.
  Construction is collective: every rank of the group must call it."""

  def __init__(self, group=None):
    lib = _load()
    self.group = group
    self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
    uid = _UniqueId()
    if self.rank == 0:
      _check(lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
    # (all 128 bytes: reading the c_char array field would stop at the first NUL)
    box = [ctypes.string_at(ctypes.addressof(uid), 128) if self.rank == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if len(box[0]) != 128:
      raise RuntimeError("adaptive_stereo.rccl: unique id of %d bytes" % len(box[0]))
    ctypes.memmove(ctypes.addressof(uid), box[0], 128)
    self.device = torch.cuda.current_device()
    torch.zeros(1, device="cuda")                                   # a live HIP context on this device
    comm = ctypes.c_void_p()
    with _c_stdout_to_stderr():
      _check(lib.ncclCommInitRank(ctypes.byref(comm), self.world, uid, self.rank), "ncclCommInitRank")
    self._comm = comm

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
    if self._comm is not None and self._comm.value:
      _load().ncclCommDestroy(self._comm)
    self._comm = None


last_error = None          # why the most recent try_create() on this rank gave up (for logs and tests)


def _all_agree(ok, group):
  flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
  dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
  return int(flag) == 1


def try_create(group=None):
  """A communicator for ``group`` if EVERY rank can build one, else None on every rank (the caller then keeps the c10d
  collectives, outside any capture).  Each stage is agreed on through the process group before the next collective call,
  so a rank that fails early never leaves the others waiting inside one."""
  global last_error
  last_error = None
  if not torch.cuda.is_available() or dist.get_backend(group) != "nccl":
    last_error = "not an nccl process group on a GPU"
    return None
  try:
    _load()
    ok = True
  except Exception as e:
    ok, last_error = False, repr(e)
  if not _all_agree(ok, group):
    return None
  comm = None
  try:
    comm = RcclComm(group)
  except Exception as e:
    ok, last_error = False, repr(e)
  if _all_agree(ok, group):
    try:
      probe = torch.ones(8, dtype=torch.float32, device="cuda")
      with _c_stdout_to_stderr():
        comm.all_reduce(probe)
        torch.cuda.synchronize()
      ok = float(probe[0]) == float(comm.world)
      if not ok:
        last_error = "probe all-reduce returned %r over %d ranks" % (float(probe[0]), comm.world)
    except Exception as e:
      ok, last_error = False, repr(e)
    if _all_agree(ok, group):
      return comm
  if last_error is None:
    last_error = "another rank could not build its communicator"
  if comm is not None:
    try:
      comm.destroy()
    except Exception:
      pass
  return None
