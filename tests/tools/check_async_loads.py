"""Static check of the hand-waited vector-memory loads in csrc/*.hip.

The kernels that issue `global_load_*` from inline asm and retire them with their own `s_waitcnt vmcnt(N)` tell hipcc nothing
about WHEN the destination registers become valid: the compiler may copy (or spill) them between the load and the wait — a read
of an in-flight destination, i.e. stale data that comes and goes with memory latency (this happened once: a v_mov of ten
operand registers in front of a branch that held two different waits).  This script compiles a source to ISA and reports every
instruction that reads or overwrites a register while a load into it is outstanding, along every path of the kernel's
control-flow graph (loops included: a block is walked again for every new queue of outstanding operations it is reached with).
`s_waitcnt vmcnt(N)` retires the OLDEST outstanding operations (they retire in order), the N youngest stay pending.

usage: python tests/tools/check_async_loads.py [file.hip ...]      (default: every csrc/*.hip that contains an asm load)
exit status 1 if a hazard is found."""
import glob, os, re, subprocess, sys, tempfile

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(REPO, "adaptive-stereo-icra-2021_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def regs(tok):
  out = set()
  for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
    out.update(range(int(m.group(1)), int(m.group(2)) + 1))
  for m in re.finditer(r"\bv(\d+)\b", tok):
    out.add(int(m.group(1)))
  return out


def _step(no, l, pending, hazards, seen):
  """One instruction against the queue of outstanding vector-memory operations (a tuple of (registers, line) in issue order;
  stores carry an empty register set).  Returns the queue behind the instruction."""
  parts = l.split(None, 1)
  op, args = parts[0], (parts[1] if len(parts) > 1 else "")
  if op.startswith("s_waitcnt"):
    m = re.search(r"vmcnt\((\d+)\)", args)
    if m:
      n = int(m.group(1))
      pending = pending[max(0, len(pending) - n):] if n else ()       # (vmcnt(n) with fewer than n outstanding retires nothing)
    return pending
  used = regs(args)
  inflight = set().union(*[p[0] for p in pending]) if pending else set()

  def hazard(rs):
    if (no, tuple(rs)) not in seen:
      seen.add((no, tuple(rs)))
      hazards.append((no, l, rs))
  if op.startswith(("global_load", "buffer_load", "scratch_load", "flat_load")) and "lds" not in op:
    dst = regs(args.split(",")[0])
    src = used - dst
    if src & inflight:
      hazard(sorted(src & inflight))
    return (pending + ((frozenset(dst), no),))[-63:]                  # (vmcnt counts to 63)
  if op.startswith(("global_store", "buffer_store", "scratch_store", "flat_store", "global_load_lds", "global_atomic")):
    if used & inflight:
      hazard(sorted(used & inflight))
    return (pending + ((frozenset(), no),))[-63:]
  if used & inflight:
    hazard(sorted(used & inflight))
  return pending


def check_kernel(name, lines):
  """lines: the kernel's instructions and labels in program order.  The queue of outstanding operations is carried along EVERY
  path of the control-flow graph (basic blocks cut at labels and branches; a block is walked again whenever it is reached
  with a queue it has not seen yet, so a load issued behind a wait in a loop body is seen in flight at the top of the next
  iteration, and the code of one wave role never inherits what another role — laid out in front of it, but not a predecessor —
  left in flight)."""
  hazards, seen = [], set()
  label_at = {l[:-1]: i for i, (no, l) in enumerate(lines) if l.endswith(":")}
  n = len(lines)

  def successors(i):
    """indices of the instructions that may execute after lines[i]"""
    no, l = lines[i]
    if l.endswith(":"):
      return [i + 1] if i + 1 < n else []
    parts = l.split(None, 1)
    op = parts[0]
    if op == "s_endpgm":
      return []
    if op == "s_branch":
      t = label_at.get(parts[1].strip()) if len(parts) > 1 else None
      return [t] if t is not None else []
    if op.startswith("s_cbranch"):
      t = label_at.get(parts[1].strip()) if len(parts) > 1 else None
      out = [i + 1] if i + 1 < n else []
      if t is not None:
        out.append(t)
      return out
    if op.startswith(("s_setpc", "s_swappc")):
      return []
    return [i + 1] if i + 1 < n else []

  leaders = {0}
  for i, (no, l) in enumerate(lines):
    if l.endswith(":"):
      leaders.add(i)
    else:
      op = l.split(None, 1)[0]
      if op.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc", "s_swappc")) and i + 1 < n:
        leaders.add(i + 1)
  visited = set()
  work = [(0, ())]
  states = 0
  while work:
    i, pending = work.pop()
    key = (i, tuple(p[1] for p in pending))
    if key in visited:
      continue
    visited.add(key)
    states += 1
    if states > 400000:
      raise RuntimeError("%s: more than 400000 (block, queue) states" % name)
    while True:                                # walk the block
      no, l = lines[i]
      if not l.endswith(":"):
        pending = _step(no, l, pending, hazards, seen)
      succ = successors(i)
      if len(succ) == 1 and succ[0] == i + 1 and (i + 1) not in leaders:
        i += 1
        continue
      for t in succ:
        work.append((t, pending))
      break
  return hazards


def check_kernel_linear(name, lines):
  """Straight-line pass in program order (branches not followed) plus ONE more turn of every loop body.  For kernels whose
  counted waits rest on a data-dependent invariant a path analysis cannot know (LINEAR_ONLY below)."""
  hazards, seen = [], set()
  label_at = {l[:-1]: i for i, (no, l) in enumerate(lines) if l.endswith(":")}

  def scan(seg, pending):
    for no, l in seg:
      if l.endswith(":"):
        continue
      if l.split(None, 1)[0] == "s_endpgm":
        break
      pending = _step(no, l, pending, hazards, seen)
    return pending
  pending, start = (), 0
  for i, (no, l) in enumerate(lines):
    parts = l.split(None, 1)
    if parts and parts[0].startswith(("s_cbranch", "s_branch")) and len(parts) > 1:
      tgt = label_at.get(parts[1].strip())
      if tgt is not None and tgt < i:
        pending = scan(lines[start:i], pending)
        scan(lines[tgt:i], pending)                        # the next iteration, from what this one left in flight
        start = i
  scan(lines[start:], pending)
  return hazards


# Kernels checked in program order only, and why the path analysis does not apply to them.
LINEAR_ONLY = {
    "conv32_act.hip": "its run-in waits with vmcnt(4) for a row behind 'exactly four by-product stores per wave' of the row before — "
                      "true because row j0 is always one of the piece's own rows and row j0-1 never is, which no path analysis "
                      "knows: on the (infeasible) paths with fewer stores the wait retires fewer loads",
}


MFMA_RESULT_WAIT = 18      # wait states before a vector instruction may read the destination of a 16-pass MFMA
VALU_TO_MFMA_WAIT = 2      # wait states between a vector instruction's write and an MFMA that reads the register


def check_mfma_hazards(lines):
  """Second check (round 4): vector instructions written in INLINE ASM (the packed differences of conv32_wino_dev.h) are not
  padded by hipcc for the matrix pipe's data hazards — it pads its own instructions only.  Program order, per kernel, wait states
  counted as hipcc's hazard recognizer counts them (an instruction = 1, s_nop N = N + 1):
    * a vector instruction inside an asm block that reads a register an MFMA wrote fewer than MFMA_RESULT_WAIT wait states ago;
    * an MFMA that reads a register a vector instruction inside an asm block wrote fewer than VALU_TO_MFMA_WAIT wait states ago
      (the first packed-difference build had exactly this: the last transform of a tile right in front of its MFMAs — two
      launches of one kernel differed).
  Branches are walked in program order (ages only grow along a fall-through; the matrix phases are straight-line code)."""
  mfma_age, asm_age, out, in_asm = {}, {}, [], False
  for no, l in lines:
    if l.endswith(":"):
      continue
    if l.startswith("#ASM"):
      in_asm = l == "#ASMSTART"
      continue
    parts = l.split(None, 1)
    op, args = parts[0], (parts[1] if len(parts) > 1 else "")
    step = int(args.strip()) + 1 if op == "s_nop" else 1
    ops = args.split(",")
    src = regs(",".join(ops[1:])) if len(ops) > 1 else set()
    if in_asm and op.startswith("v_") and not op.startswith("v_mfma"):
      young = sorted(r for r in src if mfma_age.get(r, 99) < MFMA_RESULT_WAIT)
      if young:
        out.append((no, l, young))
    if op.startswith("v_mfma"):
      young = sorted(r for r in src if asm_age.get(r, 99) < VALU_TO_MFMA_WAIT)
      if young:
        out.append((no, l, young))
    for d in (mfma_age, asm_age):
      for r in list(d):
        d[r] += step
        if d[r] >= 64:
          del d[r]
    if op.startswith("v_mfma") and ops:
      for r in regs(ops[0]):
        mfma_age[r] = 0
    elif in_asm and op.startswith("v_") and ops:
      for r in regs(ops[0]):
        asm_age[r] = 0
  return out


def main():
  def hand_waited(f):
    text = open(f).read()                  # asm loads of its own, or through the shared helpers of conv32_wino_dev.h
    return re.search(r'asm volatile\("global_load_dword', text) or '#include "conv32_wino_dev.h"' in text
  files = [os.path.abspath(f) for f in sys.argv[1:]] or [f for f in sorted(glob.glob(os.path.join(CSRC, "*.hip"))) if hand_waited(f)]
  bad = 0
  for f in files:
    with tempfile.TemporaryDirectory() as td:
      subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-inline-asm", "-I", CSRC, "-c", f,
                             "-save-temps", "-o", os.path.join(td, "x.o")] + os.environ.get("CHECK_EXTRA_FLAGS", "").split(),
                            cwd=td, stderr=subprocess.DEVNULL)
      asm = glob.glob(os.path.join(td, "*gfx950*.s"))[0]
      text = open(asm).read().split("\n")
    kernels, cur, name = {}, None, None
    for i, l in enumerate(text, 1):
      m = re.match(r"^(_Z\w+):", l)
      if m and not l.startswith("_ZN"):
        name, cur = m.group(1), []
        kernels[name] = cur
        continue
      s = l.strip()
      if cur is not None and s.startswith(";;#ASM"):
        cur.append((i, "#ASMSTART" if "START" in s else "#ASMEND"))
      elif cur is not None and s and not s.startswith((";", "//")) and (not s.startswith(".") or s.endswith(":")):
        cur.append((i, s.split(";")[0].strip() if not s.endswith(":") else s))      # (labels kept: .LBB0_3:)
    for name, lines in kernels.items():
      if not any(x[1].startswith("global_load") for x in lines):
        continue
      linear = os.path.basename(f) in LINEAR_ONLY
      hz = check_kernel_linear(name, lines) if linear else check_kernel(name, lines)
      print("%-28s %-60s %s%s" % (os.path.basename(f), name[:60], "ok" if not hz else "%d HAZARD(S)" % len(hz),
                                  " (program order only)" if linear else ""))
      for no, l, r in hz[:6]:
        print("      line %d: %s   <- in-flight %s" % (no, l[:90], r[:8]))
      mh = check_mfma_hazards(lines) if any(x[1].startswith("v_mfma") for x in lines) else []
      if mh:
        print("%-28s %-60s %d MATRIX-PIPE DATA HAZARD(S) AROUND INLINE ASM" % ("", "", len(mh)))
        for no, l, r in mh[:6]:
          print("      line %d: %s   <- too young: v%s" % (no, l[:90], r[:8]))
        hz = hz + mh
      bad += len(hz)
  return 1 if bad else 0


if __name__ == "__main__":
  sys.exit(main())
