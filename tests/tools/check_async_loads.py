"""Static check of the hand-waited vector-memory loads in csrc/*.hip.

The kernels that issue `global_load_*` from inline asm and retire them with their own `s_waitcnt vmcnt(N)` tell hipcc nothing
about WHEN the destination registers become valid: the compiler may copy (or spill) them between the load and the wait — a read
of an in-flight destination, i.e. stale data that comes and goes with memory latency (this happened once: a v_mov of ten
operand registers in front of a branch that held two different waits).  This script compiles a source to ISA and reports every
instruction that reads or overwrites a register while an asm-issued load into it is outstanding; the register set is cleared at
every `s_waitcnt vmcnt(...)`.  Waits counted with N > 0 are handled conservatively: the OLDEST outstanding loads are retired
first (loads retire in order), N of the youngest stay pending.

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


def _scan(lines, pending, hazards, seen):
  """One straight-line pass over `lines` from the given queue of outstanding operations; returns the queue at its end."""
  for no, l in lines:
    if l.endswith(":"):
      continue
    parts = l.split(None, 1)
    op, args = parts[0], (parts[1] if len(parts) > 1 else "")
    if op.startswith("s_waitcnt"):
      m = re.search(r"vmcnt\((\d+)\)", args)
      if m:
        n = int(m.group(1))
        pending = pending[max(0, len(pending) - n):] if n else []       # (vmcnt(n) with fewer than n outstanding retires nothing)
      continue
    if op in ("s_endpgm",):
      break
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
      pending = pending + [(dst, no)]
      continue
    if op.startswith(("global_store", "buffer_store", "scratch_store", "flat_store", "global_load_lds", "global_atomic")):
      if used & inflight:
        hazard(sorted(used & inflight))
      pending = pending + [(set(), no)]
      continue
    if used & inflight:
      hazard(sorted(used & inflight))
  return pending


def check_kernel(name, lines):
  """lines: the kernel's instructions and labels in program order.  The pass is straight-line, plus ONE more turn of every
  loop: at a backward branch the body (label .. branch) is scanned again starting from the queue of operations outstanding
  at the branch, so a load issued textually BEHIND a wait in the body is seen in flight at the top of the next iteration."""
  hazards, seen = [], set()
  label_at = {l[:-1]: i for i, (no, l) in enumerate(lines) if l.endswith(":")}
  pending, start = [], 0
  for i, (no, l) in enumerate(lines):
    parts = l.split(None, 1)
    if parts and parts[0].startswith(("s_cbranch", "s_branch")) and len(parts) > 1:
      tgt = label_at.get(parts[1].strip())
      if tgt is not None and tgt < i:
        pending = _scan(lines[start:i], pending, hazards, seen)
        _scan(lines[tgt:i], list(pending), hazards, seen)        # the next iteration, from what this one left in flight
        start = i
  _scan(lines[start:], pending, hazards, seen)
  return hazards


def main():
  files = [os.path.abspath(f) for f in sys.argv[1:]] or [f for f in sorted(glob.glob(os.path.join(CSRC, "*.hip")))
                           if re.search(r'asm volatile\("global_load_dword', open(f).read())]
  bad = 0
  for f in files:
    with tempfile.TemporaryDirectory() as td:
      subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-inline-asm", "-I", CSRC, "-c", f,
                             "-save-temps", "-o", os.path.join(td, "x.o")], cwd=td, stderr=subprocess.DEVNULL)
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
      if cur is not None and s and not s.startswith((";", "//")) and (not s.startswith(".") or s.endswith(":")):
        cur.append((i, s.split(";")[0].strip() if not s.endswith(":") else s))      # (labels kept: .LBB0_3:)
    for name, lines in kernels.items():
      if not any(x[1].startswith("global_load") for x in lines):
        continue
      hz = check_kernel(name, lines)
      print("%-28s %-60s %s" % (os.path.basename(f), name[:60], "ok" if not hz else "%d HAZARD(S)" % len(hz)))
      for no, l, r in hz[:6]:
        print("      line %d: %s   <- in-flight %s" % (no, l[:90], r[:8]))
      bad += len(hz)
  return 1 if bad else 0


if __name__ == "__main__":
  sys.exit(main())
