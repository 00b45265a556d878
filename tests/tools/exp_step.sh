#!/bin/bash
# Diagnostic builds inside a real step (GPU box): for each EXTRA flag set, rebuild the given source files and print the per-launch
# averages of the kernels matching a pattern from a rocprofv3 trace of bench.py (--no-graph, 10 steps).  Results of a diagnostic
# build are WRONG by construction; only the times mean something.  The production build must be run LAST (it leaves the library).
# usage: tests/tools/exp_step.sh "<file.hip ...>" "<kernel pattern>" "<flags A>" "<flags B>" ... ""
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
files=$1; pat=$2; shift 2
n=0
for flags in "$@"; do
  n=$((n+1))
  for f in $files; do touch adaptive-stereo-icra-2021_amd/csrc/$f; done
  make -C adaptive-stereo-icra-2021_amd/csrc SCAN=0 EXTRA="$flags" > gpurun_out/exp_step_build_$n.log 2>&1 || { tail -5 gpurun_out/exp_step_build_$n.log; exit 1; }
  rm -rf gpurun_out/exp_step_$n
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/exp_step_$n -o e -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-online --no-dp-overhead --no-legs --no-graph > gpurun_out/exp_step_$n.log 2> gpurun_out/exp_step_$n.err || { tail -5 gpurun_out/exp_step_$n.err; exit 1; }
  echo "== [$flags]"
  python3 - "$pat" <<PY
import csv, glob, re, sys
f = glob.glob("gpurun_out/exp_step_$n/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
  if re.search(sys.argv[1], r["Name"]): print("   %-70s %4s x %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf gpurun_out/exp_step_$n
done
