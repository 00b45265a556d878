#!/bin/bash
# Same-box A/B of one environment switch INSIDE a step: for each value, a rocprofv3 kernel trace of bench.py (--no-graph, 10
# steps) and the per-launch averages of the kernels matching a pattern.
# usage (GPU box): bash tests/tools/ab_env_step.sh <VAR> "<kernel pattern>" <value A> <value B> ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
var=$1; pat=$2; shift 2
n=0
for v in "$@"; do
  n=$((n+1))
  rm -rf gpurun_out/ab_env_step_$n
  export $var=$v
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_env_step_$n -o e -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-online --no-dp-overhead --no-legs --no-graph > gpurun_out/ab_env_step_$n.log 2> gpurun_out/ab_env_step_$n.err || { tail -5 gpurun_out/ab_env_step_$n.err; exit 1; }
  echo "== [$var=$v]"
  python3 - "$pat" <<PY
import csv, glob, re, sys
f = glob.glob("gpurun_out/ab_env_step_$n/**/*kernel_stats.csv", recursive=True)[0]
tot = 0.0
for r in csv.DictReader(open(f)):
  if re.search(sys.argv[1], r["Name"]):
    print("   %-70s %4s x %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3)); tot += float(r["TotalDurationNs"]) / 1e3
print("   total of the matching kernels over the run: %.1f us" % tot)
PY
  rm -rf gpurun_out/ab_env_step_$n
done
