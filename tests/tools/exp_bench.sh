#!/bin/bash
# Same-box A/B of BUILD flags on the bench line itself (graph replay, default workload): for each EXTRA flag set, rebuild the given
# sources and run bench.py twice; prints ms per step and forward ms.  The production build ("") must be run LAST.
# usage: tests/tools/exp_bench.sh "<file.hip ...>" "<flags A>" "<flags B>" ... ""
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
files=$1; shift
n=0
for flags in "$@"; do
  n=$((n+1))
  for f in $files; do touch adaptive-stereo-icra-2021_amd/csrc/$f; done
  make -C adaptive-stereo-icra-2021_amd/csrc SCAN=0 EXTRA="$flags" > gpurun_out/exp_bench_build_$n.log 2>&1 || { tail -5 gpurun_out/exp_bench_build_$n.log; exit 1; }
  for rep in 1 2; do
    timeout -k 10 300 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-online --no-dp-overhead --no-legs > gpurun_out/exp_bench_$n.json 2> gpurun_out/exp_bench_$n.err || { tail -5 gpurun_out/exp_bench_$n.err; exit 1; }
    python3 -c "
import json; d = json.load(open('gpurun_out/exp_bench_$n.json'))
print('[%s] run $rep: %.3f ms per step, forward %.3f ms' % ('$flags', d['ms_per_step'], d['fwd_ms_per_step']))"
  done
done
