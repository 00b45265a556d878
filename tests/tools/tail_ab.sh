#!/bin/bash
# a4 + a5 + a8 tail: first generation (LDS-staged planes, EXTRA=-DTAIL_FIRST_GENERATION) against the second (direct loads, no
# barrier in the plane loop): rocprofv3 kernel trace over tests/tools/microbench_agg3d.py.  usage: tests/tools/tail_ab.sh [pairs ...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
for flags in "-DTAIL_FIRST_GENERATION" ""; do
  touch adaptive-stereo-icra-2021_amd/csrc/agg_tail.hip
  make -C adaptive-stereo-icra-2021_amd/csrc SCAN=0 EXTRA="$flags" > gpurun_out/tail_ab_build.log 2>&1 || exit 1
  for pairs in ${@:-4}; do
    rm -rf gpurun_out/tail_ab
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tail_ab -o w -- python3 tests/tools/microbench_agg3d.py $pairs > gpurun_out/tail_ab.log 2>&1 || { tail -5 gpurun_out/tail_ab.log; exit 1; }
    echo "== [$flags] $pairs pairs"
    python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/tail_ab/**/*kernel_stats.csv",recursive=True)[0]
for r in sorted(csv.DictReader(open(f)), key=lambda r: r["Name"]):
  if "agg_tail" in r["Name"]: print("   %-60s %4s x %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
  done
done
