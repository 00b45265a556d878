#!/bin/bash
# A/B builds of the minimal-filtering kernels on the GPU box: for each EXTRA flag set, rebuild and print per-kernel averages.
# usage: tests/tools/wino_exp.sh "<flags A>" "<flags B>" ...     (an empty string = the production build)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
n=0
for flags in "$@"; do
  n=$((n+1))
  touch adaptive-stereo-icra-2021_amd/csrc/conv32_wino.hip adaptive-stereo-icra-2021_amd/csrc/conv32_wino_wgrad.hip
  make -C adaptive-stereo-icra-2021_amd/csrc SCAN=0 EXTRA="$flags" > gpurun_out/wino_exp_build_$n.log 2>&1 || exit 1
  rm -rf gpurun_out/wino_exp_$n
  WMB_ONLY=${WMB_ONLY:-bwd} timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/wino_exp_$n -o w -- python3 tests/tools/wino_microbench.py ${WMB_PAIRS:-4} > gpurun_out/wino_exp_$n.log 2>&1 || exit 1
  echo "== [$flags]"
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/wino_exp_$n/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
  if "wino_" in r["Name"] and "pack" not in r["Name"]: print("   %-60s %4s x %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
