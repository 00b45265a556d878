#!/bin/bash
# Diagnostic builds of the generic weight-gradient kernel on the GPU box: for each EXTRA flag set, rebuild conv32_mfma.hip and
# print the per-grid medians of conv32_wgrad_kernel over tests/tools/head_ab.py.  (-DWG_EXP_NOLOAD: operands without loads,
# -DWG_EXP_NOMFMA: loads without matrix instructions — results are wrong, only the time means something.)
# usage: tests/tools/head_exp.sh "<flags A>" "<flags B>" ...     (an empty string = the production build, run it LAST)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
n=0
for flags in "$@"; do
  n=$((n+1))
  touch adaptive-stereo-icra-2021_amd/csrc/conv32_mfma.hip
  make -C adaptive-stereo-icra-2021_amd/csrc SCAN=0 EXTRA="$flags" > gpurun_out/head_exp_build_$n.log 2>&1 || { tail -5 gpurun_out/head_exp_build_$n.log; exit 1; }
  rm -rf gpurun_out/head_exp_$n
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/head_exp_$n -o h -- python3 tests/tools/head_ab.py ${HEAD_IMAGES:-8} > gpurun_out/head_exp_$n.log 2>&1 || { tail -5 gpurun_out/head_exp_$n.log; exit 1; }
  echo "== [$flags]"
  python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/head_exp_$n/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
  if "wgrad" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]:
    d[(r["Kernel_Name"][:44], int(r["Grid_Size_X"]) // 256)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()): print("   %-46s %5d workgroups  %3d x  median %8.1f us  min %8.1f" % (k[0], k[1], len(v), sorted(v)[len(v) // 2], min(v)))
PY
done
