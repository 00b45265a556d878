#!/bin/bash
# Data gradient of a full-resolution layer, generation 1 (conv32_wino.hip MODE 2) against generation 2 (conv32_wino_dgrad.hip):
# rocprofv3 kernel trace over tests/tools/wino_microbench.py, per-kernel averages.  usage: tests/tools/dgrad_gen_ab.sh [pairs]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
rm -rf gpurun_out/dgrad_gen_ab
WMB_ONLY=bwd WMB_GEN=1,2 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dgrad_gen_ab -o w -- python3 tests/tools/wino_microbench.py ${1:-4} > gpurun_out/dgrad_gen_ab.log 2>&1 || { tail -20 gpurun_out/dgrad_gen_ab.log; exit 1; }
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/dgrad_gen_ab/**/*kernel_stats.csv",recursive=True)[0]
for r in sorted(csv.DictReader(open(f)), key=lambda r: r["Name"]):
  if "wino_" in r["Name"] and "pack" not in r["Name"]: print("   %-60s %4s x %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
