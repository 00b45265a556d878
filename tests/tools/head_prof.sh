#!/bin/bash
# rocprofv3 kernel trace over tests/tools/head_ab.py: per-kernel averages of the strided head (generic and staged-row kernels).
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_head -o head -- python3 tests/tools/head_ab.py ${1:-8} > gpurun_out/prof_head.log 2>&1 || { tail -20 gpurun_out/prof_head.log; exit 1; }
f=$(find gpurun_out/prof_head -name "*kernel_stats.csv" | sort | tail -1)
[ -n "$f" ] || { echo "no kernel_stats.csv"; exit 1; }
python3 - "$f" <<'PY' | tee gpurun_out/head_kernels.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
  print("%-90s calls %5s  avg %9.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
