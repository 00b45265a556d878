#!/bin/bash
# Instruction-mix counters of the minimal-filtering kernels (one rocprofv3 --pmc pass over tests/tools/wino_microbench.py).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
rm -rf gpurun_out/pmc_wino
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_wino -o p -- python3 tests/tools/wino_microbench.py ${1:-4} > gpurun_out/pmc_wino.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_wino/**/p_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
  n = r["Kernel_Name"]
  if "wino_kernel" in n or "wino_wgrad" in n or "conv32_act" in n or "bwd_fused" in n:
    acc[n[:46]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in sorted(acc.items()):
  print(n)
  for c, v in sorted(d.items()):
    print("    %-28s %.5g  (x%d)" % (c, sum(v) / len(v), len(v)))
PY
