cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d gpurun_out/pmc_bwd -o p -- python3 tests/tools/microbench_bwd.py 4 > gpurun_out/pmc_bwd.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_bwd/**/p_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
  n = r["Kernel_Name"]
  if "bwd_fused" in n or "wgrad_lds2" in n or "conv32_lds_kernel" in n:
    acc[n[:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in acc.items():
  for c, v in d.items():
    # launches come in groups of 23 per dilation (3 warm-up + 20): show the mean per dilation
    k = len(v) // 4
    print("%-50s %-28s %s" % (n, c, " ".join("%.4g" % (sum(v[i*k:(i+1)*k]) / k) for i in range(4))))
PY
