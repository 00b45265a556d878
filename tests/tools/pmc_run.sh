#!/bin/bash
# Three separate rocprofv3 --pmc passes over tests/tools/pmc_conv.py (counters must not be combined with tracing
# domains other than the kernel trace), summarised into gpurun_out/pmc_<tag>.json.
# usage (GPU box): tests/tools/pmc_run.sh <tag> [pairs per launch, default 4]
tag=${1:-x}
pairs=${2:-4}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
run() {  # name counters...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${tag}_$name -o p -- python3 tests/tools/pmc_conv.py $pairs > gpurun_out/pmc_$name.log 2>&1 || exit 1
  cp $(find gpurun_out/pmc_${tag}_$name -name "p_counter_collection.csv" | head -1) gpurun_out/pmc_${tag}_$name.csv
  rm -rf gpurun_out/pmc_${tag}_$name
}
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE && run fetch FETCH_SIZE && run write WRITE_SIZE GRBM_GUI_ACTIVE &&
python tests/tools/pmc_summarize.py gpurun_out/pmc_${tag}_sq.csv gpurun_out/pmc_${tag}_fetch.csv gpurun_out/pmc_${tag}_write.csv gpurun_out/pmc_$tag.json $pairs > /dev/null &&
python - <<PY
import json
d = json.load(open("gpurun_out/pmc_$tag.json"))
for k, v in d["kernels"].items():
  print("%-36s hbm %7.1f MB (%.2fx)  mfma busy %.3f  lds conflicts %s" % (k, v["hbm_bytes_per_launch"] / 1e6, v["traffic_over_algorithmic"], v["mfma_busy_fraction_of_simd_cycles"], v["SQ_LDS_BANK_CONFLICT"]))
PY
