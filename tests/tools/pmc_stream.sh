#!/bin/bash
# HBM traffic of a continuous adaptation run (BASELINE configs[4], one-GPU leg): two rocprofv3 --pmc passes (FETCH_SIZE;
# WRITE_SIZE GRBM_GUI_ACTIVE — counters in their own runs, kernel trace only) over tests/tools/adapt_stream.py, summarised per
# kernel and per step by tests/tools/pmc_stream_summarize.py.
# usage (GPU box): bash tests/tools/pmc_stream.sh <tag> [steps, default 1000]
tag=${1:-x}
steps=${2:-1000}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
run() {
  name=$1; shift
  timeout -k 10 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmcs_${tag}_$name -o p -- python3 tests/tools/adapt_stream.py --steps $steps --height 375 --width 1242 > gpurun_out/pmcs_${tag}_$name.log 2>&1 || exit 1
  python3 tests/tools/pmc_stream_summarize.py $(find gpurun_out/pmcs_${tag}_$name -name "p_counter_collection.csv" | head -1) gpurun_out/pmcs_${tag}_$name.json || exit 1
  rm -rf gpurun_out/pmcs_${tag}_$name
}
run fetch FETCH_SIZE && run write WRITE_SIZE GRBM_GUI_ACTIVE &&
python3 tests/tools/pmc_stream_summarize.py --merge gpurun_out/pmcs_${tag}_fetch.json gpurun_out/pmcs_${tag}_write.json $steps gpurun_out/pmcs_${tag}_fetch.log gpurun_out/pmc_stream_$tag.json
