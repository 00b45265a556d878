#!/bin/bash
# PMC rows for the SMALL kernels of a step (feature towers, BatchNorm residue): one rocprofv3 --pmc pass over bench.py
# at <pairs> per step, eager launches on one stream, summarised per kernel name into gpurun_out/pmc_small_<tag>.json.
# usage (GPU box): tests/tools/pmc_small.sh <tag> [pairs per step, default 1]
tag=${1:-x}
pairs=${2:-1}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
  --output-format csv -d gpurun_out/pmcs_$tag -o p -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-online --no-dp-overhead --no-legs --no-graph --one-stream --batch $pairs > gpurun_out/pmcs_$tag.log 2>&1 || exit 1
python3 tests/tools/pmc_small_summarize.py $(find gpurun_out/pmcs_$tag -name "p_counter_collection.csv" | head -1) $pairs gpurun_out/pmc_small_$tag.json
rm -rf gpurun_out/pmcs_$tag
