#!/bin/bash
# usage (on the GPU box, via gpurun): tests/tools/prof_step.sh <tag> [bench.py arguments, default --no-graph]
#   -> gpurun_out/ss_<tag>.txt, timeline_<tag>.txt      (PAIRS=<pairs per step> when --batch is given: star_kernels.py needs it)
tag=${1:-x}
shift
extra=${@:---no-graph}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o e -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-online --no-dp-overhead --no-legs $extra > gpurun_out/be.log 2>gpurun_out/be.err || exit 1
trace=$(find gpurun_out/prof_$tag -name "e_kernel_trace.csv" | head -1)
python tests/tools/steady_state.py $trace 10 "$tag steady state ($extra)" > gpurun_out/ss_$tag.txt
python tests/tools/step_timeline.py $trace fastest > gpurun_out/timeline_$tag.txt
python tests/tools/forward_window.py $trace 10 > gpurun_out/fw_$tag.txt
python tests/tools/star_kernels.py $trace 10 ${PAIRS:-4} > gpurun_out/star_$tag.txt
cp $(find gpurun_out/prof_$tag -name "e_kernel_stats.csv" | head -1) gpurun_out/ks_$tag.csv
rm -rf gpurun_out/prof_$tag
tail -1 gpurun_out/be.log | cut -c1-120
