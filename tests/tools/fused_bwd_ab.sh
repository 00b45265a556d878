#!/bin/bash
# Same-box A/B of build-time variants of the one-launch backward (GPU box): for each EXTRA flag set, rebuild
# conv32_wino_bwd.hip and time as_conv32_wino_bwd_fused at the bench workload (tests/tools/wino_microbench.py, "wino 1L" lines).
# The production build (no flags) must be run LAST: it leaves the library.
# usage: tests/tools/fused_bwd_ab.sh "<flags A>" "<flags B>" ... ""
cd $GRAFT_REPO_ROOT || exit 1
for flags in "$@"; do
  touch adaptive-stereo-icra-2021_amd/csrc/conv32_wino_bwd.hip
  make -C adaptive-stereo-icra-2021_amd/csrc SCAN=0 EXTRA="$flags" > gpurun_out/fused_ab_build.log 2>&1 || { tail -5 gpurun_out/fused_ab_build.log; exit 1; }
  echo "== [$flags]"
  WMB_ONLY=bwd WMB_GEN= timeout -k 10 200 python tests/tools/wino_microbench.py 4 2>&1 | grep "1L" | cut -c1-60
done
