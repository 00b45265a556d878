#!/bin/bash
# Timing-only variants of the fused backward kernel (results are WRONG in these builds): what does each phase cost?
# usage (GPU box): bash tests/tools/bwd_experiments.sh
cd $GRAFT_REPO_ROOT/adaptive-stereo-icra-2021_amd/csrc || exit 1
for v in "-DBW_EXP_NOEPI" "-DBW_EXP_NOCONV" "-DBW_EXP_NOEPI -DBW_EXP_NOCONV" "-DBW_EXP_NOEPI -DBW_EXP_NOCONV -DBW_EXP_NOB1"; do
  touch conv32_bwd.hip && make EXTRA="$v" > /dev/null 2>&1 || exit 1
  echo "== $v"
  (cd ../.. && timeout -k 10 120 python tests/tools/microbench_bwd.py 4 2>&1 | grep fused | sed 's/.*| //')
done
touch conv32_bwd.hip && make > /dev/null 2>&1
