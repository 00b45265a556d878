#!/bin/bash
# Timing-only variants of conv32_act_kernel (results are WRONG in these builds): what does each phase cost?
cd $GRAFT_REPO_ROOT/adaptive-stereo-icra-2021_amd/csrc || exit 1
for v in "-DCA_EXP_NOCONV" "-DCA_EXP_NOMOM" "-DCA_EXP_NOCONV -DCA_EXP_NOMOM"; do
  touch conv32_act.hip && make EXTRA="$v" > /dev/null 2>&1 || exit 1
  echo "== $v"
  (cd ../.. && timeout -k 10 120 python tests/tools/microbench_act.py 4 2>&1 | grep fused | sed 's/.*| //')
done
touch conv32_act.hip && make > /dev/null 2>&1
