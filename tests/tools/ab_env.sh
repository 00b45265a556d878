#!/bin/bash
# Same-box A/B of an environment switch: bash tests/tools/ab_env.sh VAR [batch] — alternates VAR=1 / VAR=0 three times each
# (box-to-box variation is +-2 %, a 1 % effect only shows on one box, interleaved).
var=$1; b=${2:-4}
for rep in 1 2 3; do for v in 1 0; do
  env $var=$v timeout -k 10 300 python bench.py --batch $b --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$var=$v batch $b: %.3f ms/step  %.1f pairs/s' % (d['ms_per_step'], d['value']))"
done; done
