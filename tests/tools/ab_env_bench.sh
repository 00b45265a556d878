#!/bin/bash
# Same-box A/B of one environment switch on the bench line, interleaved: A B A B ...
# usage (GPU box): bash tests/tools/ab_env_bench.sh <VAR> <value A> <value B> [rounds, default 3] [bench.py arguments]
var=$1; a=$2; b=$3; n=${4:-3}; shift 4
for i in $(seq 1 $n); do
  for v in $a $b; do
    line=$(env $var=$v timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-online --no-dp-overhead --no-legs "$@" 2>/dev/null | tail -1)
    echo "$var=$v  $(echo "$line" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step  %.1f pairs/s  fwd %.3f ms' % (d['ms_per_step'], d['value'], d['fwd_ms_per_step']))")"
  done
done
