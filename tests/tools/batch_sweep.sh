#!/bin/bash
# pairs/s (fwd+adapt, fwd) over pairs per GPU; usage (GPU box): tests/tools/batch_sweep.sh > gpurun_out/sweep.txt
for b in 1 2 4 8 16; do
  timeout -k 10 300 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-online --no-dp-overhead --no-legs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('pairs/GPU %2d: adapt %7.1f pairs/s (%6.2f ms/step)   forward %7.1f pairs/s (%5.2f ms)   conv32_lds %5.1f TFLOP/s' % ($b, d['value'], d['ms_per_step'], d['fwd_pairs_per_s'], d['fwd_ms_per_step'], d['roofline']['achieved']))"
done
