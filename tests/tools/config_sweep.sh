#!/bin/bash
# pairs/s (fwd+adapt, fwd) at the sizes of SURVEY §8's table; usage (GPU box): tests/tools/config_sweep.sh > gpurun_out/configs.txt
run() {  # name height width k maxdisp batch
  out=$(timeout -k 10 300 python bench.py --height $2 --width $3 --k $4 --maxdisp $5 --batch $6 --steps 10 --warmup 3 --no-cpu-baseline --no-online --no-dp-overhead --no-legs 2>gpurun_out/config_sweep.err | tail -1)
  echo "$out" | python -c "
import json,sys
try:
  d=json.loads(sys.stdin.read())
  print('%-28s pairs/GPU %d: adapt %7.1f pairs/s (%6.2f ms/step)   forward %7.1f pairs/s (%5.2f ms)   dominant kernel %6.1f %s = %.2f of its bound (%s)' % ('$1', $6, d['value'], d['ms_per_step'], d['fwd_pairs_per_s'], d['fwd_ms_per_step'], d['roofline']['achieved'], d['roofline']['unit'], d['roofline']['frac'], d['roofline']['bound']))
except Exception as e:
  print('%-28s pairs/GPU %d: FAILED (%s)' % ('$1', $6, e))"
}
for b in 1 4; do
  run "plumbing 240x320 D=64 k=3"  240  320 3  64 $b
  run "SceneFlow 540x960 k=4"      540  960 4 192 $b
  run "SceneFlow 540x960 k=3"      540  960 3 192 $b
  run "KITTI 375x1242 k=4"         375 1242 4 192 $b
  run "KITTI 375x1242 k=3"         375 1242 3 192 $b
done
