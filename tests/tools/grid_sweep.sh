#!/bin/bash
# Grid-size sweep of the two persistent full-resolution kernels (conv32_bwd_fused_kernel, conv32_act_kernel): the grid is a
# BUILD-time constant (BW_GRID / CA_GRID, 512 = two resident workgroups per CU), so each point rebuilds the library.
# usage (GPU box): bash tests/tools/grid_sweep.sh
for g in 512 384 256; do
  make -C adaptive-stereo-icra-2021_amd/csrc SCAN=0 -B -j16 EXTRA="-DBW_GRID=$g -DCA_GRID=$g" conv32_bwd.o conv32_act.o > /dev/null && make -C adaptive-stereo-icra-2021_amd/csrc SCAN=0 -j16 EXTRA="-DBW_GRID=$g -DCA_GRID=$g" > /dev/null || exit 1
  for b in 1 2; do
    timeout -k 10 300 python bench.py --batch $b --steps 60 --warmup 5 --no-cpu-baseline --no-online --no-dp-overhead --no-legs --no-online 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('grid $g batch $b: %.3f ms/step; fused bwd %.1f us, act %.1f us' % (d['ms_per_step'], r['flavours'][0]['avg_launch_us'], r['flavours'][1]['avg_launch_us']))"
  done
done
make -C adaptive-stereo-icra-2021_amd/csrc SCAN=0 -B -j16 conv32_bwd.o conv32_act.o > /dev/null && make -C adaptive-stereo-icra-2021_amd/csrc SCAN=0 -j16 > /dev/null
