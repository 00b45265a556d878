for g in 512 384 256; do
  AS_BW_GRID=$g AS_CA_GRID=$g timeout -k 10 300 python bench.py --batch 1 --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('grid $g batch 1: %.3f ms/step; fused bwd %.1f us, act %.1f us' % (d['ms_per_step'], r['flavours'][0]['avg_launch_us'], r['flavours'][1]['avg_launch_us']))"
done
for g in 512 384; do
  AS_BW_GRID=$g AS_CA_GRID=$g timeout -k 10 300 python bench.py --batch 2 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('grid $g batch 2: %.3f ms/step; fused bwd %.1f us, act %.1f us' % (d['ms_per_step'], r['flavours'][0]['avg_launch_us'], r['flavours'][1]['avg_launch_us']))"
done
