for b in 4 1; do for s in 1 0; do
AS_WGRAD_SIDE=$s timeout -k 10 300 python bench.py --batch $b --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/ab_${b}_$s.json 2> gpurun_out/ab_${b}_$s.err || exit 1
python -c "
import json; d=json.loads(open('gpurun_out/ab_${b}_$s.json').read().strip().splitlines()[-1]); print('batch $b side $s:', d['value'], d['ms_per_step'], d.get('eager_ms_per_step'))"
done; done
