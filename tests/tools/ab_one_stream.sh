for b in 4 1; do for rep in 1 2; do for f in "" "--one-stream"; do
timeout -k 10 300 python bench.py --batch $b --steps 40 --warmup 5 --no-cpu-baseline --no-online --no-dp-overhead --no-legs $f 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('batch $b [$f]: %.3f ms/step, forward %.3f ms' % (d['ms_per_step'], d['fwd_ms_per_step']))"
done; done; done
