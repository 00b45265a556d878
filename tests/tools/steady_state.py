"""Summarises the timed adaptation steps of a `rocprofv3 --kernel-trace` run of bench.py.
usage: python tests/tools/steady_state.py <kernel_trace.csv> <timed_steps> [title]"""
import collections, csv, sys
path, steps = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
# two adam launches per step; the timed steps are the last `steps` of them
win = rows[adam[-2 * steps - 1] + 1:adam[-1] + 1]
t0, t1 = int(win[0]['Start_Timestamp']), int(win[-1]['End_Timestamp'])
agg = collections.defaultdict(lambda: [0, 0])
for r in win:
  d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
  agg[r['Kernel_Name']][0] += d; agg[r['Kernel_Name']][1] += 1
tot = sum(v[0] for v in agg.values())
if len(sys.argv) > 3:
  print(sys.argv[3])
print("steady-state window = the %d timed adaptation steps" % steps)
print("window wall %.2f ms; kernel busy %.2f ms (%.0f%%); %.0f kernels per step; %.2f ms per step" % (
    (t1 - t0) / 1e6, tot / 1e6, 100 * tot / (t1 - t0), len(win) / steps, (t1 - t0) / 1e6 / steps))
import os
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:(1000 if os.environ.get('SS_ALL') else 45)]:
  print("%6.2f%% %8.1f us/step  x%-4d avg %7.1fus  %s" % (100 * v[0] / tot, v[0] / steps / 1e3, v[1] // steps, v[0] / v[1] / 1e3, k[:100]))
