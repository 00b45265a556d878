"""Lists every kernel launch of ONE adaptation step in launch order: by default the last one of the run; with
``fastest`` the step with the shortest wall time (a hipGraph replay when bench.py ran in graph mode).
usage: python tests/tools/step_timeline.py <kernel_trace.csv> [fastest]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
# two adam launches per step: step j ends at adam[2j+1] and starts after adam[2j-1]
steps = [(adam[2 * j - 1] + 1, adam[2 * j + 1] + 1) for j in range(1, len(adam) // 2)]
if len(sys.argv) > 2 and sys.argv[2] == "fastest":
  lo, hi = min(steps, key=lambda se: int(rows[se[1] - 1]['End_Timestamp']) - int(rows[se[0]]['Start_Timestamp']))
else:
  lo, hi = steps[-1]
win = rows[lo:hi]
t0 = int(win[0]['Start_Timestamp'])
print("step wall %.1f us, %d kernels, busy %.1f us" % (
    (int(win[-1]['End_Timestamp']) - t0) / 1e3, len(win),
    sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in win) / 1e3))
prev_end = t0
for r in win:
  s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
  print("%9.1f us  +%6.1f gap  %8.1f us  q%-2s grid %-8s %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3,
        r.get('Queue_Id', '?'), r.get('Grid_Size_X', r.get('Grid_Size', '?')), r['Kernel_Name'][:90]))
  prev_end = max(prev_end, e)
