"""Lists every kernel launch of ONE steady-state adaptation step (the last one) in launch order.
usage: python tests/tools/step_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
win = rows[adam[-3] + 1:adam[-1] + 1]
t0 = int(win[0]['Start_Timestamp'])
prev_end = t0
for r in win:
  s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
  print("%9.1f us  +%6.1f gap  %8.1f us  grid %-8s %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3,
        r.get('Grid_Size_X', r.get('Grid_Size', '?')), r['Kernel_Name'][:90]))
  prev_end = e
