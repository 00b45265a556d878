"""Lists the kernels of the LAST forward-only (inference) step of a rocprofv3 --kernel-trace of bench.py with their
start offsets, so that overlap between the two feature-extraction streams is visible.
usage: python tests/tools/forward_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
fwd = rows[adam[-1] + 1:]
ends = [i for i, r in enumerate(fwd) if 'conv32to1_2d_fwd_kernel' in r['Kernel_Name']]
win = fwd[ends[-2] + 1:ends[-1] + 1]
t0 = int(win[0]['Start_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in win)
print("wall %.1f us, busy %.1f us, %d kernels" % ((int(win[-1]['End_Timestamp']) - t0) / 1e3, busy / 1e3, len(win)))
for r in win:
  s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
  print("%9.1f .. %9.1f us  %7.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'][:80]))
