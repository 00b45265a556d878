"""Summarises the forward-only (inference) steps at the end of a rocprofv3 --kernel-trace of bench.py.
usage: python tests/tools/forward_window.py <kernel_trace.csv> <timed_steps>"""
import collections, csv, sys
path, steps = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
fwd = rows[adam[-1] + 1:]
# an inference ends with the refinement's 32->1 output convolution; keep the last `steps` of them
ends = [i for i, r in enumerate(fwd) if 'conv32to1_2d_fwd_kernel' in r['Kernel_Name'] or 'refine_out_kernel' in r['Kernel_Name']]
first = ends[-steps - 1] + 1
win = fwd[first:ends[-1] + 1]
t0, t1 = int(win[0]['Start_Timestamp']), int(win[-1]['End_Timestamp'])
agg = collections.defaultdict(lambda: [0, 0])
for r in win:
  d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
  agg[r['Kernel_Name']][0] += d; agg[r['Kernel_Name']][1] += 1
tot = sum(v[0] for v in agg.values())
print("forward-only window = the last %d inference steps: wall %.2f ms, kernel busy %.2f ms (%.0f%%), %.0f kernels per step, %.2f ms per step" % (
    steps, (t1 - t0) / 1e6, tot / 1e6, 100 * tot / (t1 - t0), len(win) / steps, (t1 - t0) / 1e6 / steps))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:30]:
  print("%6.2f%% %8.1f us/step  x%-4d avg %7.1fus  %s" % (100 * v[0] / tot, v[0] / steps / 1e3, v[1] // steps, v[0] / v[1] / 1e3, k[:100]))
