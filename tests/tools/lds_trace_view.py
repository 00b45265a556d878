import numpy as np, sys
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(512, 32, 8)
hw = t[:, 0, 7].copy()
key = ((hw >> 32) & 15) << 8 | ((hw >> 8) & 0xFF)
ts = t[:, :, :7].astype(np.int64)
valid = ts[:, :, 0] > 0
t0 = ts[valid].min()
ts = (ts - t0) * 10   # ns (100 MHz)
names = ["wait_dma", "barrier1", "mfma", "barrier2", "issue_dma", "epilogue"]
print("kernel span %.1f us" % (ts[valid].max() / 1e3))
d = np.diff(ts, axis=2)   # [wg][tile][6]
nt = valid.sum(1)
for i, n in enumerate(names):
  x = d[:, 1:25, i][valid[:, 1:25]]
  print("%-10s mean %7.0f ns  p10 %7.0f p50 %7.0f p90 %7.0f" % (n, x.mean(), *np.percentile(x, [10, 50, 90])))
per = (ts[:, 1:25, 0][:, 1:] - ts[:, 1:25, 0][:, :-1])
print("tile period mean %.0f ns" % per[valid[:, 2:25]].mean())
# one CU pair timeline
import collections
groups = collections.defaultdict(list)
for b in range(512): groups[int(key[b])].append(b)
k0 = sorted(groups)[5]
print("CU key %x blocks %s" % (k0, groups[k0]))
for b in groups[k0]:
  print("block", b)
  for it in range(3, 9):
    print("  tile %2d: " % it + " ".join("%7.2f" % (ts[b, it, s] / 1e3) for s in range(7)))
# overlap: fraction of time both WGs of a CU are in the MFMA phase
tot_both = tot_any = 0
for k, bl in groups.items():
  if len(bl) != 2: continue
  a, b = bl
  ev = []
  for w in (a, b):
    for it in range(1, 25):
      if valid[w, it]: ev.append((ts[w, it, 2], ts[w, it, 3], w))
  # sample on a grid
  lo = min(e[0] for e in ev); hi = max(e[1] for e in ev)
  grid = np.arange(lo, hi, 50)
  ca = np.zeros(len(grid), bool); cb = np.zeros(len(grid), bool)
  for s, e, w in ev:
    m = (grid >= s) & (grid < e)
    if w == a: ca |= m
    else: cb |= m
  tot_both += (ca & cb).sum(); tot_any += (ca | cb).sum(); 
  tot = len(grid)
print("both in MFMA %.2f, any in MFMA %.2f of sampled time (last CU: grid %d)" % (tot_both / (tot * len(groups)), tot_any / (tot * len(groups)), tot))
