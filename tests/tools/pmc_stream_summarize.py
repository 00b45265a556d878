"""Reduces a rocprofv3 --pmc counter CSV of a long run to per-kernel sums (streaming: the CSV has one row per dispatch and
counter), or merges the FETCH and WRITE passes into the committed summary.
  python pmc_stream_summarize.py <counter_collection.csv> <out.json>
  python pmc_stream_summarize.py --merge <fetch.json> <write.json> <steps> <adapt_stream log> <out.json>
HBM bytes follow MI355X_MICROARCH.md: read bytes = 2 x FETCH_SIZE(KB) x 1024 on gfx950, WRITE_SIZE(KB) x 1024 is exact."""
import collections, csv, json, re, sys


def reduce_csv(path, out):
  acc = collections.defaultdict(lambda: collections.defaultdict(float))
  calls = collections.Counter()
  with open(path) as f:
    for r in csv.DictReader(f):
      name = re.sub(r"\(.*", "", r["Kernel_Name"])[:80]
      acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
      if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
        calls[name] += 1
  json.dump({k: dict(v, launches=calls[k]) for k, v in acc.items()}, open(out, "w"))


def merge(fetch, write, steps, log, out):
  fe, wr = json.load(open(fetch)), json.load(open(write))
  steps = int(steps)
  rows = []
  for k in sorted(set(fe) | set(wr)):
    rd = 2.0 * fe.get(k, {}).get("FETCH_SIZE", 0.0) * 1024
    wt = wr.get(k, {}).get("WRITE_SIZE", 0.0) * 1024
    rows.append({"kernel": k, "launches": int(fe.get(k, wr.get(k, {})).get("launches", 0)), "read_bytes": rd, "write_bytes": wt})
  rows.sort(key=lambda r: -(r["read_bytes"] + r["write_bytes"]))
  tot_r, tot_w = sum(r["read_bytes"] for r in rows), sum(r["write_bytes"] for r in rows)
  gui = sum(v.get("GRBM_GUI_ACTIVE", 0.0) for v in wr.values()) / 8.0          # summed over the 8 XCDs
  summary = None
  for line in open(log):
    if line.startswith("{"):
      summary = json.loads(line)
  res = {"what": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE GRBM_GUI_ACTIVE (two passes) over tests/tools/adapt_stream.py "
                 "--steps %d --height 375 --width 1242 (VS+ER, one pair per step)" % steps,
         "correction": "gfx950: read bytes = 2 x FETCH_SIZE(KB) x 1024; WRITE_SIZE(KB) x 1024 exact (MI355X_MICROARCH.md, HBM)",
         "steps": steps, "hbm_read_bytes_total": tot_r, "hbm_write_bytes_total": tot_w,
         "hbm_bytes_per_step": (tot_r + tot_w) / steps,
         "gpu_busy_cycles_sum_over_kernels": gui,
         "mean_hbm_GBps_while_kernels_run_at_2.4GHz": (tot_r + tot_w) / (gui / 2.4e9) / 1e9 if gui else None,
         "stream_summary_under_the_profiler": summary, "top_kernels": rows[:25], "kernels": len(rows)}
  json.dump(res, open(out, "w"), indent=1)
  print(json.dumps({k: res[k] for k in ("steps", "hbm_bytes_per_step", "mean_hbm_GBps_while_kernels_run_at_2.4GHz")}))


if __name__ == "__main__":
  if sys.argv[1] == "--merge":
    merge(*sys.argv[2:7])
  else:
    reduce_csv(sys.argv[1], sys.argv[2])
