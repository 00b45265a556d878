"""Per-kernel means of one rocprofv3 --pmc pass over bench.py (tests/tools/pmc_small.sh).
usage: python tests/tools/pmc_small_summarize.py <counter_collection.csv> <pairs> <out.json>
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES
counts cycles summed over SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs (kernel cycles = value / 8; a counter pass stretches short
dispatches, so ratios rather than absolute times are what to read)."""
import collections, csv, json, sys
path, pairs, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
grid = {}
for r in csv.DictReader(open(path)):
  acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
  grid[r["Kernel_Name"]] = (r.get("Grid_Size"), r.get("Workgroup_Size"), r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"))
res = {}
for k, d in acc.items():
  m = {c: sum(v) / len(v) for c, v in d.items()}
  n = len(next(iter(d.values())))
  cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8.0
  rec = {"dispatches": n, "grid": grid[k][0], "workgroup": grid[k][1], "lds_bytes": grid[k][2], "vgpr": grid[k][3], "agpr": grid[k][4],
         "kernel_cycles": round(cyc, 1), "waves": m.get("SQ_WAVES"),
         "mfma_busy_fraction_of_simd_cycles": round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024.0 * cyc), 4) if cyc else None,
         "wave_cycles_per_wave": round(4 * m.get("SQ_WAVE_CYCLES", 0) / max(m.get("SQ_WAVES", 1), 1), 1),
         "wait_any_share": round(m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), 4),
         "wait_inst_share": round(m.get("SQ_WAIT_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), 4),
         "active_inst_share": round(m.get("SQ_ACTIVE_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), 4),
         "wave_residency_of_kernel": round(4 * m.get("SQ_WAVE_CYCLES", 0) / max(m.get("SQ_WAVES", 1), 1) / cyc, 4) if cyc else None,
         "raw": {c: round(v, 1) for c, v in m.items()}}
  res[k[:110]] = rec
json.dump({"pairs_per_step": pairs, "command": "rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph --one-stream --batch %d" % pairs,
           "kernels": dict(sorted(res.items(), key=lambda kv: -kv[1]["kernel_cycles"] * kv[1]["dispatches"]))}, open(out, "w"), indent=1)
for k, v in list(json.load(open(out))["kernels"].items())[:40]:
  print("%-70s n=%4d cyc %8.0f waves %6.0f mfma %.3f wait %.2f inst-wait %.2f active %.2f resid %.2f" % (k[:70], v["dispatches"], v["kernel_cycles"], v["waves"] or 0, v["mfma_busy_fraction_of_simd_cycles"] or 0, v["wait_any_share"], v["wait_inst_share"], v["active_inst_share"], v["wave_residency_of_kernel"] or 0))
