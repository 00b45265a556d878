"""Summarises the three rocprofv3 --pmc passes of tests/tools/pmc_conv.py into profiles/pmc_conv32_lds.json.
usage: python tests/tools/pmc_summarize.py <sq.csv> <fetch.csv> <write.csv> <out.json> [pairs per launch, default 4]
HBM bytes follow MI355X_MICROARCH.md (HBM / rocprofv3): on gfx950 FETCH_SIZE tallies 64 B per 128-B request
of a wide coalesced stream -> read bytes = 2 x FETCH_SIZE(KB) x 1024; WRITE_SIZE(KB) x 1024 is exact.
MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles): MFMA_BUSY is the chip-wide sum
of per-SIMD matrix-pipe cycles (it equals launches' MFMA count x 64 exactly for v_mfma_f32_32x32x2_f32);
kernel cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the counter over the 8 XCDs)."""
import collections, csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import csrc_digest

def load(path):
  acc = collections.defaultdict(lambda: collections.defaultdict(list))
  for r in csv.DictReader(open(path)):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
  return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}

sq, fe, wr = load(sys.argv[1]), load(sys.argv[2]), load(sys.argv[3])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 4
H, W = 375, 1242
vox = B * H * W
flops = 2.0 * vox * 1024 * 9
vox3 = B * 12 * 24 * 78
alg = {"conv32_lds_kernel<0, false>": 2 * vox * 128, "conv32_lds_kernel<2, true>": 3 * vox * 128,
       "conv32_lds_kernel<3, true>": 4 * vox * 128,            # + the next layer's pre-activation (fused BN-backward sums)
       "conv32_wgrad_lds_kernel": 2 * vox * 128, "conv32_wgrad_lds2_kernel<false>": 2 * vox * 128,
       "conv32_wgrad_lds2_kernel<true>": 4 * vox * 128,        # x, g_a, z read; g_z written (fused BN-backward apply)
       "conv32_act_kernel<true>": 4 * vox * 128,               # z_prev, a_prevprev read; a_prev (by-product), z written
       "conv32_bwd_fused_kernel": 5 * vox * 128,               # x, g_a, z, z_next read; g_x written (2 x the flops)
       "conv32_wino_kernel<1, 0>": 4 * vox * 128,              # minimal filtering: z_prev, a_prevprev read; a_prev, z written
       "conv32_wino_kernel<0, 0>": 3 * vox * 128,              # ... no skip input
       "conv32_wino_kernel<2, 0>": 5 * vox * 128,              # g_a, z, z_next read; g_z (by-product), g_x written
       "conv32_wino_dgrad_kernel<0>": 5 * vox * 128,           # second generation of the data gradient (conv32_wino_dgrad.hip)
       "conv32_wino_wgrad_kernel<0>": 2 * vox * 128,           # x, g_z read
       "conv32_wino_bwd_kernel<0>": 5 * vox * 128,             # both gradients in one launch: x, g_a, z, z_next read; g_x written
       "conv32_fwd_kernel<27>": 2 * vox3 * 128 + 27 * 4096, "conv32_wgrad_kernel<3>": 2 * vox3 * 128,
       "conv3d_lds_kernel": 2 * vox3 * 128 + 27 * 4096, "conv3d_wgrad_lds_kernel": 2 * vox3 * 128,
       "agg3d_kernel<0, 2, false>": 2 * vox3 * 128 + 27 * 4096, "agg3d_kernel<0, 0, false>": 2 * vox3 * 128 + 27 * 4096,
       "agg3d_kernel<2, 0, false>": 3 * vox3 * 128 + 27 * 4096,      # raw operand read, activated by-product and output written
       "agg_tail_kernel<2, 8, true>": 2 * vox3 * 128 + vox3 * 4,
       "agg_tail_direct_kernel<2, true, 16>": 2 * vox3 * 128 + vox3 * 4,      # second generation of the tail (direct loads)
       "agg_tail_direct_kernel<2, true, 32>": 2 * vox3 * 128 + vox3 * 4}
out = {"shape": "2-D 3x3 stride 1, 32->32, %d pair(s) x 375x1242 (one full-resolution refinement layer); 3-D rows: 12x24x78 per pair" % B,
       "pairs_per_launch": B, "algorithmic_flops_per_launch": flops,
       "csrc_digest": csrc_digest(),       # sha256 over csrc/*.hip, *.h as they were when the counters were taken: bench.py refuses stale files
       "command": "rocprofv3 --kernel-trace --pmc <counters> -- python3 tests/tools/pmc_conv.py   (separate passes: SQ_* ; FETCH_SIZE ; WRITE_SIZE GRBM_GUI_ACTIVE)",
       "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced streams -> read bytes = 2 x FETCH_SIZE x 1024 (MI355X_MICROARCH.md, HBM); WRITE_SIZE x 1024 exact.",
       "kernels": {}}
for name in sq:
  key = next((k for k in alg if name.startswith("void " + k) or name.startswith(k)), None)
  if key is None:
    continue
  f = next(v for k, v in fe.items() if k == name)
  w = next(v for k, v in wr.items() if k == name)
  s = sq[name]
  rec = {"algorithmic_bytes_per_launch": alg[key], "FETCH_SIZE_KB": f["FETCH_SIZE"], "WRITE_SIZE_KB": w["WRITE_SIZE"],
         "hbm_bytes_per_launch": int(2 * f["FETCH_SIZE"] * 1024 + w["WRITE_SIZE"] * 1024),
         "SQ_VALU_MFMA_BUSY_CYCLES": s["SQ_VALU_MFMA_BUSY_CYCLES"],
         "mfma_busy_fraction_of_simd_cycles": round(s["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * w["GRBM_GUI_ACTIVE"] / 8.0), 4),
         "SQ_LDS_BANK_CONFLICT": s.get("SQ_LDS_BANK_CONFLICT"), "SQ_LDS_IDX_ACTIVE": s.get("SQ_LDS_IDX_ACTIVE"),
         "GRBM_GUI_ACTIVE": w.get("GRBM_GUI_ACTIVE")}
  rec["traffic_over_algorithmic"] = round(rec["hbm_bytes_per_launch"] / alg[key], 3)
  out["kernels"][key] = rec
main = out["kernels"].get("conv32_lds_kernel<0, false>", {})
out["kernel"] = "conv32_lds_kernel<0, false>"
out["hbm_bytes_per_launch"] = main.get("hbm_bytes_per_launch")
out["algorithmic_bytes_per_launch"] = main.get("algorithmic_bytes_per_launch")
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out, indent=1))
