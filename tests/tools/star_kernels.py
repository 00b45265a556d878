"""Achieved HBM GB/s (algorithmic bytes / mean kernel time) of the HBM-bound kernels of SURVEY §8 rows a2, a4-a6, a8-a10
and TFLOP/s of the MFMA rows, from a rocprofv3 --kernel-trace of bench.py (B pairs per step at 375x1242, k=4).
usage: python tests/tools/star_kernels.py <kernel_trace.csv> <timed_steps> [B]"""
import collections, csv, sys
path, steps = sys.argv[1], int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
win = rows[adam[-2 * steps - 1] + 1:adam[-1] + 1]
H, W, Hc, Wc, Dc = 375, 1242, 24, 78, 12
F = B * 32 * Hc * Wc * 4; V = B * 32 * Dc * Hc * Wc * 4; Lg = B * Dc * Hc * Wc * 4; P = B * Hc * Wc * 4
I1 = B * H * W * 4; A = B * 32 * H * W * 4
# kernel substring -> (row, algorithmic bytes per launch, flops per launch); None = not applicable
table = [
  ("cost_volume_fwd", "a2 fwd", 2 * F + V, None), ("cost_volume_bwd", "a2 bwd", V + 2 * F, None),
  ("conv32_fwd_kernel<27>", "a3 conv3d fwd/dgrad (direct-load)", None, 2.0 * B * Dc * Hc * Wc * 1024 * 27),
  ("conv3d_lds_kernel", "a3 conv3d fwd/dgrad (LDS)", None, 2.0 * B * Dc * Hc * Wc * 1024 * 27),
  ("agg3d_kernel<0, 0,", "a3 layer 1: conv3d + moments (rolling window)", 2 * V, 2.0 * B * Dc * Hc * Wc * 1024 * 27),
  ("agg3d_kernel<2, 0,", "a3 layers 2-4: BN merge + act in LDS + conv3d + moments", 3 * V, 2.0 * B * Dc * Hc * Wc * 1024 * 27),
  ("agg3d_kernel<0, 1,", "a3 eval layer: conv3d + folded BN + LReLU", 2 * V, 2.0 * B * Dc * Hc * Wc * 1024 * 27),
  ("agg3d_kernel<0, 2,", "a3 conv3d data gradient (rolling window)", 2 * V, 2.0 * B * Dc * Hc * Wc * 1024 * 27),
  ("agg_tail_kernel", "a4+a5+a8 fused tail, LDS-staged (+ layer-4 BN/LReLU, by-product)", 2 * V + Lg + 3 * P, None),
  ("agg_tail_direct_kernel", "a4+a5+a8 fused tail (+ layer-4 BN/LReLU, by-product)", 2 * V + Lg + 3 * P, None),
  ("conv3d_wgrad_lds_kernel", "a3 conv3d wgrad (LDS)", None, 2.0 * B * Dc * Hc * Wc * 1024 * 27),
  ("agg_tail_bwd_kernel", "a4+a5 backward, one launch (softmax bwd + dgrad + wgrad)", 2 * V + Lg + P, None),
  ("conv32to1_fwd_kernel", "a4 conv3d_alone fwd", V + Lg, None),
  ("conv32to1_dgrad_kernel", "a4 dgrad", Lg + V, None), ("conv32to1_wgrad_kernel<27>", "a4 wgrad", V + Lg, None),
  ("softargmax_fwd", "a5+a8 softargmax+FCS fwd", Lg + 3 * P, None), ("softargmax_bwd", "a5 bwd", 2 * Lg + P, None),
  ("upsample_fwd", "a6/a7 bilinear up", P + I1, None), ("upsample_bwd", "a6/a7 bilinear up bwd", I1 + P, None),
  ("warp_fwd", "a9 warp fwd", 3 * I1 + I1 + 3 * I1 + I1 // 4, None), ("warp_bwd", "a9 warp bwd", 3 * I1 + 3 * I1 + 2 * I1, None),
  ("photo_rows_fwd_kernel<true, true>", "a9+a10 warp + loss + masked sum, one pass (row strips)", (1 + 3 + 3 + 3) * I1 + I1 // 4, None),
  ("photo_rows_bwd_kernel", "a9+a10 backward, one pass (row strips)", (1 + 3 + 3 + 2) * I1, None),
  ("rows_mean_term_kernel", "a10 backward: per-image mean term", 3 * I1, None),
  ("image_sum_kernel", "a10 per-image mean disparity", I1, None),
  ("monodepth_fwd", "a10 loss fwd", (1 + 3 + 3 + 1) * I1, None),
  ("monodepth_bwd_a", "a10 loss bwd A", (1 + 1 + 3 + 3 + 10) * I1, None), ("monodepth_bwd_b", "a10 loss bwd B", (1 + 1 + 3 + 3 + 10 + 1 + 3) * I1, None),
  ("conv32_lds_kernel<0, false>", "a7 conv 3x3 fwd (LDS)", 2 * A, 2.0 * B * H * W * 1024 * 9),
  ("conv32_lds_kernel<2, true>", "a7 conv 3x3 dgrad+skip (LDS)", 3 * A, 2.0 * B * H * W * 1024 * 9),
  ("conv32_wgrad_lds", "a7 conv 3x3 wgrad (LDS)", 2 * A, 2.0 * B * H * W * 1024 * 9),
  ("bn_act_fwd_kernel<true>", "a7/a1 BN+LReLU+skip fwd (mixed sizes)", None, None),
]
agg = collections.defaultdict(lambda: [0, 0])
for r in win:
  d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
  agg[r['Kernel_Name']][0] += d; agg[r['Kernel_Name']][1] += 1
print("per-launch means over the %d timed steps, %d pairs per launch; HBM peak 8000 GB/s, fp32 MFMA peak 157.3 TFLOP/s" % (steps, B))
print("%-38s %9s %10s %9s %8s" % ("row / kernel", "us", "GB/s", "of peak", "TFLOP/s"))
for sub, row, by, fl in table:
  hits = [(k, v) for k, v in agg.items() if sub in k]
  if not hits:
    continue
  ns = sum(v[0] for _, v in hits); n = sum(v[1] for _, v in hits)
  us = ns / n / 1e3
  gbs = "%10.0f" % (by / (us * 1e-6) / 1e9) if by else "%10s" % "-"
  frac = "%8.1f%%" % (100 * by / (us * 1e-6) / 8e12) if by else "%9s" % "-"
  tf = "%8.1f" % (fl / (us * 1e-6) / 1e12) if fl else "%8s" % "-"
  print("%-38s %9.1f %s %s %s" % (row, us, gbs, frac, tf))
