"""Merges gpurun_out/pmc_<tag>_p{1,2,4}.json (tests/tools/pmc_run.sh <tag>_p<N> <N>) into one file keyed by pairs per launch.
usage: python tests/tools/pmc_merge.py <tag> <out.json>"""
import json, os, sys
tag, out = sys.argv[1], sys.argv[2]
merged = {}
for n in (1, 2, 4):
  path = "gpurun_out/pmc_%s_p%d.json" % (tag, n)
  if os.path.exists(path):
    merged[str(n)] = json.load(open(path))
# the commit the counters were taken on (this script runs in the repository, after the GPU call; the box itself has no .git)
import subprocess
try:
  merged["git_commit"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
  merged["git_dirty"] = bool(subprocess.run(["git", "status", "--porcelain", "--", "adaptive-stereo-icra-2021_amd/csrc"], capture_output=True,
                                            text=True, check=True).stdout.strip())
except Exception:      # noqa: BLE001
  pass
json.dump(merged, open(out, "w"), indent=1)
print({k: sorted(v["kernels"].keys()) for k, v in merged.items() if k.isdigit()})
