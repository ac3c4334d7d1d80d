#!/bin/bash
# usage (GPU box): tools/gpu_dense_split.sh  ->  gpurun_out/dense_split.txt : per-kernel times of the non-incremental sweep on the
# dense case (rocprofv3 --kernel-trace --stats), the streaming reference points of tools/probes/ring_probe on the same box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
{
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dsplit -o d -- python tools/probes/dense_only.py 10 0 2>&1 | grep "dense mode"
  python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/dsplit/**/d_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "occupancy" in r["Name"]:
            print("%-60s calls %s avg %.1f us min %.1f max %.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  [ -x tools/probes/ring_probe ] && timeout 120 tools/probes/ring_probe | head -3
} > gpurun_out/dense_split.txt 2>&1
cat gpurun_out/dense_split.txt
