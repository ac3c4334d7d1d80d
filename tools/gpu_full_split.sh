#!/bin/bash
# usage (GPU box): tools/gpu_full_split.sh [lib tags under build/ab ...]  ->  gpurun_out/full_split.txt : per-launch times of
# the two kernels of the non-incremental sweep on the empty map (first 11 launches) and on the benchmark map (last 11)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
one() {
  rm -rf gpurun_out/fsplit
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fsplit -o f -- python tools/probes/full_only.py 2>&1 | grep full_ms
  python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/fsplit/**/f_kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "occupancy" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for name in ("k_occupancy_scan", "k_occupancy_listed", "k_occupancy_dense"):
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if name in r["Kernel_Name"]]
        print("  %-18s us per launch:" % name, " ".join("%.1f" % x for x in d[-8:]), "| empty map:", " ".join("%.1f" % x for x in d[2:5]))
PY
  rm -rf gpurun_out/fsplit
}
{
  one
  for tag in "$@"; do
    echo "== $tag"
    SDM_LIB_PATH=build/ab/libsdm_$tag.so one
  done
} > gpurun_out/full_split.txt 2>&1
cat gpurun_out/full_split.txt
