#!/bin/bash
# usage (GPU box): tools/gpu_driven_sweep.sh [lib tags under build/ab ...]  ->  gpurun_out/driven_sweep.txt : the non-incremental sweep on a map grown by the
# `driven` scene, per launch and per kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export SDM_DRIVEN_CACHE=/tmp/sdm_driven_sweep_$$
{
  timeout 600 python tools/probes/driven_sweep.py
  for tag in "$@"; do SDM_LIB_PATH=build/ab/libsdm_$tag.so timeout 600 python tools/probes/driven_sweep.py; done
  rm -rf gpurun_out/dsw
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/dsw -o f -- python tools/probes/driven_sweep.py 2>&1 | grep driven_ms
  python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/dsw/**/f_kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "occupancy_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for name in ("k_occupancy_scan", "k_occupancy_listed", "k_occupancy_dense"):
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if name in r["Kernel_Name"]]
        print("  %-18s us per launch, grown map:" % name, " ".join("%.1f" % x for x in d[-8:]), "| empty map:", " ".join("%.1f" % x for x in d[2:5]))
PY
  rm -rf gpurun_out/dsw $SDM_DRIVEN_CACHE.*.npy
} > gpurun_out/driven_sweep.txt 2>&1
cat gpurun_out/driven_sweep.txt
