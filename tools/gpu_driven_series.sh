#!/bin/bash
# usage (GPU box): tools/gpu_driven_series.sh [kernel name fragment ...]  ->  gpurun_out/driven_series.txt
# the duration of every launch of the named kernels (default: the move stage's) over bench.py's `driven` leg, frame by frame,
# under rocprofv3 --kernel-trace: where a kernel's average over the drive comes from
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export SDM_DRIVEN_CACHE=/tmp/sdm_driven_frames_$$
timeout 900 python bench.py --only-driven > /dev/null 2>&1
SDM_GRAPH=0 timeout 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_series -o s -- python bench.py --only-driven > gpurun_out/driven_series.log 2>&1
python - "$@" > gpurun_out/driven_series.txt <<'PY'
import csv, glob, sys
names = sys.argv[1:] or ["k_move_replay", "k_move_apply", "k_frame_begin"]
rows = []
for f in glob.glob("gpurun_out/prof_series/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for n in names:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if n in r["Kernel_Name"]]
    print(n, "launches", len(d), "mean %.1f" % (sum(d) / max(len(d), 1)))
    for i in range(0, len(d), 20):
        print("  %3d: " % i + " ".join("%5.1f" % x for x in d[i:i + 20]))
PY
rm -rf gpurun_out/prof_series $SDM_DRIVEN_CACHE.*.npy
cat gpurun_out/driven_series.txt
