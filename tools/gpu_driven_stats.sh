#!/bin/bash
# usage (GPU box): tools/gpu_driven_stats.sh <tag>  ->  gpurun_out/<tag>_driven.json (the `driven` leg of bench.py) and
# gpurun_out/<tag>_driven_kernel_stats.txt (rocprofv3 --kernel-trace of the same command: per-kernel statistics over the drive
# from the empty map and the timed frames, and the launch timeline of one of the last frames)
set -u
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
# (the frames are rendered once, by the first run, and handed to the profiled run through a file: rocprofv3 attaches to
# every worker process of the renderer, which took the profiled run past its time limit twice)
export SDM_DRIVEN_CACHE=/tmp/sdm_driven_frames_$$
timeout 900 python bench.py --only-driven > gpurun_out/${tag}_driven.json 2> gpurun_out/${tag}_driven.err
SDM_GRAPH=0 timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_drv -o drv -- python bench.py --only-driven > gpurun_out/${tag}_driven_prof.log 2>&1
python tools/trace_db.py gpurun_out/prof_${tag}_drv/drv_results.db 6 > gpurun_out/${tag}_driven_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof_${tag}_drv $SDM_DRIVEN_CACHE.*.npy
python - <<PY
import json
d = json.loads(open("gpurun_out/${tag}_driven.json").read().strip().splitlines()[-1])["driven"]
print({k: d[k] for k in ("ms_per_step", "live_particles", "visible_particles_per_frame", "stage_ms", "x_cpu")})
PY
head -24 gpurun_out/${tag}_driven_kernel_stats.txt
