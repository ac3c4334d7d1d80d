#!/bin/bash
# usage (GPU box): tools/gpu_driven_ab.sh <tags of A/B builds under build/ab ...>  ->  gpurun_out/driven_ab.txt
# bench.py's `driven` leg and its headline frame, default build and every tag, two rounds alternating (the frames rendered once)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export SDM_DRIVEN_CACHE=/tmp/sdm_driven_frames_$$
{
  for i in 1 2; do
    for lib in default "$@"; do
      if [ $lib = default ]; then unset SDM_LIB_PATH; else export SDM_LIB_PATH=build/ab/libsdm_$lib.so; fi
      timeout 600 python bench.py --only-driven 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])['driven']; print('$lib driven', d['ms_per_step'], d['stage_ms'])"
      timeout 300 python bench.py --no-cpu --no-dense --no-strong 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib c3', d['ms_per_step'], d['stage_ms'])"
    done
  done
  rm -f $SDM_DRIVEN_CACHE.*.npy
} > gpurun_out/driven_ab.txt 2>&1
cat gpurun_out/driven_ab.txt
