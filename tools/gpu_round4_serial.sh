#!/bin/bash
# chains of dependent loads taken apart (k_visibility, sweeps, birth replay; frame inputs as global memory): the GPU
# suite on the new build, frame time old / new in one run, kernel statistics of the new build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/serial
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/serial/gpu_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/serial/gpu_tests.log | tail -3
B="--no-cpu --no-dense --no-strong --no-adapter --no-grown --no-stress"
for round in 1 2 3; do
  for tag in old cur; do
    lib=build/ab/libsdm_$tag.so; [ $tag = cur ] && lib=semantic_dsp_map_amd/csrc/libsdm_hip.so
    SDM_LIB_PATH=$lib timeout 300 python bench.py $B --steps 300 --warmup 30 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$tag', j['ms_per_step'], j['stage_ms'])"
  done
done
for tag in old cur; do
  lib=build/ab/libsdm_$tag.so; [ $tag = cur ] && lib=semantic_dsp_map_amd/csrc/libsdm_hip.so
  SDM_GRAPH=0 SDM_LIB_PATH=$lib timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- python bench.py $B --steps 20 --warmup 5 > gpurun_out/serial/${tag}_prof.log 2>&1
  python tools/trace_db.py gpurun_out/prof_$tag/${tag}_results.db 8 > gpurun_out/serial/${tag}_kernel_stats.txt 2>&1
  rm -rf gpurun_out/prof_$tag
  echo == $tag; head -16 gpurun_out/serial/${tag}_kernel_stats.txt
done
