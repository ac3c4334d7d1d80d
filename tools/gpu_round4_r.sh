#!/bin/bash
# A/B: fewer dependent steps in k_visibility and k_birth_replay
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r
for rep in 1 2; do
for tag in base new; do
  SDM_LIB_PATH=build/ab/libsdm_$tag.so timeout 300 python bench.py --no-cpu --no-dense --no-strong --no-adapter --steps 300 --warmup 30 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$tag', j['ms_per_step'], j.get('stage_ms'))"
done; done
for tag in base new; do
  SDM_LIB_PATH=build/ab/libsdm_$tag.so timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- python bench.py --no-cpu --no-dense --no-strong --no-adapter --steps 20 --warmup 5 > gpurun_out/r/${tag}_prof.log 2>&1
  python tools/trace_db.py gpurun_out/prof_$tag/${tag}_results.db 8 > gpurun_out/r/${tag}_kernel_stats.txt 2>&1
  rm -rf gpurun_out/prof_$tag
  grep -E "k_visibility|k_birth_replay" gpurun_out/r/${tag}_kernel_stats.txt | head -4
done
SDM_LIB_PATH=build/ab/libsdm_new.so timeout 900 python -m pytest tests -m gpu -x -q -k "parity or stagewise or golden" 2>&1 | tail -3
