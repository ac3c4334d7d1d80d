#!/bin/bash
# usage (GPU box): tools/gpu_frame_ab3.sh <tag of an A/B build under build/ab> : C3 frame and busy-scene frame of the default
# build against build/ab/libsdm_<tag>.so, two rounds alternating (variants are only comparable inside one call)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1
for i in 1 2; do
  for lib in default $tag; do
    if [ $lib = default ]; then unset SDM_LIB_PATH; else export SDM_LIB_PATH=build/ab/libsdm_$lib.so; fi
    timeout 300 python bench.py --no-dense --no-strong --no-driven --no-adapter --cpu-frames 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); st = d.get('stress') or {}
print('$lib', d['ms_per_step'], 'birth', d['stage_ms']['birth'], '| stress', st.get('ms_per_step'), (st.get('stage_ms') or {}).get('birth'))"
  done
done
