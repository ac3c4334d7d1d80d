#!/bin/bash
# A/B: particles per wave (U) and grid of k_weight
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/z
B="--no-cpu --no-dense --no-strong --no-adapter --no-grown --no-stress"
for tag in base u2 u3 g1280 base u2; do
  SDM_LIB_PATH=build/ab/libsdm_$tag.so timeout 300 python bench.py $B --steps 300 --warmup 30 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$tag', j['ms_per_step'])"
done
for tag in base u2 u3 g1280; do
  SDM_LIB_PATH=build/ab/libsdm_$tag.so timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- python bench.py $B --steps 20 --warmup 5 > gpurun_out/z/${tag}_prof.log 2>&1
  python tools/trace_db.py gpurun_out/prof_$tag/${tag}_results.db 8 > gpurun_out/z/${tag}_kernel_stats.txt 2>&1
  rm -rf gpurun_out/prof_$tag
  grep -E "k_weight" gpurun_out/z/${tag}_kernel_stats.txt | head -1
done
