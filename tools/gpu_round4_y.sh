#!/bin/bash
# A/B: alias table walked eight entries per round (births, move replay, removals)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/y
B="--no-cpu --no-dense --no-strong --no-adapter --no-grown --no-stress"
for rep in 1 2 3; do
for tag in base alias; do
  SDM_LIB_PATH=build/ab/libsdm_$tag.so timeout 300 python bench.py $B --steps 300 --warmup 30 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$tag', j['ms_per_step'])"
done; done
for tag in base alias; do
  SDM_LIB_PATH=build/ab/libsdm_$tag.so timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- python bench.py $B --steps 20 --warmup 5 > gpurun_out/y/${tag}_prof.log 2>&1
  python tools/trace_db.py gpurun_out/prof_$tag/${tag}_results.db 8 > gpurun_out/y/${tag}_kernel_stats.txt 2>&1
  rm -rf gpurun_out/prof_$tag
  grep -E "k_birth_replay|k_move_replay|k_remove" gpurun_out/y/${tag}_kernel_stats.txt | head -3
done
timeout 1200 python -m pytest tests -m gpu -x -q -k "fuzz or alias or owner or parity or removal or edge" 2>&1 | tail -3
