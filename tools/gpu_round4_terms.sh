#!/bin/bash
# the table requests of several terms in one round (k_ck heavy part: four terms, k_weight: U x ROUNDS): weight-update parity
# tests, then kernel statistics prev (the tree before) / cur in one run
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/terms
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_kat_gpu.py tests/test_configs_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
B="--no-cpu --no-dense --no-strong --no-adapter --no-grown"
for tag in prev cur prev cur; do
  lib=build/ab/libsdm_$tag.so; [ $tag = cur ] && lib=semantic_dsp_map_amd/csrc/libsdm_hip.so
  SDM_LIB_PATH=$lib timeout 300 python bench.py $B --steps 40 --warmup 10 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$tag', 'C3', j['ms_per_step'], 'stress', j['stress']['ms_per_step'], j['stage_ms'])"
done
for tag in prev cur; do
  lib=build/ab/libsdm_$tag.so; [ $tag = cur ] && lib=semantic_dsp_map_amd/csrc/libsdm_hip.so
  SDM_GRAPH=0 SDM_LIB_PATH=$lib timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- python bench.py $B --no-stress --steps 40 --warmup 10 > gpurun_out/terms/${tag}_prof.log 2>&1
  python tools/trace_db.py gpurun_out/prof_$tag/${tag}_results.db 8 > gpurun_out/terms/${tag}_kernel_stats.txt 2>&1
  rm -rf gpurun_out/prof_$tag
  echo == $tag; grep -E "^k_ck |^k_weight|^k_ck_classify|^k_visibility|^k_birth_replay" gpurun_out/terms/${tag}_kernel_stats.txt
done
