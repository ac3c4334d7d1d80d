#!/bin/bash
# the replay of moved copies without scratch memory / with its batch fetched in one round: move-related parity tests, then
# frame time and kernel statistics old / cur in one run (default steps: a bench.py run renders its frames on the host)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/serial2
timeout 900 python -m pytest tests/test_fuzz_gpu.py tests/test_parity_gpu.py tests/test_owner_tracks_gpu.py tests/test_kat_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
B="--no-cpu --no-dense --no-strong --no-adapter --no-grown --no-stress"
for tag in old cur; do
  lib=build/ab/libsdm_$tag.so; [ $tag = cur ] && lib=semantic_dsp_map_amd/csrc/libsdm_hip.so
  SDM_GRAPH=0 SDM_LIB_PATH=$lib timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- python bench.py $B --steps 40 --warmup 10 > gpurun_out/serial2/${tag}_prof.log 2>&1
  python tools/trace_db.py gpurun_out/prof_$tag/${tag}_results.db 8 > gpurun_out/serial2/${tag}_kernel_stats.txt 2>&1
  rm -rf gpurun_out/prof_$tag
  echo == $tag; grep '"metric"' gpurun_out/serial2/${tag}_prof.log | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('ms_per_step (under rocprof)', j['ms_per_step'])"
  head -18 gpurun_out/serial2/${tag}_kernel_stats.txt
done
