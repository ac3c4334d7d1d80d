#!/bin/bash
# A/B: k_bin_rows with two workgroups per CU (640 threads, 96 registers)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/x
for rep in 1 2 3; do
for tag in base rows; do
  SDM_LIB_PATH=build/ab/libsdm_$tag.so timeout 300 python bench.py --no-cpu --no-dense --no-strong --no-adapter --no-grown --steps 300 --warmup 30 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$tag', j['ms_per_step'], j['stress']['ms_per_step'] if 'stress' in j else None)"
done; done
for tag in base rows; do
  SDM_LIB_PATH=build/ab/libsdm_$tag.so timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- python bench.py --no-cpu --no-dense --no-strong --no-adapter --no-stress --no-grown --steps 20 --warmup 5 > gpurun_out/x/${tag}_prof.log 2>&1
  python tools/trace_db.py gpurun_out/prof_$tag/${tag}_results.db 8 > gpurun_out/x/${tag}_kernel_stats.txt 2>&1
  rm -rf gpurun_out/prof_$tag
  grep -E "k_bin_rows|k_ck_classify|k_visibility" gpurun_out/x/${tag}_kernel_stats.txt | head -3
done
timeout 1500 python -m pytest tests -m gpu -x -q -k "wide or bins or parity or stagewise or golden or edge or other_window" 2>&1 | tail -3
