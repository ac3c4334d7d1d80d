#!/bin/bash
# usage (GPU box): tools/gpu_sharded_ab.sh <tags of A/B builds under build/ab ...>  ->  gpurun_out/sharded_ab.txt
# the plain C3 frame and the one-rank sharded frame (RCCL, peer-memory exchange) of bench.py, default build and every tag,
# two rounds alternating (variants are only comparable inside one call)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
{
  for i in 1 2; do
    for lib in default "$@"; do
      if [ $lib = default ]; then unset SDM_LIB_PATH; else export SDM_LIB_PATH=build/ab/libsdm_$lib.so; fi
      timeout 400 python bench.py --no-cpu --no-dense --no-strong --sharded-leg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['sharded_one_rank']
print('$lib plain', d['ms_per_step'], 'rccl', s['ms_per_step'], s['exchanges_us'], 'ipc', s['ipc_exchange']['ms_per_step'], s['ipc_exchange']['exchanges_us'])"
    done
  done
} > gpurun_out/sharded_ab.txt 2>&1
cat gpurun_out/sharded_ab.txt
