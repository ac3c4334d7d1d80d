#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for v in 16 24 32 64 16 24 32 64; do
  export SDM_LIB_PATH=build/ab/libsdm_cks$v.so
  timeout 600 python bench.py --no-cpu --no-dense --no-strong --no-stress --no-grown --no-adapter > gpurun_out/m_bench.json 2> gpurun_out/m_bench.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/m_bench.json').read().strip().splitlines()[-1])
print('per-shard $v bench', d['ms_per_step'], d.get('stage_ms')['weight'])
PY
done
for v in 16 24 32 64; do
  export SDM_LIB_PATH=build/ab/libsdm_cks$v.so
  SDM_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_m -o m -- python bench.py --no-cpu --no-dense --no-strong --no-stress --no-grown --no-adapter > gpurun_out/m_prof.log 2>&1
  python tools/trace_db.py gpurun_out/prof_m/m_results.db 8 2>/dev/null | grep -E "^k_ck " | sed "s/^/per-shard $v: /"
  rm -rf gpurun_out/prof_m
done
