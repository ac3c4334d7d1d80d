#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/e_pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/e_pytest.log | head -2
for ck in terms lists terms lists; do
  SDM_CK=$ck timeout 600 python bench.py --no-cpu --no-dense --no-strong --no-stress --no-grown > gpurun_out/e_bench_$ck.json 2> gpurun_out/e_bench.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/e_bench_$ck.json').read().strip().splitlines()[-1])
print('bench SDM_CK=$ck', d['ms_per_step'], d['value'], d.get('stage_ms'))
PY
done
SDM_GRAPH=0 tools/prof_bench.sh e
head -24 gpurun_out/e_kernel_stats.txt
SDM_LIB_PATH=build/ab/libsdm_timers.so timeout 300 python tools/probes/timers.py 2>&1 | tail -24
