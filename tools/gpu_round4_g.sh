#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/g_pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/g_pytest.log | head -1; grep -E "^(FAILED|ERROR)|Error" gpurun_out/g_pytest.log | head -5
SDM_LIB_PATH=build/ab/libsdm_timers.so timeout 300 python tools/probes/timers.py 2>&1 | tail -23
for rep in 1 2; do
timeout 600 python bench.py --no-cpu --no-dense --no-strong --no-stress --no-grown > gpurun_out/g_bench$rep.json 2> gpurun_out/g_bench.err
python - <<PY
import json
d=json.loads(open('gpurun_out/g_bench$rep.json').read().strip().splitlines()[-1])
print('bench', d['ms_per_step'], d['value'], d['config']['live_particles'], d.get('stage_ms'))
PY
done
