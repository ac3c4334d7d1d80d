#!/bin/bash
# GPU call: parity suite on the new move stage, in-kernel timers, cross-frame gap, a short bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/a_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/a_pytest.log
tail -5 gpurun_out/a_pytest.log
SDM_LIB_PATH=build/ab/libsdm_timers.so timeout 300 python tools/probes/timers.py > gpurun_out/a_timers.txt 2>&1
tail -32 gpurun_out/a_timers.txt
SDM_LIB_PATH=build/ab/libsdm_timers.so timeout 300 python tools/probes/crossframe.py 3 > gpurun_out/a_crossframe.txt 2>&1
cat gpurun_out/a_crossframe.txt | tail -4
timeout 600 python bench.py --no-cpu --no-dense --no-strong > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/a_bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d.get('stage_ms'))
PY
