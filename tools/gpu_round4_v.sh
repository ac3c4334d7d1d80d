#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/v
timeout 1200 python -m pytest tests -m gpu -x -q -k "adapter or point or colour or color or owner or track or stats or emit or cloud or raw or binding" > gpurun_out/v/tests.log 2>&1
tail -4 gpurun_out/v/tests.log
SDM_ADAPTER_DIR=/tmp/ad timeout 300 python bench.py --no-dense --no-strong > gpurun_out/v/bench.json 2> gpurun_out/v/bench.err
python - <<'PY'
import json
l=[x for x in open('gpurun_out/v/bench.json') if x.startswith('{')][-1]
j=json.loads(l)
print(j['ms_per_step'], j['value'])
print(json.dumps({k:v for k,v in j['adapter_e2e'].items() if k in ('ms_per_update','min_ms_per_update','phase_ms','error')}))
PY
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/prof_v -o ad -- /tmp/ad/adapter_e2e /tmp/ad/clip.bin /tmp/ad/out.bin time > gpurun_out/v/run.log 2>&1
tail -3 gpurun_out/v/run.log
find gpurun_out/prof_v -name "*.db" | while read f; do python tools/adapter_timeline.py "$f" > gpurun_out/v/timeline.txt 2>&1; tail -22 gpurun_out/v/timeline.txt; done
rm -rf gpurun_out/prof_v
