#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s
timeout 600 python bench.py --no-dense --no-strong > gpurun_out/s/bench.json 2> gpurun_out/s/bench.err
python - <<'PY'
import json
l=[x for x in open('gpurun_out/s/bench.json') if x.startswith('{')][-1]
j=json.loads(l)
print(j['ms_per_step'], j['value'])
print(json.dumps({k:v for k,v in j['adapter_e2e'].items() if k in ('ms_per_update','min_ms_per_update','phase_ms','error')}))
PY
