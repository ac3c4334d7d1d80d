#!/bin/bash
# full GPU suite on the tree with the new result-list kernels / getters / raw upload order, then the bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/t
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/t/tests.log 2>&1
tail -6 gpurun_out/t/tests.log
timeout 600 python bench.py --no-dense --no-strong > gpurun_out/t/bench.json 2> gpurun_out/t/bench.err
python - <<'PY'
import json
l=[x for x in open('gpurun_out/t/bench.json') if x.startswith('{')][-1]
j=json.loads(l)
print(j['ms_per_step'], j['value'])
print(json.dumps({k:v for k,v in j['adapter_e2e'].items() if k in ('ms_per_update','min_ms_per_update','phase_ms','error')}))
PY
