#!/bin/bash
# fused sdm_clear: parity tests that use clear, then the bench line (roofline.clear)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/q
timeout 900 python -m pytest tests -m gpu -x -q -k "clear or wrap or stagewise" > gpurun_out/q/tests.log 2>&1
tail -5 gpurun_out/q/tests.log
timeout 600 python bench.py --no-adapter > gpurun_out/q/bench.json 2> gpurun_out/q/bench.err
python - <<'PY'
import json
l=[x for x in open('gpurun_out/q/bench.json') if x.startswith('{')][-1]
j=json.loads(l)
print(j['ms_per_step'], j['value'])
print(json.dumps(j['roofline'].get('clear')))
PY
