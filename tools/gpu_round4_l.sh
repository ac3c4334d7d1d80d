#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 300 python tools/probes/two_maps.py 2>&1 | tail -6
timeout 1500 python bench.py > gpurun_out/l_bench.json 2> gpurun_out/l_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/l_bench.json').read().strip().splitlines()[-1])
print('C3', d['ms_per_step'], d['value'], 'live', d['config']['live_particles'], d['issue_mode'])
print('stage', d.get('stage_ms'))
r=d['roofline']; print('roofline', r['frac'], r['avg_launch_ms'], {k:(r[k]['frac'], r[k].get('avg_launch_ms', r[k].get('ms_per_call'))) for k in ('full_evaluation','dense_case','dense_case_surface','clear') if k in r}, r.get('survey_dense_frac'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['ms_per_frame'], 'x', round(d['value']/d['cpu_baseline']['value'],1))
for k in ('stress','grown','strong_scaling'):
    if k in d: print(k, d[k].get('ms_per_step'), d[k].get('x_cpu'), d[k].get('stage_ms'))
print('adapter', d.get('adapter_e2e'))
PY
tail -5 gpurun_out/l_bench.err
