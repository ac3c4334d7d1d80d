#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest -x -q -m gpu tests/test_fuzz_gpu.py tests/test_parity_edge_gpu.py tests/test_kat_gpu.py tests/test_graph_gpu.py tests/test_sharded_gpu.py tests/test_parity_gpu.py > gpurun_out/i_pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/i_pytest.log | head -1
SDM_GRAPH=0 tools/prof_bench.sh i
head -28 gpurun_out/i_kernel_stats.txt
for rep in 1 2; do
timeout 600 python bench.py --no-cpu --no-dense --no-strong --no-stress --no-grown > gpurun_out/i_bench$rep.json 2> gpurun_out/i_bench.err
python - <<PY
import json
d=json.loads(open('gpurun_out/i_bench$rep.json').read().strip().splitlines()[-1])
print('bench', d['ms_per_step'], d['value'], d['config']['live_particles'], d.get('stage_ms'))
PY
done
