#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest -x -q -m gpu tests/test_kat_gpu.py tests/test_parity_gpu.py tests/test_parity_edge_gpu.py tests/test_fuzz_gpu.py tests/test_graph_gpu.py tests/test_sharded_gpu.py > gpurun_out/n_pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/n_pytest.log | head -1; grep -E "^(FAILED|ERROR)" gpurun_out/n_pytest.log | head
for ck in terms lists terms lists; do
  SDM_CK=$ck timeout 600 python bench.py --no-cpu --no-dense --no-strong --no-stress --no-grown --no-adapter > gpurun_out/n_bench.json 2> gpurun_out/n_bench.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/n_bench.json').read().strip().splitlines()[-1])
print('bench SDM_CK=$ck', d['ms_per_step'], d['value'], d.get('stage_ms'))
PY
done
SDM_GRAPH=0 tools/prof_bench.sh n
head -24 gpurun_out/n_kernel_stats.txt
