#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest -x -q -m gpu tests/test_kat_gpu.py tests/test_parity_gpu.py tests/test_parity_edge_gpu.py tests/test_fuzz_gpu.py tests/test_graph_gpu.py tests/test_sharded_gpu.py > gpurun_out/o_pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/o_pytest.log | head -1; grep -E "^(FAILED|ERROR)" gpurun_out/o_pytest.log | head
SDM_LIB_PATH=build/ab/libsdm_timers.so timeout 300 python tools/probes/timers.py 2>&1 | tail -24 | grep -E "ck_sum|slowest|bin_sort|      ck |weight  "
for ck in terms lists terms lists; do
  SDM_CK=$ck timeout 600 python bench.py --no-cpu --no-dense --no-strong --no-stress --no-grown --no-adapter > gpurun_out/o_bench.json 2> gpurun_out/o_bench.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/o_bench.json').read().strip().splitlines()[-1])
print('bench SDM_CK=$ck', d['ms_per_step'], d['value'], d.get('stage_ms'))
PY
done
SDM_GRAPH=0 tools/prof_bench.sh o
grep -E "k_ck|k_weight|k_bin_rows" gpurun_out/o_kernel_stats.txt | head
