#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for c in 1 0; do for rep in 1 2 3; do
  echo "== SDM_STREAM_CACHE=$c process $rep"
  SDM_STREAM_CACHE=$c SDM_LIB_PATH=build/ab/libsdm_timers.so timeout 300 python tools/probes/crossframe.py 4 2>&1 | grep -E "^map" | cut -c1-120
done; done
timeout 300 python tools/probes/two_maps.py 2>&1 | tail -12
timeout 900 python -m pytest -x -q -m gpu tests/test_fuzz_gpu.py tests/test_parity_edge_gpu.py tests/test_graph_gpu.py tests/test_sharded_gpu.py > gpurun_out/j_pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/j_pytest.log | head -1
SDM_GRAPH=0 tools/prof_bench.sh j
grep -E "k_bin_rows|k_ck_classify|k_frame_begin|k_move" gpurun_out/j_kernel_stats.txt | head
timeout 600 python bench.py --no-cpu --no-dense --no-strong --no-stress --no-grown > gpurun_out/j_bench.json 2> gpurun_out/j_bench.err
python - <<PY
import json
d=json.loads(open('gpurun_out/j_bench.json').read().strip().splitlines()[-1])
print('bench', d['ms_per_step'], d['value'], d.get('stage_ms'))
PY
