#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest -x -q -m gpu tests/test_fuzz_gpu.py tests/test_parity_edge_gpu.py tests/test_kat_gpu.py tests/test_graph_gpu.py tests/test_sharded_gpu.py tests/test_owner_tracks_gpu.py tests/test_adapter_parity.py tests/test_replay.py > gpurun_out/d_pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/d_pytest.log
for pr in 0 1; do
  for rep in 1 2; do
    echo "== SDM_MAIN_PRIORITY=$pr process $rep"
    SDM_MAIN_PRIORITY=$pr SDM_LIB_PATH=build/ab/libsdm_timers.so timeout 300 python tools/probes/crossframe.py 3 2>&1 | grep -E "^map|frame_begin"
  done
done
SDM_LIB_PATH=build/ab/libsdm_timers.so timeout 300 python tools/probes/timers.py 2>&1 | tail -22 | head -12
for rep in 1 2; do
timeout 600 python bench.py --no-cpu --no-dense --no-strong > gpurun_out/d_bench$rep.json 2> gpurun_out/d_bench.err
python - <<PY
import json
d=json.loads(open('gpurun_out/d_bench$rep.json').read().strip().splitlines()[-1])
print('bench', d['ms_per_step'], d['value'], d.get('stage_ms'))
PY
done
