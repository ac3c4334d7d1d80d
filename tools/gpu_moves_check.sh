#!/bin/bash
# usage (GPU box): tools/gpu_moves_check.sh  ->  gpurun_out/moves_check.txt : the parity tests that move objects, then the
# driven leg of bench.py (frames rendered once and cached)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export SDM_DRIVEN_CACHE=/tmp/sdm_driven_frames_$$
{
  timeout 1200 python -m pytest tests/test_driven_gpu.py tests/test_long_object_lists_gpu.py tests/test_parity_gpu.py tests/test_sharded_gpu.py tests/test_sweep_dense_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
  timeout 900 python bench.py --only-driven 2>/dev/null | grep '"metric"\|driven' | python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    d = j.get('driven', j)
    print('driven: ms_per_step', d.get('ms_per_step'), 'stage', d.get('stage_ms'), 'full_evaluation', d.get('full_evaluation'))"
} > gpurun_out/moves_check.txt 2>&1
cat gpurun_out/moves_check.txt
