#!/bin/bash
# usage (GPU box): tools/gpu_skip_scan_ab.sh  ->  gpurun_out/skip_scan_ab.txt
# the non-incremental sweep of a map whose every group is dense as one launch (default) against two
# (SDM_SWEEP_SKIP_SCAN=0), five rounds alternating in one call; the empty and the benchmark map beside it (both always
# take two launches: nothing may change there); then the sweep's parity tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
{
  timeout 900 python -m pytest tests/test_sweep_dense_gpu.py -x -q -m gpu 2>&1 | tail -3
  for i in 1 2 3 4 5; do
    for skip in 1 0; do
      echo "== SDM_SWEEP_SKIP_SCAN=$skip round $i"
      SDM_SWEEP_SKIP_SCAN=$skip timeout 300 python tools/probes/dense_only.py 10 0 1 2>&1 | grep "dense mode"
      SDM_SWEEP_SKIP_SCAN=$skip timeout 300 python tools/probes/full_only.py 2>&1 | tail -1
    done
  done
  timeout 900 python -m pytest tests/test_clear_gpu.py tests/test_configs_gpu.py tests/test_parity_gpu.py -x -q -m gpu 2>&1 | tail -3
} > gpurun_out/skip_scan_ab.txt 2>&1
cat gpurun_out/skip_scan_ab.txt
