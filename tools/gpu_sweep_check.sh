#!/bin/bash
# usage (GPU box): tools/gpu_sweep_check.sh [lib tags under build/ab ...]  ->  gpurun_out/sweep_check.txt
# parity tests of the occupancy sweeps and sdm_clear, then the times of the non-incremental sweep on the empty, the
# benchmark and the dense map - for the default build and for each A/B build named
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
{
  timeout 900 python -m pytest tests/test_sweep_dense_gpu.py tests/test_clear_gpu.py tests/test_kat_gpu.py tests/test_epoch_wrap_gpu.py tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -3
  timeout 300 python tools/probes/full_only.py --dense
  timeout 300 python tools/probes/dense_only.py 10 0 1
  for tag in "$@"; do
    echo "== $tag"
    SDM_LIB_PATH=build/ab/libsdm_$tag.so timeout 300 python -m pytest tests/test_sweep_dense_gpu.py -x -q -m gpu 2>&1 | tail -1
    SDM_LIB_PATH=build/ab/libsdm_$tag.so timeout 300 python tools/probes/full_only.py --dense
    SDM_LIB_PATH=build/ab/libsdm_$tag.so timeout 300 python tools/probes/dense_only.py 10 0 1
  done
} > gpurun_out/sweep_check.txt 2>&1
cat gpurun_out/sweep_check.txt
