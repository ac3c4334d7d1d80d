#!/bin/bash
# usage (GPU box): tools/gpu_dense_ab.sh <tags of A/B builds under build/ab ...>  ->  gpurun_out/dense_ab.txt
# the non-incremental sweep on the dense case (both track patterns), the empty map and the benchmark map, default build and
# every tag, three rounds alternating (variants are only comparable inside one call); then the sweep's parity tests on the
# last tag
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
{
  for i in 1 2 3; do
    for lib in default "$@"; do
      if [ $lib = default ]; then unset SDM_LIB_PATH; else export SDM_LIB_PATH=build/ab/libsdm_$lib.so; fi
      echo "== $lib round $i"
      timeout 300 python tools/probes/dense_only.py 10 0 1 2>&1 | grep "dense mode"
      timeout 300 python tools/probes/full_only.py 2>&1 | tail -1
    done
  done
  last="${@: -1}"
  [ -n "$last" ] && SDM_LIB_PATH=build/ab/libsdm_$last.so timeout 900 python -m pytest tests/test_sweep_dense_gpu.py tests/test_clear_gpu.py tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -3
} > gpurun_out/dense_ab.txt 2>&1
cat gpurun_out/dense_ab.txt
