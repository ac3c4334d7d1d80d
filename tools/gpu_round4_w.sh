#!/bin/bash
# A/B: two-track vote path in the sweeps (dense cases), then parity tests on the main library (= "two")
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/w
for rep in 1 2; do
for tag in base two two3; do
  echo "== $tag"; SDM_LIB_PATH=build/ab/libsdm_$tag.so timeout 300 python tools/probes/dense_only.py 10 0 1 2>&1 | tail -2
done; done
timeout 1500 python -m pytest tests -m gpu -x -q -k "occup or sweep or dense or golden or parity or stagewise or config" > gpurun_out/w/tests.log 2>&1
tail -4 gpurun_out/w/tests.log
