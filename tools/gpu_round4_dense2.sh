#!/bin/bash
# dense sweep A/B: the kernel as committed (old) against the fetch without a wait behind the slab stamps and without
# a branch per load, with one (dnew) or two (d2) register buffers; sweep and clear tests on the default build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/dense2
timeout 900 python -m pytest tests/test_sweep_dense_gpu.py tests/test_clear_gpu.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -5
for round in 1 2 3; do
  for tag in old dnew d2; do
    SDM_LIB_PATH=build/ab/libsdm_$tag.so timeout 300 python tools/probes/full_only.py --dense 2>&1 | tail -1
  done
done
for tag in old dnew d2; do
  SDM_LIB_PATH=build/ab/libsdm_$tag.so timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- python tools/probes/full_only.py --dense > gpurun_out/dense2/${tag}_prof.log 2>&1
  python tools/trace_db.py gpurun_out/prof_$tag/${tag}_results.db 0 > gpurun_out/dense2/${tag}_kernel_stats.txt 2>&1
  rm -rf gpurun_out/prof_$tag
  grep -E "k_occupancy_dense|k_occupancy_scan" gpurun_out/dense2/${tag}_kernel_stats.txt | head -4
done
