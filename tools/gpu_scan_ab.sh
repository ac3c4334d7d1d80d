#!/bin/bash
# usage (GPU box): tools/gpu_scan_ab.sh [lib tags under build/ab ...]  ->  gpurun_out/scan_ab.txt
# the non-incremental sweep on the benchmark map, right after the load (tools/probes/full_only.py) and after the bench's
# frames (bench.py's roofline.full_evaluation), for the default build and each A/B build named
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
one() {
  timeout 300 python tools/probes/full_only.py
  timeout 600 python bench.py --no-cpu --no-stress --no-driven --no-adapter --no-strong --steps 20 --warmup 5 2>/dev/null | grep '"metric"' | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); r = j['roofline']
    print('  bench: frame', j['ms_per_step'], 'full', r['full_evaluation']['avg_launch_ms'], r['full_evaluation']['frac'], 'dense', r['dense_case']['avg_launch_ms'], 'live voxels', r.get('voxels_with_live_slots'))"
}
{
  timeout 600 python -m pytest tests/test_sweep_dense_gpu.py tests/test_kat_gpu.py tests/test_clear_gpu.py tests/test_configs_gpu.py tests/test_epoch_wrap_gpu.py -x -q -m gpu 2>&1 | tail -2
  one
  for tag in "$@"; do
    echo "== $tag"
    export SDM_LIB_PATH=build/ab/libsdm_$tag.so
    timeout 300 python -m pytest tests/test_sweep_dense_gpu.py tests/test_kat_gpu.py -x -q -m gpu 2>&1 | tail -1
    one
    unset SDM_LIB_PATH
  done
} > gpurun_out/scan_ab.txt 2>&1
cat gpurun_out/scan_ab.txt
