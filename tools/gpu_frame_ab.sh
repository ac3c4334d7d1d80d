#!/bin/bash
# usage (GPU box): tools/gpu_frame_ab.sh <tag of an A/B build under build/ab> [pytest files ...]
# the parity tests named (default: the weight-update ones) on the default build, then the C3 frame time of the default build
# and of build/ab/libsdm_<tag>.so, twice each, alternating (variants are only comparable inside one call)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
tests=${@:-tests/test_parity_gpu.py tests/test_kat_gpu.py tests/test_golden.py tests/test_graph_gpu.py}
timeout 1200 python -m pytest $tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
for i in 1 2; do
  for lib in default $tag; do
    if [ $lib = default ]; then unset SDM_LIB_PATH; else export SDM_LIB_PATH=build/ab/libsdm_$lib.so; fi
    timeout 300 python bench.py --no-cpu --no-dense --no-strong --no-stress --no-driven --no-adapter 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['ms_per_step'], d['stage_ms'])"
  done
done
