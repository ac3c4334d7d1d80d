#!/bin/bash
# the round's tracked profiles from the final tree, smoke(), and the full GPU suite
cd "$GRAFT_REPO_ROOT"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
tools/round_profile.sh r04 > gpurun_out/r04_round_profile.log 2>&1
tail -12 gpurun_out/r04_round_profile.log
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gpu_tests.log 2>&1
tail -3 gpurun_out/r04_gpu_tests.log
