#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
SDM_LIB_PATH=build/ab/libsdm_timers.so timeout 300 python tools/probes/timers.py > gpurun_out/r04_in_kernel_timers.txt 2>&1
SDM_LIB_PATH=build/ab/libsdm_timers.so timeout 300 python tools/probes/crossframe.py 3 > gpurun_out/r04_crossframe.txt 2>&1
timeout 300 python tools/probes/two_maps.py > gpurun_out/r04_two_maps.txt 2>&1
tail -5 gpurun_out/r04_crossframe.txt; cat gpurun_out/r04_two_maps.txt | tail -6
