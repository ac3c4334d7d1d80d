#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
SDM_GRAPH=0 tools/prof_bench.sh h
head -30 gpurun_out/h_kernel_stats.txt; sed -n '/timeline of one frame/,$p' gpurun_out/h_kernel_stats.txt
SDM_LIB_PATH=build/ab/libsdm_timers.so timeout 300 python tools/probes/timers.py 2>&1 | tail -23
