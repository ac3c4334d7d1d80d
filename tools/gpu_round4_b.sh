#!/bin/bash
# GPU call: rocprofv3 kernel statistics + one frame's timeline of the bench frames, cross-frame gaps with in-kernel clocks
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
SDM_GRAPH=0 tools/prof_bench.sh b
head -30 gpurun_out/b_kernel_stats.txt; sed -n '/timeline of one frame/,$p' gpurun_out/b_kernel_stats.txt
for stop in move birth; do
  CROSSFRAME_STOP=$stop SDM_LIB_PATH=build/ab/libsdm_timers.so timeout 300 python tools/probes/crossframe.py 2 > gpurun_out/b_crossframe_$stop.txt 2>&1
  echo "== crossframe, frame 9 stops after $stop"; tail -8 gpurun_out/b_crossframe_$stop.txt
done
