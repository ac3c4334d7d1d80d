#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export SDM_LIB_PATH=build/ab/libsdm_timers.so
for q in default 2 8 16; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  echo "== GPU_MAX_HW_QUEUES=$q"
  timeout 300 python tools/probes/crossframe.py 3 2>&1 | grep -E "^map|k_frame_begin" 
done
unset GPU_MAX_HW_QUEUES
echo "== queue ids under rocprofv3 (default)"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/c_trace -o c -- python tools/probes/crossframe.py 3 > gpurun_out/c_trace.log 2>&1
f=$(find gpurun_out/c_trace -name "*kernel_trace.csv" | head -1)
python tools/probes/queue_map.py $f
grep -E "^map" gpurun_out/c_trace.log
rm -rf gpurun_out/c_trace
