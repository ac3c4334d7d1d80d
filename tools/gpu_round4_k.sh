#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for g in 2 0; do
echo "== SDM_GRAPH=$g"
SDM_GRAPH=$g SDM_LIB_PATH=build/ab/libsdm_timers.so timeout 300 python tools/probes/crossframe.py 4 2>&1 | grep -E "^map|issue" | cut -c1-110
done
