#!/bin/bash
# device timeline of SemanticDSPMap::update (kernels + copies): the adapter driver alone under the tracer
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/u
SDM_ADAPTER_DIR=/tmp/ad timeout 300 python bench.py --no-dense --no-strong --steps 5 --warmup 2 > gpurun_out/u/bench.log 2>&1
ls -la /tmp/ad
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/prof_u -o ad -- /tmp/ad/adapter_e2e /tmp/ad/clip.bin /tmp/ad/out.bin time > gpurun_out/u/run.log 2>&1
tail -3 gpurun_out/u/run.log
find gpurun_out/prof_u -name "*.db" | while read f; do echo "== $f"; python tools/adapter_timeline.py "$f" > gpurun_out/u/timeline.txt 2>&1; tail -45 gpurun_out/u/timeline.txt; done
rm -rf gpurun_out/prof_u
