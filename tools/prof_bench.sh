#!/bin/bash
# usage: tools/prof_bench.sh <tag>   (on the GPU box) -> gpurun_out/<tag>_kernel_stats.txt, <tag>_bench.json
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- python bench.py --no-cpu --no-dense --no-strong --steps 20 --warmup 5 > gpurun_out/${tag}_prof_bench.log 2>&1
python tools/trace_db.py gpurun_out/prof_$tag/${tag}_results.db 8 > gpurun_out/${tag}_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof_$tag
