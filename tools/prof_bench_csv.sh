#!/bin/bash
# usage (GPU box): tools/prof_bench_csv.sh <tag> -> gpurun_out/<tag>_kernel_stats.csv : rocprofv3's own --stats table
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profcsv_$tag -o $tag -- python bench.py --no-cpu --no-dense --no-strong --steps 20 --warmup 5 > gpurun_out/${tag}_profcsv_bench.log 2>&1
find gpurun_out/profcsv_$tag -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_kernel_stats.csv \;
rm -rf gpurun_out/profcsv_$tag
