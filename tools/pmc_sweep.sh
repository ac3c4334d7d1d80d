#!/bin/bash
# usage (GPU box): tools/pmc_sweep.sh <tag> -> gpurun_out/<tag>_sweep_{FETCH_SIZE,WRITE_SIZE}.csv + <tag>_sweep_bench.json
# HBM traffic of the occupancy sweep launches of the bench command: the in-frame launches (k_occupancy) and the
# non-incremental / dense-case launches (k_occupancy_scan + k_occupancy_dense).  FETCH_SIZE and WRITE_SIZE need separate passes (3 + 2 of the
# 4 TCC slots); no other trace domains are enabled.  SDM_GRAPH=0: every kernel is its own dispatch.
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  SDM_GRAPH=0 timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc $c -d gpurun_out/pmc_${tag}_$c -o s -- python bench.py --no-cpu --no-strong --no-stress --steps 20 --warmup 5 > gpurun_out/${tag}_sweep_$c.log 2>&1
  grep -E "Kernel_Name|k_occupancy" gpurun_out/pmc_${tag}_$c/s_counter_collection.csv > gpurun_out/${tag}_sweep_$c.csv
  rm -rf gpurun_out/pmc_${tag}_$c
done
grep '"metric"' gpurun_out/${tag}_sweep_FETCH_SIZE.log | tail -1 > gpurun_out/${tag}_sweep_bench.json
python tools/pmc_sweep_json.py $tag
