#!/bin/bash
# usage (GPU box): tools/pmc_sweep.sh <tag> -> gpurun_out/<tag>_sweep_{fetch,write}.csv + <tag>_sweep.log
# FETCH_SIZE and WRITE_SIZE need separate passes (3 + 2 of the 4 TCC slots).
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $c -d gpurun_out/pmc_${tag}_$c -o s -- python tools/sweep_only.py 10 > gpurun_out/${tag}_sweep_$c.log 2>&1
  grep -E "Kernel_Name|k_occupancy" gpurun_out/pmc_${tag}_$c/s_counter_collection.csv > gpurun_out/${tag}_sweep_$c.csv
  rm -rf gpurun_out/pmc_${tag}_$c
done
grep "sweep avg" gpurun_out/${tag}_sweep_FETCH_SIZE.log
