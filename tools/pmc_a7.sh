#!/bin/bash
# usage (GPU box): tools/pmc_a7.sh <tag> -> profiles-ready gpurun_out/<tag>_a7_sq.json
# SQ counter passes (separate runs, --kernel-trace only) over the weight-update and visibility kernels, on the
# benchmark frames (C3) and on the busy scene (bench.py --only-stress).
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() {  # workload name, pass name, bench args..., -- counters...
  wl=$1; pass=$2; shift 2
  args=()
  while [ "$1" != "--" ]; do args+=("$1"); shift; done
  shift
  SDM_GRAPH=0 timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d gpurun_out/pmca7_${tag}_${wl}_$pass -o p -- \
    python bench.py "${args[@]}" > gpurun_out/pmca7_${tag}_${wl}_$pass.log 2>&1
}
for wl in c3 stress; do
  if [ $wl = c3 ]; then a=(--no-cpu --no-dense --no-strong --steps 6 --warmup 3); else a=(--only-stress); fi
  run $wl valu "${a[@]}" -- SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY
  run $wl lds "${a[@]}" -- SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE
done
python tools/pmc_a7_json.py $tag
