#!/bin/bash
# usage (GPU box): tools/pmc_frame.sh <tag>  -> gpurun_out/<tag>_pmc.txt : per-kernel PMC averages of a few bench frames.
# Counter passes are separate runs (SQ: 8 slots, TCC: 4), never combined with sys/hip/hsa traces.
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() {  # name, counters...
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d gpurun_out/pmc_${tag}_$name -o p -- \
    python bench.py --no-cpu --steps 3 --warmup 2 > gpurun_out/pmc_${tag}_$name.log 2>&1
}
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE
run mem SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM TCP_PENDING_STALL_CYCLES_sum
python tools/pmc_summary.py gpurun_out/pmc_${tag}_sq gpurun_out/pmc_${tag}_tcc gpurun_out/pmc_${tag}_mem > gpurun_out/${tag}_pmc.txt 2>&1
