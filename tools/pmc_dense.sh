#!/bin/bash
# usage (GPU box): tools/pmc_dense.sh <tag> -> gpurun_out/<tag>_dense_pmc.txt : SQ counters of the sweep kernels on the dense case
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() {
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d gpurun_out/pmcd_${tag}_$name -o p -- \
    python tools/probes/dense_only.py 3 > gpurun_out/pmcd_${tag}_$name.log 2>&1
}
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM
run mem SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
python tools/pmc_summary.py gpurun_out/pmcd_${tag}_sq gpurun_out/pmcd_${tag}_mem > gpurun_out/${tag}_dense_pmc.txt 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${tag}_dense_pmc.txt")))
for r in rows:
    if "occupancy" in r["kernel"]:
        w=float(r["SQ_WAVES"]); ch=w*8
        print(r["kernel"], "per chunk: VALU %.0f SALU %.0f LDS %.0f VMEM %.0f | active %.0f wait %.0f stall %.0f quad-cycles" % (
            float(r["SQ_INSTS_VALU"])/ch, float(r["SQ_INSTS_SALU"])/ch, float(r["SQ_INSTS_LDS"])/ch, float(r["SQ_INSTS_VMEM"])/ch,
            float(r["SQ_ACTIVE_INST_ANY"])/ch, float(r["SQ_WAIT_ANY"])/ch, float(r["SQ_WAIT_INST_ANY"])/ch))
PY
grep dense_ms gpurun_out/pmcd_${tag}_sq.log
