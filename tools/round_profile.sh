#!/bin/bash
# usage (GPU box): tools/round_profile.sh <tag>  ->  every tracked profile of a round from ONE tree state, under gpurun_out/
# (copy them to profiles/ afterwards; profiles/<tag>_sweep_pmc* are written in place):
#   <tag>_bench_c3.json              the line `python bench.py` prints (default flags)
#   <tag>_kernel_stats.txt           rocprofv3 --kernel-trace --stats of the benchmark frames, launch by launch
#   <tag>_stress_kernel_stats.txt    the same on the busy scene
#   <tag>_sweep_pmc.json (+ CSVs)    HBM traffic of the sweep launches (FETCH_SIZE / WRITE_SIZE, separate passes)
#   <tag>_dense_pmc.txt              SQ counters of the sweep kernels on the dense case
#   <tag>_a7_sq.json                 SQ counters of the weight-update / visibility / birth / sweep kernels, C3 + busy scene
#   <tag>_driven.json, <tag>_driven_kernel_stats.txt   the `driven` leg alone and its kernel trace
#   <tag>_dense_split.txt            per-kernel times of the non-incremental sweep on the dense case + the streaming probes of this box
#   <tag>_clear.txt                  sdm_clear, ten calls
#   <tag>_driven_sweep.txt           the non-incremental sweep on a map the filter grew (150 frames of the driven scene), per launch and per kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r03}
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/${tag}_bench_c3.json 2> gpurun_out/${tag}_bench_c3.err
tail -c 400 gpurun_out/${tag}_bench_c3.json; echo
SDM_GRAPH=0 tools/prof_bench.sh $tag
SDM_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_stress -o ${tag}_stress -- python bench.py --only-stress > gpurun_out/${tag}_stress_prof.log 2>&1
python tools/trace_db.py gpurun_out/prof_${tag}_stress/${tag}_stress_results.db 3 > gpurun_out/${tag}_stress_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof_${tag}_stress
tools/pmc_sweep.sh $tag
tools/pmc_dense.sh $tag
tools/pmc_a7.sh $tag
tools/gpu_driven_stats.sh $tag > /dev/null 2>&1     # <tag>_driven.json, <tag>_driven_kernel_stats.txt
tools/gpu_dense_split.sh > /dev/null 2>&1; cp gpurun_out/dense_split.txt gpurun_out/${tag}_dense_split.txt
timeout 300 python tools/probes/clear_time.py 10 > gpurun_out/${tag}_clear.txt 2>&1
tools/gpu_driven_sweep.sh > /dev/null 2>&1; cp gpurun_out/driven_sweep.txt gpurun_out/${tag}_driven_sweep.txt
ls -la gpurun_out | grep $tag
