#!/bin/bash
# closing run of round 4 on the final tree (after the wait chains were taken apart): the full GPU suite, smoke(), the bench
# line, kernel statistics (C3 + busy scene), the sweep's HBM counters, SQ counters of the dense sweep, in-kernel clocks,
# cross-frame gaps.  Everything lands under gpurun_out/ with the prefix r04 (tools/round_profile.sh minus the A7 passes).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 700 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gpu_tests.log 2>&1
grep -E "passed|failed" gpurun_out/r04_gpu_tests.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/r04_bench_c3.json 2> gpurun_out/r04_bench_c3.err
tail -c 300 gpurun_out/r04_bench_c3.json; echo
SDM_GRAPH=0 tools/prof_bench.sh r04
tools/pmc_sweep.sh r04
tools/pmc_dense.sh r04
SDM_GRAPH=0 timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r04_stress -o r04_stress -- python bench.py --only-stress > gpurun_out/r04_stress_prof.log 2>&1
python tools/trace_db.py gpurun_out/prof_r04_stress/r04_stress_results.db 3 > gpurun_out/r04_stress_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof_r04_stress
SDM_LIB_PATH=build/ab/libsdm_timers.so timeout 200 python tools/probes/timers.py > gpurun_out/r04_in_kernel_timers.txt 2>&1
SDM_LIB_PATH=build/ab/libsdm_timers.so timeout 200 python tools/probes/crossframe.py 3 > gpurun_out/r04_crossframe.txt 2>&1
tail -4 gpurun_out/r04_crossframe.txt
