#!/bin/bash
# the round's closing run on the final tree: smoke(), bench line, kernel statistics (C3 + busy scene), in-kernel clocks,
# cross-frame gaps, the full GPU suite
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r04_bench_c3.json 2> gpurun_out/r04_bench_c3.err
tail -c 300 gpurun_out/r04_bench_c3.json; echo
SDM_GRAPH=0 tools/prof_bench.sh r04
SDM_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r04_stress -o r04_stress -- python bench.py --only-stress > gpurun_out/r04_stress_prof.log 2>&1
python tools/trace_db.py gpurun_out/prof_r04_stress/r04_stress_results.db 3 > gpurun_out/r04_stress_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof_r04_stress
SDM_LIB_PATH=build/ab/libsdm_timers.so timeout 300 python tools/probes/timers.py > gpurun_out/r04_in_kernel_timers.txt 2>&1
SDM_LIB_PATH=build/ab/libsdm_timers.so timeout 300 python tools/probes/crossframe.py 3 > gpurun_out/r04_crossframe.txt 2>&1
tail -4 gpurun_out/r04_crossframe.txt
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gpu_tests.log 2>&1
grep -E "passed|failed" gpurun_out/r04_gpu_tests.log | tail -2
