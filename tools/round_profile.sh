cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tools/prof_bench.sh r01k
tools/pmc_sweep.sh r01k
timeout 600 python bench.py > gpurun_out/r01k_bench_c3.json 2> gpurun_out/r01k_bench_c3.err
tail -c 1500 gpurun_out/r01k_bench_c3.json
