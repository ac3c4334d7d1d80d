cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r02}
SDM_GRAPH=0 tools/prof_bench.sh $tag
tools/pmc_sweep.sh $tag
timeout 900 python bench.py > gpurun_out/${tag}_bench_c3.json 2> gpurun_out/${tag}_bench_c3.err
tail -c 600 gpurun_out/${tag}_bench_c3.json
