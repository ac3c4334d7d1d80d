cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
SDM_BENCH_SHARDED=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_sh -o sh -- python bench.py --steps 20 --warmup 5 --no-cpu --no-stress --no-driven --no-adapter --no-dense --no-strong > gpurun_out/sh_prof.log 2>&1
python tools/trace_db.py gpurun_out/prof_sh/sh_results.db 4 > gpurun_out/r06_sharded_one_rank_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof_sh
grep '"metric"' gpurun_out/sh_prof.log | tail -1 | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('sharded one rank: ms_per_step', j['ms_per_step'], j.get('collectives_us'))"
sed -n '/timeline/,$p' gpurun_out/r06_sharded_one_rank_kernel_stats.txt | head -60
