#!/bin/bash
# the N > 1 code path of bench.py (gloo group, RCCL communicator, sdm_update_sharded, collective timers) with one rank
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/tr
SDM_BENCH_SHARDED=1 timeout 280 python bench.py --steps 20 --warmup 5 --no-cpu --no-stress --no-grown --no-adapter --no-dense > gpurun_out/tr/sharded.log 2> gpurun_out/tr/sharded.err
echo rc=$?
grep '"metric"' gpurun_out/tr/sharded.log | tail -1 | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('sharded rehearsal: ms_per_step', j['ms_per_step'], 'collectives_us', j.get('collectives_us'), j['config']['parallelism'], 'strong', j.get('strong_scaling', {}).get('ms_per_step'))"
tail -2 gpurun_out/tr/sharded.err
