#!/bin/bash
# the way the driver launches N > 1 (python -m torch.distributed.run ...), with one rank; and the sharded engine with
# RCCL forced on at world size 1 (SDM_BENCH_SHARDED=1: communicator, collectives and their timers around one shard)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/tr
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-stress --no-grown --no-adapter --no-dense > gpurun_out/tr/out.log 2> gpurun_out/tr/err.log
echo rc=$?
grep '"metric"' gpurun_out/tr/out.log | tail -1 | cut -c1-400
tail -3 gpurun_out/tr/err.log
SDM_BENCH_SHARDED=1 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu --no-stress --no-grown --no-adapter --no-dense > gpurun_out/tr/sharded.log 2> gpurun_out/tr/sharded.err
echo rc=$?
grep '"metric"' gpurun_out/tr/sharded.log | tail -1 | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('sharded rehearsal: ms_per_step', j['ms_per_step'], 'collectives_us', j.get('collectives_us'), 'parallelism', j['config']['parallelism'])"
tail -3 gpurun_out/tr/sharded.err
