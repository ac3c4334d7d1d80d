#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/tr
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/tr/out.log 2> gpurun_out/tr/err.log
echo rc=$?
grep '"metric"' gpurun_out/tr/out.log | tail -1 | cut -c1-600
tail -3 gpurun_out/tr/err.log
