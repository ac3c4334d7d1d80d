tools/round_profile.sh r03 > gpurun_out/r03_round_profile.log 2>&1
SDM_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/libsdm_timers.so SDM_GRAPH=0 timeout 600 python tools/probes/crossframe.py 3 2>&1 | tail -3 > gpurun_out/r03_crossframe.txt
N=$(rocm-smi --showtoponuma 2>/dev/null | grep -i "Numa Node:" | head -1 | awk '{print $NF}')
LOCAL=$(cat /sys/devices/system/node/node$N/cpulist); OTHER=$(cat /sys/devices/system/node/node$((1-N))/cpulist)
run() { timeout 300 "$@" python bench.py --no-cpu --no-dense --no-strong --no-stress --no-grown --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(j['ms_per_step'], end=' ')"; }
{ echo "GPU on NUMA node $N (cpus $LOCAL); C3 frame, ms, one bench.py process per number"
  echo -n "bench.py as it is (pins itself to the GPU's node before HIP comes up): "; for i in 1 2 3 4; do run env; done; echo
  echo -n "SDM_NUMA_BIND=0, taskset to the GPU's node: "; for i in 1 2 3; do run env SDM_NUMA_BIND=0 taskset -c $LOCAL; done; echo
  echo -n "SDM_NUMA_BIND=0, taskset to the other node: "; for i in 1 2 3; do run env SDM_NUMA_BIND=0 taskset -c $OTHER; done; echo
  echo -n "SDM_NUMA_BIND=0, not pinned: "; for i in 1 2 3 4; do run env SDM_NUMA_BIND=0; done; echo
} > gpurun_out/r03_numa.txt 2>&1
