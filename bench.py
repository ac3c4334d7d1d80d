#!/usr/bin/env python
"""bench.py — whole-frame throughput of the particle-grid update hot path on MI355X.

Metric (BASELINE.json): Mvoxels updated / s = voxels of the map / frame time of the hot path
(one "step" = one subObjectLevelUpdate-equivalent frame: ego shift, object moves, visibility/binning,
SMC-PHD weight update, births + resampling, occupancy/semantic sweep), inputs resident in HBM.

  N = 1 : BASELINE config C3 — 256^3 voxels, 8 slots/voxel, ~2M live particles, 1242x375 VKITTI2 camera,
          dynamic objects (4x4 transforms), cfg/options_virtual_kitti2.yaml parameters.
  N > 1 : weak scaling, 2^24 voxels and ~2M particles per GPU (…C5 = 512^3 / 16M particles at N = 8),
          Z-slab shards, one process per GPU, exchanges over RCCL inside the library (member counts, slab-crossing
          copies, chunk-owner reduction of the partial ck images); per-collective GPU times in `collectives_us`.

Prints ONE JSON line on rank 0 (contract in the task description), with two extra objects:
  roofline     — occupancy/semantic sweep kernel: algorithmic bytes (80 B/voxel at 8 slots)
                 / HIP-event time on the stream it runs on, against the 8 TB/s HBM peak; beside it what this
                 layout really moves per launch (`layout`), the PMC traffic, the non-incremental launch
                 (`full_evaluation`) and the dense case.
  cpu_baseline — the CPU oracle (literal single-thread restatement of the reference, kind "port") timed on
                 the same workload on this box's host cores (bounded sample), rank 0 at N = 1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
# RCCL shares buffers between the ranks' processes through dmabuf IPC; the host driver here supports nothing else
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_BPS = 8.0e12  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
PREFILL_FACTOR = 1.16   # prefilled / live particles after the first sweeps' culls (see main)


def timed_run(synth, sharded, dist, rank, world, local_rank, cfg, params, particles_per_shard, steps, warmup, scene_kw,
              prefill_kw=None, force_comm=None, exchange=None):
    """One more map of its own: frames rendered and uploaded, map prefilled, `warmup` + `steps` frames issued back to back,
    barrier + synchronize on both sides of the timed ones, max over ranks.  Returns the numbers and the engine (open).
    force_comm: the frames go through sdm_update_sharded on an RCCL communicator whatever the world size."""
    eng = sharded.NativeShardedMap(cfg, params, rank, world, local_rank, dist=dist,
                                   force_comm=(dist is not None) if force_comm is None else force_comm,
                                   exchange=exchange or os.environ.get("SDM_EXCHANGE", "rccl"))
    m = eng.map
    m.generate_noise_table(seed=20250217)
    scene = synth.Scene(cfg, **scene_kw)
    frames = []
    for t in range(warmup + steps):
        depth, cloud, pos, q = scene.render(t, params)
        frames.append((depth, cloud, pos, q, scene.moves(t), m.device_put(depth), m.device_put(cloud)))
    st, ring, _ = synth.prefill_state(cfg, scene, int(particles_per_shard * PREFILL_FACTOR), shard_rank=rank, shard_count=world, **(prefill_kw or {}))
    m.load_state(st)
    m.set_ring_state(ring)

    def fence():
        m.synchronize()
        m.device_synchronize()
        if dist is not None:
            dist.barrier()

    def run(lo, hi):
        for t in range(lo, hi):
            eng.update(frames[t][5], frames[t][6], frames[t][2], frames[t][3], frames[t][4])

    run(0, warmup)
    fence()
    t0 = time.perf_counter()
    run(warmup, warmup + steps)
    t_enq = time.perf_counter() - t0
    fence()
    dt = time.perf_counter() - t0
    stats = m.stats(count_live=True)
    live, n_vis = stats["live_particles"], stats["n_visible"]
    if dist is not None:
        import torch
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        lt = torch.tensor([live, n_vis], dtype=torch.int64)
        dist.all_reduce(lt)
        live, n_vis = int(lt[0].item()), int(lt[1].item())
    V = 1 << (cfg["x_n"] + cfg["y_n"] + cfg["z_n"])
    res = {"value": round(V / (dt / steps) / 1e6, 1), "unit": "Mvoxels/s", "ms_per_step": round(dt * 1e3 / steps, 4), "steps": steps,
           "warmup": warmup, "n_gpus": world, "voxels": V, "live_particles": live, "visible_particles": n_vis,
           "host_enqueue_ms_per_step": round(t_enq * 1e3 / steps, 4)}
    return res, eng, scene, frames, (st, ring)


def stress_run(synth, sharded, steps=8, warmup=14, cpu_frames=4):
    """A busier frame: cluttered street (200 static boxes, 12 moving ones), three noisy births per point (thick
    surfaces), the camera turning 1.5 deg/frame and drifting sideways so that x slabs of the ring are recycled too, the
    map topped up to 2.0 M particles.  Timed like the headline run; then per-stage GPU times of four more
    frames, and the oracle on the very same map state (dumped from the GPU after the warm-up) and frames."""
    cfg = synth.CONFIGS["C3"]
    params = synth.PARAMS["vkitti2_nb3"]
    scene_kw = dict(n_static=200, n_dynamic=12, seed=11, yaw_rate_deg=1.5, lateral_extra=(0, 0.04))
    res, eng, scene, frames, (st, ring) = timed_run(synth, sharded, None, 0, 1, 0, cfg, params, 2000000, steps, warmup, scene_kw)
    m = eng.map
    # the oracle starts where the timed region started: replay is deterministic, so re-run the warm-up on a second map
    # would do too - cheaper: a fresh map, warm-up, dump
    eng_b = sharded.NativeShardedMap(cfg, params, 0, 1, 0)
    mb = eng_b.map
    noise = m.download_noise_table()
    mb.upload_noise_table(noise)
    mb.load_state(st)
    mb.set_ring_state(ring)
    del st
    for t in range(warmup):
        eng_b.update(frames[t][5], frames[t][6], frames[t][2], frames[t][3], frames[t][4])
    mb.synchronize()
    state0, ring0, stamps0 = mb.dump_state(), mb.ring_state(), mb.stamps()
    # per-stage GPU times of the timed frames, on the copy (same state, same frames)
    mb.set_profiling(True)
    acc, live_l, tiles_l, slabs, vis_l = np.zeros(8), [], [], np.zeros(3), []
    n_prof = min(4, steps)
    for t in range(warmup, warmup + n_prof):
        eng_b.update(frames[t][5], frames[t][6], frames[t][2], frames[t][3], frames[t][4])
        mb.synchronize()
        stt = mb.stats()
        acc += np.array(stt["stage_ms"])
        live_l.append(stt["sweep_live_voxels"])
        tiles_l.append(stt["sweep_tiles"])
        vis_l.append(stt["n_visible"])
        slabs += np.array(stt["restamped_slabs"])
    mb.close()
    names = ["", "ego", "move", "remove", "visibility", "weight", "birth", "occupancy"]
    res["stage_ms"] = {k: round(acc[i] / n_prof, 4) for i, k in enumerate(names) if k}
    res["sweep"] = {"tiles_looked_into": int(np.mean(tiles_l)), "voxels_evaluated_in_full": int(np.mean(live_l))}
    res["visible_particles_per_frame"] = int(np.mean(vis_l))
    res["restamped_slabs_per_frame"] = [round(float(x) / n_prof, 2) for x in slabs]
    res["workload"] = ("stress: C3 grid, 200 static + 12 moving boxes, 3 noisy births per point, yaw 1.5 deg/frame + sideways drift, "
                       "%d live particles, %d visible per frame" % (res["live_particles"], res["visible_particles_per_frame"]))
    if cpu_frames > 0:
        from oracle import oracle as orc
        o = orc.OracleMap(dict(cfg, bin_order=0), params, noise)
        o.load_state(state0)
        o.set_stamps(*stamps0)
        o.set_ring_state(ring0)
        del state0
        times = []
        for t in range(warmup, warmup + min(cpu_frames + 1, steps)):
            t0 = time.perf_counter()
            o.update(frames[t][0], frames[t][1], frames[t][2], frames[t][3], frames[t][4])
            if t > warmup:  # the first frame warms the caches
                times.append(time.perf_counter() - t0)
        med = float(np.median(times))
        res["cpu_baseline"] = {"value": round(res["voxels"] / med / 1e6, 2), "unit": "Mvoxels/s", "cores": 1, "kind": "port",
                               "ms_per_frame": round(med * 1e3, 1),
                               "sample": "%d frames from the same map state (dumped from the GPU after the warm-up), 1 thread" % len(times)}
        res["x_cpu"] = round(res["value"] / res["cpu_baseline"]["value"], 1)
    m.close()
    return res


def driven_run(synth, sharded, steps=20, cpu_frames=3):
    """The workload nothing is prefilled for: start with an EMPTY C3 map, drive synth.DRIVEN_FRAMES frames down the
    cluttered street of synth.DRIVEN_SCENE (0.3 m per frame: 66 m, more than the map is long; ring shifts on two axes; 12
    moving boxes; three noisy births per point), then time `steps` more frames issued back to back like the headline
    run; per-stage GPU times of four more frames, and the oracle (literal order, one thread) on the state the timed
    region started from and its frames.  Every live particle is one the filter put there, every count is what comes out.
    tests/test_driven_gpu.py runs the oracle BESIDE the GPU over the same drive from frame 0."""
    cfg = synth.CONFIGS["C3"]
    params = synth.PARAMS[synth.DRIVEN_PARAMS]
    scene_kw = dict(synth.DRIVEN_SCENE)
    n_grow = synth.DRIVEN_FRAMES
    scene = synth.Scene(cfg, **scene_kw)
    n_prof = 4
    t0 = time.time()
    # (SDM_DRIVEN_CACHE=<file>.npz, development: A/B runs of the library on the same rendered frames; the file is keyed by
    # everything the frames depend on and holds plain arrays)
    rendered = synth.render_frames_cached(cfg, params, scene_kw, range(n_grow + steps + n_prof), os.environ.get("SDM_DRIVEN_CACHE"))  # worker processes: seconds per frame on one core
    t_render = time.time() - t0
    eng = sharded.NativeShardedMap(cfg, params, 0, 1, 0)
    m = eng.map
    m.generate_noise_table(seed=20250217)
    noise = m.download_noise_table()
    t0 = time.time()
    pending = []
    for t in range(n_grow):
        depth, cloud, pos, q = rendered[t]
        dd, dc = m.device_put(depth), m.device_put(cloud)
        eng.update(dd, dc, pos, q, scene.moves(t))
        pending += [dd, dc]
        if len(pending) >= 16:
            m.synchronize()
            for ptr in pending:
                m.device_free(ptr)
            pending = []
    m.synchronize()
    for ptr in pending:
        m.device_free(ptr)
    state0, ring0, stamps0 = m.dump_state(), m.ring_state(), m.stamps()
    frames = []
    for t in range(n_grow, n_grow + steps + n_prof):
        depth, cloud, pos, q = rendered[t]
        frames.append((depth, cloud, pos, q, scene.moves(t), m.device_put(depth), m.device_put(cloud)))
    del rendered
    t_grow = time.time() - t0

    def fence():
        m.synchronize()
        m.device_synchronize()

    fence()
    t0 = time.perf_counter()
    for f in frames[:steps]:
        eng.update(f[5], f[6], f[2], f[3], f[4])
    fence()
    dt = time.perf_counter() - t0
    stats = m.stats(count_live=True)
    m.set_profiling(True)
    acc, vis_l, live_l, tiles_l = np.zeros(8), [], [], []
    for f in frames[steps:]:
        eng.update(f[5], f[6], f[2], f[3], f[4])
        m.synchronize()
        stt = m.stats()
        acc += np.array(stt["stage_ms"])
        vis_l.append(stt["n_visible"])
        live_l.append(stt["sweep_live_voxels"])
        tiles_l.append(stt["sweep_tiles"])
    m.set_profiling(False)
    V = 1 << (cfg["x_n"] + cfg["y_n"] + cfg["z_n"])
    names = ["", "ego", "move", "remove", "visibility", "weight", "birth", "occupancy"]
    res = {"value": round(V / (dt / steps) / 1e6, 1), "unit": "Mvoxels/s", "ms_per_step": round(dt * 1e3 / steps, 4), "steps": steps,
           "frames_grown_from_empty": n_grow, "live_particles": stats["live_particles"], "live_voxels": stats["live_voxels"],
           "visible_particles_per_frame": int(np.mean(vis_l)),
           "stage_ms": {k: round(acc[i] / n_prof, 4) for i, k in enumerate(names) if k},
           "sweep": {"tiles_looked_into": int(np.mean(tiles_l)), "voxels_evaluated_in_full": int(np.mean(live_l))},
           "render_s": round(t_render, 1), "grow_s": round(t_grow, 1),
           "workload": "driven: C3 grid, EMPTY map, %d frames (%.0f m) down a street of %d static + %d moving boxes, 3 noisy births per point, "
                       "forward %.1f m + yaw %.1f deg per frame, ring shifts on z and x; then %d timed frames.  No prefill: %d live particles in %d "
                       "voxels and %d visible per frame are what the filter and the camera yield"
                       % (n_grow, n_grow * scene.speed, scene_kw["n_static"], scene_kw["n_dynamic"], scene.speed, scene_kw["yaw_rate_deg"], steps,
                          stats["live_particles"], stats["live_voxels"], int(np.mean(vis_l)))}
    # the non-incremental sweep on this map (what the first sweep after sdm_load_state / sdm_set_params costs on a map the
    # filter built: its live voxels are surfaces, not the evenly strewn filler of the headline map), bytes as in
    # roofline.full_evaluation: stamp, flag read, result and flag written per voxel + the records of the live voxels
    m.time_occupancy_sweep(iters=200)
    full_ms = m.time_occupancy_sweep(iters=10)
    full_bytes = V * (2 + 1 + 8 + 1) + stats["live_voxels"] * 10 * ((1 << cfg["p_n"]) - 1)
    res["full_evaluation"] = {"avg_launch_ms": round(full_ms, 5), "bytes_per_launch": int(full_bytes),
                              "frac": round(full_bytes / full_ms / 1e6 / (HBM_PEAK_BPS / 1e9), 4), "launches_timed": 10,
                              "untimed_launches_before": 201}
    if cpu_frames > 0:
        from oracle import oracle as orc
        o = orc.OracleMap(dict(cfg, bin_order=0), params, noise)
        o.load_state(state0)
        o.set_stamps(*stamps0)
        o.set_ring_state(ring0)
        del state0
        times = []
        for k, f in enumerate(frames[:cpu_frames + 1]):
            t0 = time.perf_counter()
            o.update(f[0], f[1], f[2], f[3], f[4])
            if k > 0:  # the first frame warms the caches
                times.append(time.perf_counter() - t0)
        med = float(np.median(times))
        res["cpu_baseline"] = {"value": round(V / med / 1e6, 2), "unit": "Mvoxels/s", "cores": 1, "kind": "port", "ms_per_frame": round(med * 1e3, 1),
                               "sample": "%d frames from the same map state (dumped from the GPU after the %d growth frames), 1 thread" % (len(times), n_grow)}
        res["x_cpu"] = round(res["value"] / res["cpu_baseline"]["value"], 1)
    m.close()
    return res


import ctypes

C_INT32 = ctypes.c_int32


def pin_to_device_node(device):
    """The process moves onto the NUMA node its GPU hangs off BEFORE the HIP runtime comes up in it: a frame is a chain of
    ~50 dependent launches whose packets and completion signals live in host memory the runtime allocates where the
    process runs; from the far socket every gap between two dependent kernels is 2-4 us longer (C3: 0.266 against 0.292 ms
    per frame).  The library finds the node without the runtime (KFD topology in sysfs) when it is loaded;
    where sysfs cannot tell, a child process asks HIP and this one sets its affinity from the answer.
    SDM_NUMA_BIND=0: leave the affinity alone.  Returns a description for the bench line."""
    if os.environ.get("SDM_NUMA_BIND") == "0":
        return "not pinned (SDM_NUMA_BIND=0)"
    # 1. without the runtime: the library finds the device's node in sysfs (KFD topology) and moves this thread when it is
    #    loaded - no HIP call has happened in this process yet
    try:
        from semantic_dsp_map_amd import binding
        L = binding.load_library()
        L.sdm_host_numa_node_early.restype = C_INT32
        L.sdm_host_numa_node_early.argtypes = [C_INT32]
        if L.sdm_host_numa_node_early(device) >= 0:
            node = binding.HOST_NUMA_NODE
            if node is not None and node >= 0:
                return ("pinned to NUMA node %d (the GPU's, from the KFD topology), %d CPUs, before HIP initialised"
                        % (node, len(os.sched_getaffinity(0))))
            return "not pinned (one node, or already there)"
    except Exception:  # noqa: BLE001 - fall through to the second way
        pass
    # 2. sysfs cannot tell: a child process asks HIP, this one sets its affinity from the answer (still before HIP here)
    import subprocess
    code = ("import ctypes, sys; sys.path.insert(0, %r); from semantic_dsp_map_amd import binding; L = binding.load_library(); "
            "L.sdm_bind_host_thread.restype = ctypes.c_int32; L.sdm_bind_host_thread.argtypes = [ctypes.c_int32]; "
            "print('NODE', L.sdm_bind_host_thread(%d))" % (os.path.dirname(os.path.abspath(__file__)), device))
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300).stdout
        node = int([l for l in out.splitlines() if l.startswith("NODE")][-1].split()[1])
        if node < 0:
            return "not pinned (one node, or nothing to choose)"
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return "not pinned (no allowed CPU on node %d)" % node
        os.sched_setaffinity(0, cpus)
        return "pinned to NUMA node %d (the GPU's, asked of HIP in a child process), %d CPUs" % (node, len(cpus))
    except Exception as e:  # noqa: BLE001 - the pin is an optimisation
        return "not pinned (%s)" % type(e).__name__


def spawn_ranks(n):
    """`python bench.py --gpus N` started WITHOUT a launcher (no WORLD_SIZE in the environment): start the N ranks here -
    one child process per GPU running this very command line with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT set, what `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` would do - pass rank 0's
    stdout (the ONE JSON line) through, keep the other ranks' stdout out of it, return the worst exit code.  A rank that
    dies takes the others down with it instead of leaving them in a rendezvous."""
    import socket
    import subprocess
    env0 = dict(os.environ)
    env0.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in env0:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            env0["MASTER_PORT"] = str(sk.getsockname()[1])
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0:
                rc = rc or code
                for q in pending:  # (exact children of this process, by handle)
                    q.terminate()
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--particles", type=int, default=2000000, help="live particles per GPU after prefill")
    ap.add_argument("--cpu-frames", type=int, default=10, help="frames of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-dense", action="store_true", help="skip the dense-case sweep timing after the run")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling run (C4: 256^3 / 8M particles over the GPUs)")
    ap.add_argument("--no-sharded-leg", action="store_true", help="skip the one-rank run of the sharded frame (N = 1 only)")
    ap.add_argument("--sharded-leg", action="store_true", help="run it also with --no-cpu (which skips the side legs)")
    ap.add_argument("--no-stress", action="store_true", help="skip the busy-scene run (N = 1 only)")
    ap.add_argument("--only-stress", action="store_true", help="run nothing but the busy scene (development)")
    ap.add_argument("--no-driven", action="store_true", help="skip the drive from an empty map (N = 1 only)")
    ap.add_argument("--only-driven", action="store_true", help="run nothing but the drive from an empty map (development)")
    ap.add_argument("--no-adapter", action="store_true", help="skip the end-to-end leg through the C++ class (host buffers in, clouds out)")
    args = ap.parse_args()

    host_numa = pin_to_device_node(int(os.environ.get("LOCAL_RANK", "0")))
    from semantic_dsp_map_amd import sharded, synth

    if args.only_stress:
        print(json.dumps({"stress": stress_run(synth, sharded)}))
        return
    if args.only_driven:
        print(json.dumps({"driven": driven_run(synth, sharded)}))
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    fake = os.environ.get("SDM_BENCH_FAKE_RANK")
    if fake and "WORLD_SIZE" in os.environ:
        # tests/test_bench_contract.py: a rank that only reports the environment the launcher gave it (no GPU needed)
        if fake == "fail1" and os.environ["RANK"] == "1":
            raise SystemExit(3)
        if fake == "fail1":
            time.sleep(30)  # (rank 0 would sit in the rendezvous: the launcher has to take it down)
        print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around this process: be the launcher (one process per GPU, rank 0's line on stdout)
        raise SystemExit(spawn_ranks(args.gpus))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d under a launcher that set WORLD_SIZE=%d: the two must agree" % (args.gpus, world))
    # SDM_BENCH_SHARDED=1: take the N > 1 code path - gloo process group, RCCL communicator, sdm_update_sharded, collective
    # timers - with whatever world size there is (1 on the boxes with one GPU): a rehearsal of the multi-GPU bench line
    multi = world > 1 or os.environ.get("SDM_BENCH_SHARDED") == "1"
    dist = None
    if multi:
        # torch.distributed is plumbing only: gloo (CPU) for the rendezvous, barriers and the max over ranks.
        # The data-path collectives are RCCL calls inside libsdm_hip on the system HIP runtime.
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29531")  # (the one-process rehearsal is started without a launcher)
        with sharded.stdout_to_stderr():  # gloo prints its connection summary on stdout
            dist.init_process_group("gloo", rank=rank, world_size=world)

    base = synth.CONFIGS[args.config]
    cfg = sharded.weak_scaled_config(base, world)
    params = synth.PARAMS[synth.CONFIG_PARAMS[args.config]]
    V = 1 << (cfg["x_n"] + cfg["y_n"] + cfg["z_n"])
    S = 1 << cfg["p_n"]
    n_frames = args.warmup + args.steps

    # (SDM_EXCHANGE=ipc: the exchanges of a sharded frame through peer-mapped arenas instead of RCCL)
    eng = sharded.NativeShardedMap(cfg, params, rank, world, local_rank, dist=dist, force_comm=multi, exchange=os.environ.get("SDM_EXCHANGE", "rccl"))
    m = eng.map
    # noise table: rocRAND on the device, read back so that the CPU baseline uses the same floats
    m.generate_noise_table(seed=20250217)
    noise = m.download_noise_table()

    # ---- synthetic frames (same on every rank), uploaded to HBM before the timed region
    scene = synth.Scene(cfg, n_static=48, n_dynamic=6, seed=7)
    t0 = time.time()
    frames = []
    for t in range(n_frames):
        depth, cloud, pos, q = scene.render(t, params)
        frames.append((depth, cloud, pos, q, scene.moves(t), m.device_put(depth), m.device_put(cloud)))
    t_render = time.time() - t0

    # The first sweeps cull about one prefilled particle in eight (weights below the initial weight, operations.h:410-424):
    # the map is prefilled with that many more, so that what is LIVE during the timed frames is the 2 M BASELINE.json
    # quotes the metric on (`live_particles` in the line is counted after the timed region).
    n_prefill = int(args.particles * PREFILL_FACTOR)
    st, ring, n_pre = synth.prefill_state(cfg, scene, n_prefill, shard_rank=rank, shard_count=world)
    m.load_state(st)
    m.set_ring_state(ring)

    def run(lo, hi):
        for t in range(lo, hi):
            depth, cloud, pos, q, moves, d_depth, d_cloud = frames[t]
            eng.update(d_depth, d_cloud, pos, q, moves)

    def fence():
        # hipStreamSynchronize + hipDeviceSynchronize on the HIP runtime libsdm_hip runs on.  (torch bundles a second
        # HIP runtime; torch.cuda.synchronize() would not see this library's streams, so it is not used.)
        m.synchronize()
        m.device_synchronize()
        if dist is not None:
            dist.barrier()

    run(0, args.warmup)
    fence()
    t0 = time.perf_counter()
    run(args.warmup, n_frames)
    t_enqueue = time.perf_counter() - t0  # host time to issue the frames (the GPU runs behind it)
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    stats = m.stats(count_live=True)
    issue_mode = ("sharded: launch by launch, collectives on the streams" if multi else
                  "hipGraph replay" if stats["graph_frames"] >= args.steps else
                  "launch by launch" if stats["graph_frames"] == 0 else "mixed")
    launch_mode = {"graph_frames": stats["graph_frames"], "direct_frames": stats["direct_frames"],
                   "host_us_per_launched_frame": round(stats["host_enqueue_us"], 1),
                   "policy": "SDM_GRAPH=%s (0 launch by launch, 1 one branched hipGraph, 3 one chain graph, 4 five chain graphs on the "
                             "frame's streams, 2 = default: by the host's speed at issuing 50 empty kernel launches, measured "
                             "when the map is created - launch by launch up to 110 us, the five graphs up to 170 us, the chain beyond)" % os.environ.get("SDM_GRAPH", "2")}
    live, n_vis, live_vox_local = stats["live_particles"], stats["n_visible"], stats["live_voxels"]
    if dist is not None:
        lt = torch.tensor([live, n_vis], dtype=torch.int64)
        dist.all_reduce(lt)
        live, n_vis = int(lt[0].item()), int(lt[1].item())

    # ---- per-stage GPU times and the roofline of the dominant streaming kernel (occupancy / semantic sweep), taken on
    # a few more frames after the timed region: HIP events on the stream the kernels run on bracket every stage
    # (sdm_set_profiling), so "occupancy" is the in-frame duration of the sweep launch; the voxels it evaluated in full
    # and the tiles it looked into come from the library's counters.
    n_extra = 6
    m.set_profiling(True)
    if multi:
        m.comm_timing(True)

    comm_us = {}

    def profiled(t_lo, t_hi):
        acc, live_l, tiles_l, slabs = np.zeros(8), [], [], np.zeros(3)
        for t in range(t_lo, t_hi):
            depth, cloud, pos, q = scene.render(t, params)
            dd, dc = m.device_put(depth), m.device_put(cloud)
            eng.update(dd, dc, pos, q, scene.moves(t))
            m.synchronize()
            if multi:
                for k, v in m.comm_times().items():
                    comm_us.setdefault(k, []).append(v)
            stt = m.stats()
            acc += np.array(stt["stage_ms"])
            live_l.append(stt["sweep_live_voxels"])
            tiles_l.append(stt["sweep_tiles"])
            slabs += np.array(stt["restamped_slabs"])
        n = t_hi - t_lo
        return acc / n, float(np.mean(live_l)), float(np.mean(tiles_l)), slabs / n

    stage_ms, sweep_live_avg, sweep_tiles_avg, slabs_avg = profiled(n_frames, n_frames + n_extra)
    # the same with the camera also moving one voxel per frame sideways: an x slab of the ring is recycled every frame (a
    # turning vehicle) - one voxel of every x row of the map changes its result
    scene.lateral_extra = (n_frames + n_extra - 1, cfg["voxel_size"])
    xs_stage, xs_live, xs_tiles, xs_slabs = profiled(n_frames + n_extra, n_frames + 2 * n_extra)
    m.set_profiling(False)
    collectives = None
    if multi:
        # GPU time of each collective of a sharded frame (HIP events around it on the stream it is issued on; includes
        # waiting for the slowest shard to arrive), averaged over the profiled frames, max over the ranks; and what a shard
        # receives per frame
        m.comm_timing(False)
        import torch
        names = sorted(comm_us)
        ct = torch.tensor([float(np.mean(comm_us[k])) for k in names], dtype=torch.float64)
        dist.all_reduce(ct, op=dist.ReduceOp.MAX)
        chunk = m.ck_chunk_elems()
        seg = sharded.halo_segment_bytes(sharded.HALO_DEFAULT_CAP)
        collectives = {"us": {k: round(float(v), 1) for k, v in zip(names, ct.tolist())},
                       "frames_timed": 2 * n_extra,
                       "ck_exchange": ("one all-gather of the whole partial images (SDM_CK_EXCHANGE=allgather: its time is under ck_alltoall, "
                                       "ck_allgather did not run)" if os.environ.get("SDM_CK_EXCHANGE") == "allgather" else
                                       "chunk-owner reduction: all-to-all + slab-ordered sum + all-gather (default; SDM_CK_EXCHANGE=allgather "
                                       "selects the one-collective variant)"),
                       "bytes_received_per_shard_per_frame": {"counts_allgather": (world - 1) * sharded.HALO_OBJ * 4,
                                                              "halo_alltoall": (world - 1) * seg,
                                                              "ck_alltoall": (world - 1) * chunk * 4,
                                                              "ck_allgather": (world - 1) * chunk * 4},
                       "note": "counts_allgather rides the member-count stream beside the previous frame's sweep; the other "
                               "three are on the frame's critical path"}
        collectives["us_on_critical_path"] = round(sum(v for k, v in collectives["us"].items() if k != "counts_allgather"), 1)
    sweep_ms = float(stage_ms[7])
    ms_per_step = dt * 1e3 / args.steps
    value = V / (dt / args.steps) / 1e6  # Mvoxels / s, whole map (all shards)
    # Roofline of the occupancy / semantic sweep.  `frac` = bytes the launch has to move in this layout / launch time /
    # peak: one byte per 2048-voxel tile; for the tiles something was written or stamped in since the previous sweep,
    # the 2-byte observation stamp and 1-byte flag of every voxel; the record (10 (S - 1) bytes: round 6 took the time
    # particle's row out of it, its stamp lives in the dense stamp array), the 8-byte result and the
    # flag byte of the voxels that were written to.  (Result entries that flip to "unobserved" / "empty" are also
    # written, 9 B each, but not counted.)  `traffic` is the PMC measurement of the same launches where a committed
    # profile matches.  SURVEY.md 8(d) counts the dense-slot figure - (S-1) x 10 B + 2 B read + 8 B written per voxel,
    # 80 B at S = 8 - for every voxel of the map whatever the layout skips: that view is kept under `effective`, it is
    # a rate of voxels answered for, not of bytes moved.  `dense_case` is the launch on a map where the two views
    # coincide (every slot of every voxel live), `full_evaluation` the non-incremental launch on the benchmark map.
    TILE = 2048
    vox = V // world
    dense_slot_bytes = vox * ((S - 1) * 10 + 2 + 8)

    def in_frame_bytes(tiles, evaluated):
        return vox // TILE + tiles * TILE * (2 + 1) + evaluated * (10 * (S - 1) + 8 + 1)

    layout_bytes = in_frame_bytes(sweep_tiles_avg, sweep_live_avg)
    achieved = layout_bytes / (sweep_ms * 1e-3)
    traffic, traffic_source = pmc_traffic(S, vox, sweep_live_avg, sweep_tiles_avg)
    roofline = {"kernel": "k_occupancy<%d>" % S, "bound": "hbm",
                "case": "in-frame launch (incremental: tiles / voxels written or stamped since the previous sweep)",
                "achieved": round(achieved / 1e9, 1), "peak": HBM_PEAK_BPS / 1e9, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_BPS, 4),
                "traffic": traffic,
                "traffic_source": None if traffic is None else "%s: rocprofv3 PMC passes of this command committed with the tree "
                                                               "(FETCH_SIZE x 2 + WRITE_SIZE), not measured in this run" % traffic_source,
                "bytes_per_launch": int(layout_bytes),
                "avg_launch_ms": round(sweep_ms, 5), "voxels": vox, "launches_timed": n_extra,
                "tiles_looked_into": int(sweep_tiles_avg), "tiles": vox // TILE,
                "voxels_evaluated_in_full": int(sweep_live_avg), "voxels_with_live_slots": live_vox_local,
                "restamped_slabs_per_frame": [round(float(x), 2) for x in slabs_avg],
                "effective": {"bytes_per_launch": int(dense_slot_bytes),
                              "achieved": round(dense_slot_bytes / (sweep_ms * 1e-3) / 1e9, 1),
                              "x_peak": round(dense_slot_bytes / (sweep_ms * 1e-3) / HBM_PEAK_BPS, 3),
                              "definition": "SURVEY.md 8(d): %d B/voxel dense-slot figure x %d voxels / launch time - voxels "
                                            "answered for, not bytes moved" % ((S - 1) * 10 + 10, vox)},
                "x_shift_frames": {"restamped_slabs_per_frame": [round(float(x), 2) for x in xs_slabs],
                                   "sweep_ms": round(float(xs_stage[7]), 5), "frame_begin_ms": round(float(xs_stage[1]), 5),
                                   "gpu_frame_ms": round(float(np.sum(xs_stage[1:])), 4),
                                   "gpu_frame_ms_other_frames": round(float(np.sum(stage_ms[1:])), 4),
                                   "tiles_looked_into": int(xs_tiles), "voxels_evaluated_in_full": int(xs_live),
                                   "frac": round(in_frame_bytes(xs_tiles, xs_live) / (float(xs_stage[7]) * 1e-3) / HBM_PEAK_BPS, 4),
                                   "launches_timed": n_extra}}

    cpu = None
    if rank == 0 and not multi and not args.no_cpu and args.cpu_frames > 0:
        cpu = cpu_baseline(cfg, params, noise, frames, st, ring, args.cpu_frames, V)

    if not multi and not args.no_dense:
        # the same kernel on SURVEY.md 8(d)'s dense case (every slot of every voxel live, all of them to be evaluated):
        # run last, it overwrites the map
        # first the non-incremental launch on the benchmark state: every tile, every voxel's result written, every voxel
        # that holds a live slot evaluated (what each sweep did before the clean/dirty state, and what the first sweep
        # after sdm_load_state / sdm_set_params does)
        # (the CPU leg above left the GPU idle for seconds: the clocks come back up over the first milliseconds of work, and
        # ten launches are over in less than one - untimed launches first, for each case)
        m.time_occupancy_sweep(iters=200)
        full_ms = m.time_occupancy_sweep(iters=10)
        full_bytes = V * (2 + 1 + 8 + 1) + live_vox_local * 10 * (S - 1)
        roofline["full_evaluation"] = {"kernel": "k_occupancy_scan<%d> + k_occupancy_dense<%d> (two launches, timed together)" % (S, S), "bytes_per_launch": full_bytes,
                                       "avg_launch_ms": round(full_ms, 5),
                                       "achieved": round(full_bytes / full_ms / 1e6, 1),
                                       "frac": round(full_bytes / full_ms / 1e6 / (HBM_PEAK_BPS / 1e9), 4),
                                       "launches_timed": 10, "untimed_launches_before": 201}
        m.fill_dense()
        m.time_occupancy_sweep(iters=40)
        dense_ms = m.time_occupancy_sweep(iters=10)
        # stamp, flag read; result written; record read (10 (S - 1) B: the SURVEY's own per-voxel figure); the flag byte is
        # only written where it changes (on this map: never after the first sweep) and is not counted: 81 B/voxel at S = 8
        dense_bytes = V * (2 + 1 + 8 + 10 * (S - 1))
        roofline["dense_case"] = {"kernel": "k_occupancy_scan<%d> + k_occupancy_dense<%d> (two launches, timed together)" % (S, S), "bytes_per_launch": dense_bytes,
                                  "avg_launch_ms": round(dense_ms, 5),
                                  "achieved": round(dense_bytes / dense_ms / 1e6, 1),
                                  "frac": round(dense_bytes / dense_ms / 1e6 / (HBM_PEAK_BPS / 1e9), 4),
                                  "frac_on_survey_bytes": round(dense_slot_bytes / dense_ms / 1e6 / (HBM_PEAK_BPS / 1e9), 4),
                                  "track_ids": "every slot draws one of eight: 4-5 different track ids per voxel, the worst case for the vote",
                                  "launches_timed": 10}
        # the same with the track ids of a real map: a voxel holds particles of ONE surface, i.e. one track id (one voxel
        # in 16 two: object borders) - the vote then takes its single-track path
        m.fill_dense_ex(1)
        m.time_occupancy_sweep(iters=40)
        surf_ms = m.time_occupancy_sweep(iters=10)
        roofline["dense_case_surface"] = {"kernel": roofline["dense_case"]["kernel"], "bytes_per_launch": dense_bytes,
                                          "avg_launch_ms": round(surf_ms, 5), "achieved": round(dense_bytes / surf_ms / 1e6, 1),
                                          "frac": round(dense_bytes / surf_ms / 1e6 / (HBM_PEAK_BPS / 1e9), 4),
                                          "frac_on_survey_bytes": round(dense_slot_bytes / surf_ms / 1e6 / (HBM_PEAK_BPS / 1e9), 4),
                                          "track_ids": "every voxel draws one track id, one voxel in 16 two", "launches_timed": 10}

        roofline["survey_dense_frac"] = roofline["dense_case"]["frac_on_survey_bytes"]  # SURVEY.md 8(d)'s 80 B/voxel view, one key away
        # A13: RingBufferOperations::clear (mc_ring/operations.h:684-723, "290 ms" single thread at :700) on this map -
        # positions, weights, stamps, status of every slot reset (the forget counts stay), owner sets and results emptied.
        # Wall clock around sdm_clear + synchronize, three times.
        t0 = time.perf_counter()
        for _ in range(3):
            m.clear()
            m.synchronize()
        clear_ms = (time.perf_counter() - t0) * 1e3 / 3
        n_slots = V * S
        clear_bytes = n_slots * (16 + 2) + V * (S - 1) * (4 + 2 + 1) + V * (2 + 1 + 8)  # pos4, owner per slot; w, ts, status per particle slot (nothing read); vts, vflag, res
        roofline["clear"] = {"kernel": "sdm_clear: k_clear_map<8> (one write-only pass over the map, 16-byte lane-linear stores) + 4 small memsets", "bytes_per_call": clear_bytes,
                             "ms_per_call": round(clear_ms, 4), "achieved": round(clear_bytes / clear_ms / 1e6, 1),
                             "frac": round(clear_bytes / clear_ms / 1e6 / (HBM_PEAK_BPS / 1e9), 4),
                             "frac_on_survey_bytes": round(n_slots * 34 / clear_ms / 1e6 / (HBM_PEAK_BPS / 1e9), 4),
                             "survey_bytes": n_slots * 34, "calls_timed": 3}

    # ---- strong scaling (BASELINE.json C4): the same 256^3 map with 8 M particles, split into `world` Z slabs.  Its own
    # map and frames; a separate object in the line (the headline value above stays the weak-scaling one).
    strong = None
    if not args.no_strong:
        m.close()
        base4 = synth.CONFIGS["C4"]
        strong, eng4, _, _, _ = timed_run(synth, sharded, dist, rank, world, local_rank, base4, synth.PARAMS[synth.CONFIG_PARAMS["C4"]],
                                          8000000 // world, min(args.steps, 10), 6,  # warm-up: the state's first frame, three timed launch-by-launch frames, the graph capture
                                          dict(n_static=48, n_dynamic=6, seed=7))
        strong["config"] = "C4: 256x256x256 voxels, 8 slots/voxel, 8 M particles prefilled in all, Z-slab shards over %d GPU(s)" % world
        strong["scaling"] = "strong"
        eng4.map.close()

    # The N > 1 code path with the one rank a one-GPU box has: the same map, the same frames, issued through
    # sdm_update_sharded on an RCCL communicator of one rank (member-count all-gather, export all-to-all, ck chunk exchange
    # + slab-ordered sum + all-gather, every one of them issued).  What it costs OVER the plain frame above is what the
    # sharded path adds before a single byte crosses xGMI; the driver sees it in every --gpus 1 line.
    sharded_one = None
    if not multi and not args.no_sharded_leg and (not args.no_cpu or args.sharded_leg):
        try:
            m.close()  # (a map created while another is alive runs slower for its whole life: DESIGN.md 9)
            sharded_one = {}
            for exch in ("rccl", "ipc"):
                s1, eng1, _, _, _ = timed_run(synth, sharded, None, 0, 1, 0, cfg, params, args.particles, args.steps, max(args.warmup, 6),
                                              dict(n_static=48, n_dynamic=6, seed=7), force_comm=True, exchange=exch)
                m1 = eng1.map
                m1.comm_timing(True)
                comm1 = {}
                sc1 = synth.Scene(cfg, n_static=48, n_dynamic=6, seed=7)
                for t in range(n_frames, n_frames + 6):
                    depth, cloud, pos, q = sc1.render(t, params)
                    eng1.update(m1.device_put(depth), m1.device_put(cloud), pos, q, sc1.moves(t))
                    m1.synchronize()
                    for k, v in m1.comm_times().items():
                        comm1.setdefault(k, []).append(v)
                m1.comm_timing(False)
                leg = {"ms_per_step": s1["ms_per_step"], "over_plain_frame_ms": round(s1["ms_per_step"] - ms_per_step, 4),
                       "steps": s1["steps"], "warmup": s1["warmup"], "live_particles": s1["live_particles"],
                       "host_enqueue_ms_per_step": s1["host_enqueue_ms_per_step"],
                       "exchanges_us": {k: round(float(np.mean(v)), 1) for k, v in sorted(comm1.items())}}
                m1.close()
                if exch == "rccl":
                    sharded_one = dict(leg, path="sdm_update_sharded, RCCL communicator of 1 rank, launch by launch (sharded frames are never replayed from a graph)",
                                       ck_exchange=os.environ.get("SDM_CK_EXCHANGE", "chunks"))
                else:
                    sharded_one["ipc_exchange"] = dict(leg, path="the same frame, its exchanges through the shard's hipIpc arena (sdm_ipc_create / sdm_ipc_connect): "
                                                                 "one k_ipc_exchange launch each, no RCCL")
        except Exception as e:  # noqa: BLE001 - a side leg must not take the line down
            sharded_one = {"error": "%s: %s" % (type(e).__name__, e)}

    # The side legs: maps of their own in this process, one after the other.  (Round 3 ran them as child processes: the
    # second and later maps of a process ran 17-40 us per frame slower than its first - the launch-mode policy took each
    # map's own, noisier measurement of the host's speed and replayed later maps from graphs; fixed in sdm_create.)
    stress = None
    if not multi and not args.no_stress and not args.no_cpu:
        stress = stress_run(synth, sharded)
        stress["process"] = "this process (a later map of it)"

    driven = None
    if not multi and not args.no_driven and not args.no_cpu:
        driven = driven_run(synth, sharded)
        driven["process"] = "this process (a later map of it)"

    adapter = None
    if not multi and not args.no_adapter and not args.no_cpu:
        adapter = adapter_e2e(synth, cfg, params, scene, frames)

    if rank == 0:
        out = {
            "metric": "Mvoxels updated/sec (256^3 grid, 2M particles, VKITTI2 camera; whole hot-path frame)",
            "value": round(value, 1), "unit": "Mvoxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "issue_mode": issue_mode,
            "config": {"workload": "%s: %dx%dx%d voxels, %d slots/voxel, %dx%d image, window %d, %s params, "
                                   "6 dynamic objects, %d live particles - what the first sweeps left of %d PREFILLED ones, FILLER the camera "
                                   "cannot see (below the ground plane, behind the walls: the map carries BASELINE's 2 M, the frame works on "
                                   "what is visible) -, %d visible/frame; the workload the filter populated itself is `driven`"
                                   % (args.config if world == 1 else "%s weak-scaled x%d" % (args.config, world),
                                      1 << cfg["x_n"], 1 << cfg["y_n"], 1 << cfg["z_n"], S, cfg["width"], cfg["height"],
                                      cfg["window_half"], synth.CONFIG_PARAMS[args.config], live, n_pre * world, n_vis),
                       "voxels": V, "live_particles": live, "prefilled_particles": n_pre * world, "visible_particles": n_vis,
                       "parallelism": "zslab%d" % world, "inputs": "depth + LabeledPoint image resident in HBM",
                       "render_s": round(t_render, 1),
                       "host_enqueue_ms_per_step": round(t_enqueue * 1e3 / args.steps, 4), "launch_mode": launch_mode,
                       "host_numa": host_numa},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        if collectives is not None:
            out["collectives_us"] = collectives
        if strong is not None:
            out["strong_scaling"] = strong
        if sharded_one is not None:
            out["sharded_one_rank"] = sharded_one
        if stress is not None:
            out["stress"] = stress
        if driven is not None:
            out["driven"] = driven
        if adapter is not None:
            out["adapter_e2e"] = adapter
        if not multi:
            out["stage_ms"] = {k: round(stage_ms[i], 4) for i, k in
                               enumerate(["", "ego", "move", "remove", "visibility", "weight", "birth", "occupancy"]) if k}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def adapter_e2e(synth, cfg, params, scene, frames, n_frames=12):
    """The drop-in class end to end: include/semantic_dsp_map.h (SemanticDSPMap::update, the reference's API) compiled
    against the stand-in Eigen / OpenCV / PCL headers of tests/mock_includes (none of the three is in the image; only their
    layouts matter here) and linked with libsdm_hip.so, fed the benchmark's first frames from HOST buffers the way
    src/mapping.cpp feeds the reference: depth cv::Mat + MaskKpts (static mask, six object masks with key points) + pose in,
    occupied cloud out - mask packing, object layer, uploads, the frame, sdm_synchronize and the download of the cloud
    included.  A map of its own, grown from empty (the class has no way to load a state), in a process of its own."""
    import subprocess
    import tempfile
    from tests import adapter_clip
    try:
        tmp = os.environ.get("SDM_ADAPTER_DIR")  # (profiling: keep the driver and its clip where a tracer can run them)
        if tmp:
            os.makedirs(tmp, exist_ok=True)
        else:
            tmp = tempfile.mkdtemp(prefix="sdm_e2e_")
        exe, clip, out = os.path.join(tmp, "adapter_e2e"), os.path.join(tmp, "clip.bin"), os.path.join(tmp, "out.bin")
        csrc = os.path.join(ROOT, "semantic_dsp_map_amd", "csrc")
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "tests", "mock_includes"), "-I", os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tests", "cpp", "adapter_parity.cpp"), "-o", exe, "-L", csrc, "-lsdm_hip",
                               "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib"])
        preset = dict(x_n=cfg["x_n"], y_n=cfg["y_n"], z_n=cfg["z_n"], p_n=cfg["p_n"], voxel_size=cfg["voxel_size"], fx=cfg["fx"], fy=cfg["fy"],
                      cx=cfg["cx"], cy=cfg["cy"], width=cfg["width"], height=cfg["height"], depth_min=cfg["depth_min"],
                      depth_max=cfg["depth_max"], window_half=cfg["window_half"], consider_instance=True, src_width=0, src_height=0,
                      rescale=1.0, zed2_filters=False, object_mode=2)

        def corners(b):
            return np.array([[x, y, z] for x in (b[0], b[3]) for y in (b[1], b[4]) for z in (b[2], b[5])], np.float64)

        clip_frames = []
        n = min(n_frames, len(frames))
        for t in range(n):
            depth, cloud = frames[t][0], frames[t][1]
            static_mask, objects = synth.raw_inputs(cfg, cloud, scene)
            pos, q = scene.pose(t)
            boxes, prev = scene.dyn_boxes(t), scene.dyn_boxes(t - 1 if t else 0)
            seg = [dict(track_id=65535, label="static", kpts_current=np.zeros((0, 3)), kpts_previous=None, mask=static_mask)]
            for i, (trk, _lab, mask) in enumerate(objects):
                seg.append(dict(track_id=int(trk), label="Car", kpts_current=corners(boxes[i]), kpts_previous=corners(prev[i]), mask=mask))
            clip_frames.append(dict(depth=depth, seg=seg, pos=pos, q=q, ts=0.1 * t, free=False))
        adapter_clip.write_binary(clip, preset, params, (0.1, 0.69, 0.2, 0.1), synth.noise_table(n=100003), clip_frames, False)
        r = subprocess.run([exe, clip, out, "time"], capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("e2e median_ms_per_update")][-1].split()
        last = [l for l in r.stdout.splitlines() if l.startswith("frame ")][-1]
        V = 1 << (cfg["x_n"] + cfg["y_n"] + cfg["z_n"])
        med = float(line[2])
        # SemanticDSPMap::lastUpdateTimes() of the timed calls (the first two are warm-up), medians
        import re
        ph = [re.search(r"objects ([\d.]+), pack ([\d.]+), frame ([\d.]+), emit ([\d.]+)", l) for l in r.stdout.splitlines() if l.startswith("frame ")][2:]
        ph = np.array([[float(x) for x in mm.groups()] for mm in ph if mm])
        phases = dict(zip(("objects", "pack", "frame_upload_and_enqueue", "wait_and_emit"), np.round(np.median(ph, axis=0), 4).tolist())) if len(ph) else None
        return {"ms_per_update": med, "min_ms_per_update": float(line[4]), "frames_timed": int(line[6]), "phase_ms": phases,
                "value": round(V / med / 1e3, 1), "unit": "Mvoxels/s",
                "path": "SemanticDSPMap::update (include/semantic_dsp_map.h): host cv::Mat depth + 7 MaskKpts in, pcl::PointXYZRGB cloud out, "
                        "per call: mask packing, built-in object layer, H2D of depth + masks (5.1 MB), the frame, sdm_synchronize, D2H of the cloud",
                "map": "grown from empty over the %d calls (%s)" % (n, last.split(":")[1].split(",")[0].strip()),
                "headers": "compiled against tests/mock_includes (stand-ins for Eigen / OpenCV / PCL: none is in the image)"}
    except Exception as e:  # noqa: BLE001 - a side leg must not take the line down
        return {"error": "%s: %s" % (type(e).__name__, e)}


PMC_FILES = ("r06_sweep_pmc.json",)  # this layout only (round 6: 70-byte records)


def pmc_traffic(S, voxels, evaluated, tiles):
    """(HBM bytes per launch of the sweep kernel, the committed file they come from).  PMC counters cannot be read from
    inside this process: the figure is taken from the rocprofv3 PMC passes over this very command that are committed
    under profiles/ (FETCH_SIZE x2 per MI355X_MICROARCH.md + WRITE_SIZE, separate runs; tools/pmc_sweep.sh) - it is NOT a
    measurement of this run, `traffic_source` says so in the line.  (None, None) if no committed pass matches: different
    kernel shape, or a number of fully evaluated voxels or of visited tiles more than 25 % off."""
    for name in PMC_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                p = json.load(f)
            if (p["kernel"] == "k_occupancy<%d>" % S and p["voxels"] == voxels
                    and abs(p["voxels_evaluated_in_full"] - evaluated) <= 0.25 * max(evaluated, 1)
                    and abs(p["tiles_looked_into"] - tiles) <= 0.25 * max(tiles, 1)):
                return int(p["traffic_bytes_per_launch"]), "profiles/" + name
        except (OSError, KeyError, ValueError):
            pass
    return None, None


def cpu_baseline(cfg, params, noise, frames, st, ring, n_frames, V):
    """The oracle in its literal order (bin_order 0 = the reference's BFS push order), one thread pinned to one core
    (taskset), same prefilled map and same frames: 3 warm-up + n_frames timed frames (a few seconds of
    CPU work)."""
    from oracle import oracle as orc
    pinned = None
    try:
        old_aff = os.sched_getaffinity(0)
        pinned = max(old_aff)
        os.sched_setaffinity(0, {pinned})
    except (AttributeError, OSError):
        old_aff = None
    try:
        o = orc.OracleMap(dict(cfg, bin_order=0), params, noise)
        o.load_state(st)
        o.set_ring_state(ring)
        warm = 3
        times = []
        stages = np.zeros(8)
        for t in range(min(warm + n_frames, len(frames))):
            depth, cloud, pos, q, moves = frames[t][:5]
            t0 = time.perf_counter()
            o.update(depth, cloud, pos, q, moves)
            dt = time.perf_counter() - t0
            if t >= warm:
                times.append(dt)
                stages += np.array(o.stats()["stage_ms"])
    finally:
        if old_aff is not None:
            os.sched_setaffinity(0, old_aff)
    med = float(np.median(times))
    return {"value": round(V / med / 1e6, 2), "unit": "Mvoxels/s", "cores": 1, "kind": "port",
            "sample": "%d frames of the same workload after %d warm-up frames, median %.1f ms/frame (min %.1f); "
                      "g++ -O3 -march=native -ffp-contract=off, 1 thread pinned to core %s of %d host cores"
                      % (len(times), warm, med * 1e3, min(times) * 1e3, pinned, os.cpu_count()),
            "ms_per_frame": round(med * 1e3, 2), "min_ms_per_frame": round(min(times) * 1e3, 2),
            "stage_ms": {k: round(stages[i] / len(times), 2) for i, k in
                         enumerate(["", "ego", "move", "remove", "visibility", "weight", "birth", "occupancy"]) if k}}


if __name__ == "__main__":
    main()
