"""The real sharded engine with more than one rank: two PROCESSES, each with its own libsdm_hip shard map of a
Z-split map, both on GPU 0 (the boxes have one), the three per-frame exchanges carried by torch.distributed/gloo through
host tensors (semantic_dsp_map_amd.sharded.GlooShardEngine + ShardedDriver).  Every rank also runs the single-map
oracle (slab-ordered ck summation, ck_slabs = 2) and compares its own slab bit for bit after every frame: particle
state, results, cursors.  The scene's dynamic boxes drive along z, i.e. across the slab border of the ring."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def worker(rank, world, port, cfg_name, params_name, n_frames, scene_kw, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as orc
        from semantic_dsp_map_amd import sharded, synth
        from tests import parity_utils as pu
        cfg = synth.CONFIGS[cfg_name]
        params = synth.PARAMS[params_name]
        scene = synth.Scene(cfg, **scene_kw)
        noise = synth.noise_table()
        o = orc.OracleMap(dict(cfg, bin_order=1, ck_slabs=world), params, noise)
        eng = sharded.GlooShardEngine(cfg, params, rank, world, device=0, noise_table=noise)
        drv = sharded.ShardedDriver(eng, rank, world, dist)
        S = 1 << cfg["p_n"]
        V = 1 << (cfg["x_n"] + cfg["y_n"] + cfg["z_n"])
        lo, hi = rank * (V // world), (rank + 1) * (V // world)
        for t in range(n_frames):
            depth, cloud, pos, q = scene.render(t, params)
            moves = scene.moves(t)
            o.update(depth, cloud, pos, q, moves)
            drv.update(depth, cloud, pos, q, moves)
            eng.map.synchronize()
            so, sg = o.dump_state(), eng.map.dump_state()
            for k in pu.STATE_KEYS:
                r = pu.diff_report("rank %d frame %d state.%s" % (rank, t, k), so[k][lo * S:hi * S], sg[k])
                assert r is None, r
            vo, vg = o.voxels(), eng.map.voxels()
            for k in ("occ", "label", "track", "wsum"):
                r = pu.diff_report("rank %d frame %d voxels.%s" % (rank, t, k), vo[k][lo:hi], vg[k])
                assert r is None, r
            ro, rs = o.ring_state(), eng.map.ring_state()
            assert rs["move_cursor"] == ro["move_cursor"] and rs["birth_cursor"] == ro["birth_cursor"]
        live = eng.map.stats(count_live=True)["live_particles"]
        results[rank] = ("ok", dict(eng.bytes_exchanged), int(live), n_frames)
        eng.close()
    except Exception as e:  # noqa: BLE001
        import traceback
        results[rank] = ("FAILED: %r\n%s" % (e, traceback.format_exc()),)
    finally:
        dist.destroy_process_group()


def run(cfg_name, params_name, n_frames, scene_kw):
    import torch.multiprocessing as mp
    world = 2
    port = free_port()
    mgr = mp.get_context("spawn").Manager()
    results = mgr.dict()
    mp.spawn(worker, args=(world, port, cfg_name, params_name, n_frames, scene_kw, results), nprocs=world, join=True)
    res = dict(results)
    assert all(r[0] == "ok" for r in res.values()), res
    return res


def test_two_process_real_engine_gloo_small():
    res = run("T0", "vkitti2", 8, dict(n_dynamic=3, dyn_speed=(0.8, 1.6)))
    assert sum(r[1]["halo_records"] for r in res.values()) > 0, "no particle crossed the slab border"


def test_two_process_real_engine_gloo():
    """C3-sized map (256^3, 8 slots, 1242x375) split in two Z slabs, six dynamic objects."""
    res = run("C3", "vkitti2", 5, dict(n_static=48, n_dynamic=6, seed=7))
    ex = {k: sum(r[1][k] for r in res.values()) for k in res[0][1]}
    assert ex["halo_records"] > 0, "no particle crossed the slab border"
    print("exchanged per frame and rank (bytes): counts %d, halo %d (%d records in all), ck %d"
          % (ex["counts"] // (2 * res[0][3]), ex["halo"] // (2 * res[0][3]), ex["halo_records"], ex["ck"] // (2 * res[0][3])))
