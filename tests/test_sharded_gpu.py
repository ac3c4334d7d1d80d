"""Z-slab sharding exercised on ONE GPU: G shard maps live side by side on device 0, the test plays
the role of the collectives (concatenating the per-shard buffers in slab order, exactly what all_gather does) and the
union of the shard states must equal the oracle run with the same slab-ordered ck summation (ck_slabs = G)."""
import numpy as np
import pytest

from oracle import oracle as orc
from semantic_dsp_map_amd import binding, synth
from tests import parity_utils as pu

pytestmark = pytest.mark.gpu


def run_sharded_frame(shards, torch, frame, gathered, parts):
    depth, cloud, pos, q, moves = frame
    G = len(shards)
    hw = depth.size
    for r, g in enumerate(shards):
        g.update_begin(depth, cloud, pos, q, moves)
    for g in shards:
        g.synchronize()
    allp = torch.cat(parts)
    for r, g in enumerate(shards):
        gathered[r].copy_(allp)
    torch.cuda.synchronize()
    for r, g in enumerate(shards):
        g.update_finish(gathered[r].data_ptr(), G)
    for g in shards:
        g.synchronize()


def union_state(shards):
    sts = [g.dump_state() for g in shards]
    return {k: np.concatenate([s[k] for s in sts]) for k in sts[0]}


@pytest.mark.parametrize("G,cfg_name,params_name,n_frames,kw", [
    (2, "T0", "vkitti2", 6, dict(n_dynamic=0)),
    (4, "T1", "zed2", 5, dict(n_dynamic=0)),
    (2, "T0", "noisy3", 5, dict(n_dynamic=0)),
])
def test_static_scene_shards_match_oracle(G, cfg_name, params_name, n_frames, kw):
    import torch
    cfg, params, frames = synth.make_frames(cfg_name, n_frames, params_name, **kw)
    noise = synth.noise_table()
    o = orc.OracleMap(dict(cfg, bin_order=1, ck_slabs=G), params, noise)
    shards = [binding.SdmMap(cfg, params, noise, shard_rank=r, shard_count=G) for r in range(G)]
    hw = cfg["width"] * cfg["height"]
    dev = torch.device("cuda", 0)
    parts = [torch.zeros(hw, dtype=torch.float32, device=dev) for _ in range(G)]
    gathered = [torch.zeros(G * hw, dtype=torch.float32, device=dev) for _ in range(G)]
    for r, g in enumerate(shards):
        g.set_ck_buffer(parts[r].data_ptr())
    S = 1 << cfg["p_n"]
    for t, frame in enumerate(frames):
        o.update(*frame)
        run_sharded_frame(shards, torch, frame, gathered, parts)
        so, sg = o.dump_state(), union_state(shards)
        for k in pu.STATE_KEYS:
            r = pu.diff_report("frame %d state.%s" % (t, k), so[k], sg[k])
            assert r is None, r
        vo = o.voxels()
        vg = np.concatenate([g.voxels() for g in shards])
        for k in ("occ", "label", "track", "wsum"):
            r = pu.diff_report("frame %d voxels.%s" % (t, k), vo[k], vg[k])
            assert r is None, r
        assert sum(g.stats()["n_visible"] for g in shards) == o.stats()["n_visible"]
    for g in shards:
        g.close()


class LocalGather:
    """torch.distributed stand-in for G engines that live in one process: gathers by concatenation in rank order."""

    def __init__(self, torch):
        self.torch = torch


def run_sharded_dynamic_frame(engines, torch, frame):
    depth, cloud, pos, q, moves = frame
    G = len(engines)
    has_moves = len(moves) > 0
    for e in engines:
        e.start(depth, cloud, pos, q, moves, on_device=False)
    if has_moves:
        torch.cuda.synchronize()
        allc = torch.cat([e.counts_local for e in engines])
        for e in engines:
            e.counts_all.copy_(allc)
        torch.cuda.synchronize()
    for e in engines:
        e.moves()
    exported = 0
    if has_moves:
        torch.cuda.synchronize()
        alls = torch.cat([e.halo_send for e in engines])
        exported = sum(int(e.halo_send[:4].view(torch.int32).item()) for e in engines)
        for e in engines:
            e.halo_recv.copy_(alls)
        torch.cuda.synchronize()
    for e in engines:
        e.predict()
    torch.cuda.synchronize()
    allp = torch.cat([e.part for e in engines])
    for e in engines:
        e.gathered.copy_(allp)
    torch.cuda.synchronize()
    for e in engines:
        e.finish(e.gathered, G)
    for e in engines:
        e.synchronize()
    return exported


@pytest.mark.parametrize("G,cfg_name,params_name,n_frames,kw", [
    (4, "T0", "vkitti2", 9, dict(n_dynamic=3, dyn_speed=(0.8, 1.6))),
    (2, "T1", "zed2", 7, dict(n_dynamic=2, dyn_speed=(0.8, 1.6), speed=0.6)),
])
def test_moving_objects_cross_slabs(G, cfg_name, params_name, n_frames, kw):
    """Objects drive through slab borders: the halo exchange must reproduce the single-map result."""
    import torch
    from semantic_dsp_map_amd import sharded
    cfg, params, frames = synth.make_frames(cfg_name, n_frames, params_name, **kw)
    noise = synth.noise_table()
    o = orc.OracleMap(dict(cfg, bin_order=1, ck_slabs=G), params, noise)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        engines = [sharded.HipEngine(cfg, params, r, G, 0, noise_table=noise, halo_cap=4096) for r in range(G)]
        exported = 0
        for t, frame in enumerate(frames):
            o.update(*frame)
            exported += run_sharded_dynamic_frame(engines, torch, frame)
            so = o.dump_state()
            sts = [e.map.dump_state() for e in engines]
            sg = {k: np.concatenate([s[k] for s in sts]) for k in sts[0]}
            for k in pu.STATE_KEYS:
                r = pu.diff_report("frame %d state.%s" % (t, k), so[k], sg[k])
                assert r is None, r
            assert o.stats()["alias_events"] == 0
            ro = o.ring_state()
            for e in engines:
                assert e.map.ring_state()["move_cursor"] == ro["move_cursor"]
        assert exported > 0, "no particle crossed a slab border: the test does not exercise the halo exchange"
        for e in engines:
            e.map.close()
