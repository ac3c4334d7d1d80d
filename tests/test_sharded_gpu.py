"""Z-slab sharding exercised on ONE GPU: G shard maps live side by side on device 0, the test plays
the role of the collectives (member counts: all-gather; export segments: all-to-all; partial ck chunks: all-to-all, then
the summed chunks: all-gather) and the union of the shard states must equal the oracle run with the same slab-ordered ck
summation (ck_slabs = G).  The RCCL calls themselves are exercised with a 1-rank communicator
(test_native_rccl_single_rank); the same protocol with one PROCESS per shard in tests/test_sharded_multiprocess_gpu.py."""
import numpy as np
import pytest

from oracle import oracle as orc
from semantic_dsp_map_amd import binding, sharded, synth
from tests import parity_utils as pu

pytestmark = pytest.mark.gpu


class Shard:
    def __init__(self, cfg, params, noise, r, G, halo_cap=1024):
        self.m = m = binding.SdmMap(cfg, params, noise, shard_rank=r, shard_count=G)
        self.hw = cfg["width"] * cfg["height"]
        self.chunk = m.ck_chunk_elems()
        assert self.chunk * G >= self.hw and self.chunk % 64 == 0
        self.seg = sharded.halo_segment_bytes(halo_cap)
        self.ck_part = m.device_put(np.zeros(G * self.chunk, np.float32))
        self.ck_stage = m.device_put(np.zeros(G * self.chunk, np.float32))
        self.ck_full = m.device_put(np.zeros(G * self.chunk, np.float32))
        self.counts_local = m.device_alloc(sharded.HALO_OBJ * 4)
        self.counts_all = m.device_alloc(G * sharded.HALO_OBJ * 4)
        self.send = m.device_put(np.zeros(G * self.seg, np.uint8))
        self.recv = m.device_put(np.zeros(G * self.seg, np.uint8))
        m.set_ck_buffer(self.ck_part)
        m.set_halo_buffers(self.counts_local, self.counts_all, self.send, self.recv, halo_cap)


def gather(shards, src_attr, dst_attr, nbytes, src_offset=lambda r: 0):
    """all-gather: every shard's `nbytes` at src, concatenated in shard order, land at every shard's dst"""
    allb = np.concatenate([s.m.device_download(getattr(s, src_attr) + src_offset(r), nbytes) for r, s in enumerate(shards)])
    for s in shards:
        s.m.device_upload(getattr(s, dst_attr), allb)
    return allb


def all_to_all(shards, src_attr, dst_attr, piece):
    """all-to-all: piece d of shard s's src becomes piece s of shard d's dst"""
    G = len(shards)
    src = [s.m.device_download(getattr(s, src_attr), G * piece).reshape(G, piece) for s in shards]
    for d, s in enumerate(shards):
        s.m.device_upload(getattr(s, dst_attr), np.concatenate([src[r][d] for r in range(G)]))
    return src


def run_frame(shards, frame, remove_tracks=None):
    depth, cloud, pos, q, moves = frame
    G = len(shards)
    has_moves = len(moves) > 0
    for s in shards:
        s.m.frame_start(depth, cloud, pos, q, moves, remove_tracks)
    if has_moves:
        gather(shards, "counts_local", "counts_all", sharded.HALO_OBJ * 4)
    for s in shards:
        s.m.frame_moves()
    while shards[0].m.frame_moves_pending():   # a list of more than SDM_MAX_MOVES objects: batch by batch (every shard alike)
        assert all(s.m.frame_moves_pending() for s in shards)
        gather(shards, "counts_local", "counts_all", sharded.HALO_OBJ * 4)
        for s in shards:
            s.m.frame_moves()
    exported = 0
    if has_moves:
        src = all_to_all(shards, "send", "recv", shards[0].seg)
        exported = int(sum(int(x[:, :4].copy().view(np.uint32).sum()) for x in src))
    for s in shards:
        s.m.frame_predict()
    all_to_all(shards, "ck_part", "ck_stage", shards[0].chunk * 4)
    for s in shards:
        s.m.ck_reduce(s.ck_stage, s.ck_full)
    gather(shards, "ck_full", "ck_full", shards[0].chunk * 4, src_offset=lambda r: r * shards[0].chunk * 4)
    for s in shards:
        s.m.update_finish(s.ck_full, 1)
    for s in shards:
        s.m.synchronize()
    return exported


def compare_union(o, shards, t, S):
    so = o.dump_state()
    sts = [s.m.dump_state() for s in shards]
    sg = {k: np.concatenate([st[k] for st in sts]) for k in sts[0]}
    for k in pu.STATE_KEYS:
        r = pu.diff_report("frame %d state.%s" % (t, k), so[k], sg[k])
        assert r is None, r
    vo = o.voxels()
    vg = np.concatenate([s.m.voxels() for s in shards])
    for k in ("occ", "label", "track", "wsum"):
        r = pu.diff_report("frame %d voxels.%s" % (t, k), vo[k], vg[k])
        assert r is None, r
    assert sum(s.m.stats()["n_visible"] for s in shards) == o.stats()["n_visible"]
    ro = o.ring_state()
    for s in shards:
        rs = s.m.ring_state()
        assert rs["move_cursor"] == ro["move_cursor"] and rs["birth_cursor"] == ro["birth_cursor"]


@pytest.mark.parametrize("G,cfg_name,params_name,n_frames,kw", [
    (2, "T0", "vkitti2", 6, dict(n_dynamic=0)),
    (4, "T1", "zed2", 5, dict(n_dynamic=0)),
    (2, "T0", "noisy3", 5, dict(n_dynamic=0)),
    (4, "T0", "vkitti2", 9, dict(n_dynamic=3, dyn_speed=(0.8, 1.6))),
    (2, "T1", "zed2", 7, dict(n_dynamic=2, dyn_speed=(0.8, 1.6), speed=0.6)),
    (4, "T0", "vkitti2", 60, dict(n_dynamic=3, seed=21, yaw_rate_deg=2.0)),   # long: ring shifts, re-used owner slots
])
def test_shards_match_oracle(G, cfg_name, params_name, n_frames, kw):
    cfg, params, frames = synth.make_frames(cfg_name, n_frames, params_name, **kw)
    noise = synth.noise_table()
    o = orc.OracleMap(dict(cfg, bin_order=1, ck_slabs=G), params, noise)
    shards = [Shard(cfg, params, noise, r, G) for r in range(G)]
    S = 1 << cfg["p_n"]
    exported = 0
    for t, frame in enumerate(frames):
        o.update(*frame)
        exported += run_frame(shards, frame)
        compare_union(o, shards, t, S)
    if kw.get("n_dynamic", 0) > 0:
        assert exported > 0, "no particle crossed a slab border: the halo exchange was not exercised"
    for s in shards:
        s.m.close()


def test_native_rccl_single_rank():
    """sdm_comm_init + sdm_update_sharded with a 1-rank RCCL communicator: same result as the plain frame."""
    cfg, params, frames = synth.make_frames("T0", 4, "vkitti2", n_dynamic=2)
    noise = synth.noise_table()
    a = binding.SdmMap(cfg, params, noise)
    b = binding.SdmMap(cfg, params, noise)
    b.comm_init(binding.comm_unique_id(), 1024)
    b.comm_timing(True)
    for depth, cloud, pos, q, moves in frames:
        a.update(depth, cloud, pos, q, moves, sync=True)
        b.update_sharded(depth, cloud, pos, q, moves)
        b.synchronize()
        ct = b.comm_times()
        assert set(ct) == {"counts_allgather", "halo_alltoall", "ck_alltoall", "ck_allgather"}
        assert ct["ck_alltoall"] > 0 and ct["ck_allgather"] > 0 and all(0 <= v < 1e5 for v in ct.values())
        assert (ct["counts_allgather"] > 0) == (len(moves) > 0) == (ct["halo_alltoall"] > 0)
    sa, sb = a.dump_state(), b.dump_state()
    for k in pu.STATE_KEYS:
        assert pu.diff_report(k, sa[k], sb[k]) is None
    assert np.array_equal(a.voxels(), b.voxels())
    a.close()
    b.close()


def test_native_rccl_single_rank_one_collective():
    """the ck exchange as ONE all-gather of the whole partial images (sdm_comm_set_options): same result, and the timers say
    which collectives ran"""
    cfg, params, frames = synth.make_frames("T0", 4, "vkitti2", n_dynamic=2)
    noise = synth.noise_table()
    a = binding.SdmMap(cfg, params, noise)
    b = binding.SdmMap(cfg, params, noise)
    b.comm_init(binding.comm_unique_id(), 1024)
    b.comm_set_options(ck_exchange=1, timeout_ms=20000)
    b.comm_timing(True)
    for depth, cloud, pos, q, moves in frames:
        a.update(depth, cloud, pos, q, moves, sync=True)
        b.update_sharded(depth, cloud, pos, q, moves)
        b.synchronize()
        ct = b.comm_times()
        assert ct["ck_alltoall"] > 0 and ct["ck_allgather"] == 0   # (slot 2 times the one collective, slot 3 did not run)
    sa, sb = a.dump_state(), b.dump_state()
    for k in pu.STATE_KEYS:
        assert pu.diff_report(k, sa[k], sb[k]) is None
    assert np.array_equal(a.voxels(), b.voxels())
    with pytest.raises(Exception):
        b.comm_set_options(ck_exchange=5)
    a.close()
    b.close()


def test_gloo_rendezvous_then_rccl_in_one_process():
    """The order bench.py uses at N > 1: torch + a gloo process group first (CPU only), then libsdm_hip, the RCCL id
    through sharded.broadcast_unique_id, sdm_comm_init and sharded frames.  Run in a fresh interpreter: torch brings
    its own HIP runtime into the process and must not disturb the one the library runs on."""
    import os
    import subprocess
    import sys
    code = r'''
import os, sys
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
import torch
import torch.distributed as dist
dist.init_process_group("gloo", rank=0, world_size=1)
import numpy as np
from semantic_dsp_map_amd import binding, sharded, synth
cfg, params, frames = synth.make_frames("T0", 3, "vkitti2", n_dynamic=2)
noise = synth.noise_table()
uid = sharded.broadcast_unique_id(dist, 0)
assert len(uid) == 128 and any(uid)
a = binding.SdmMap(cfg, params, noise)
b = binding.SdmMap(cfg, params, noise)
b.comm_init(uid, 1024)
for depth, cloud, pos, q, moves in frames:
    a.update(depth, cloud, pos, q, moves, sync=True)
    b.update_sharded(depth, cloud, pos, q, moves)
    b.synchronize()
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert np.array_equal(a.voxels(), b.voxels())
assert not torch.cuda.is_initialized()
dist.destroy_process_group()
print("OK")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("G,params_name,seed", [(2, "vkitti2", 5), (4, "noisy3", 6)])
def test_shards_match_oracle_on_random_frames(G, params_name, seed):
    """The seeded random frames of tests/test_fuzz_gpu.py (overlapping tracks, re-used owner slots, removals are left
    out: the split frame entry points take none) through G Z-slab shards against the unsharded oracle, 50 frames."""
    from tests.test_fuzz_gpu import random_frame
    cfg = synth.CONFIGS["T0"]
    params = synth.PARAMS[params_name]
    rng = np.random.default_rng(seed)
    noise = synth.noise_table()
    o = orc.OracleMap(dict(cfg, bin_order=1, ck_slabs=G), params, noise)
    shards = [Shard(cfg, params, noise, r, G) for r in range(G)]
    S = 1 << cfg["p_n"]
    pos = np.zeros(3)
    yaw = 0.0
    exported = 0
    for t in range(50):
        pos = pos + rng.normal(0, 0.35, 3) * np.array([1.0, 0.2, 1.0])
        yaw += rng.normal(0, 0.08)
        depth, cloud, mv, _ = random_frame(rng, cfg, params, t, pos, yaw)
        frame = (depth, cloud, pos.astype(np.float32), synth.yaw_quat(yaw).astype(np.float32), mv)
        o.update(*frame)
        exported += run_frame(shards, frame)
        if t % 10 == 9:
            compare_union(o, shards, t, S)
    assert exported > 0 and o.stats()["alias_events"] > 0
    for s in shards:
        s.m.close()


@pytest.mark.parametrize("G,params_name,seed", [(4, "vkitti2", 11), (4, "noisy3", 12)])
def test_shards_take_object_lists_of_any_length(G, params_name, seed):
    """200 moving objects and 300 removals in ONE frame of a Z-slab sharded map (the reference loops over whatever the object
    layer hands it, semantic_dsp_map.h:588-736): the frames of tests/test_long_object_lists_gpu.py through G shards - every
    batch of 48 objects with its own exchange of member counts, the copies of all batches in one export exchange, ranks and
    noise draws running on from batch to batch and from shard to shard - against the unsharded oracle, bit for bit."""
    from tests.test_long_object_lists_gpu import N_TRACKS, frame as ll_frame, moves_of
    cfg = synth.CONFIGS["T0"]
    params = synth.PARAMS[params_name]
    rng = np.random.default_rng(seed)
    noise = synth.noise_table()
    o = orc.OracleMap(dict(cfg, bin_order=1, ck_slabs=G), params, noise)
    shards = [Shard(cfg, params, noise, r, G, halo_cap=4096) for r in range(G)]
    S = 1 << cfg["p_n"]
    pos, yaw = np.zeros(3), 0.0
    exported, n_moved_max = 0, 0
    for t in range(8):
        pos = pos + rng.normal(0, 0.2, 3) * np.array([1.0, 0.1, 1.0])
        yaw += rng.normal(0, 0.03)
        depth, cloud = ll_frame(rng, cfg, params, pos, yaw)
        if t < 2:
            mv, remove = moves_of(rng, []), None
        elif t % 3 == 2:
            mv, remove = moves_of(rng, rng.permutation(np.arange(1, N_TRACKS + 1))[:200]), None
        elif t % 3 == 0:
            mv = moves_of(rng, rng.permutation(np.arange(1, N_TRACKS + 1))[:60])
            remove = [int(x) for x in rng.permutation(np.arange(1, 401))[:300]]
        else:
            mv = moves_of(rng, rng.permutation(np.arange(1, N_TRACKS + 1))[:150])
            remove = [int(x) for x in rng.permutation(np.arange(1, 401))[:140]]
        fr = (depth, cloud, pos.astype(np.float32), synth.yaw_quat(yaw).astype(np.float32), mv)
        o.update(*fr, remove)
        exported += run_frame(shards, fr, remove)
        compare_union(o, shards, t, S)
        n_moved_max = max(n_moved_max, o.stats()["n_moved"])
    assert exported > 0 and n_moved_max > 2000, (exported, n_moved_max)
    for s in shards:
        s.m.close()


def test_native_rccl_single_rank_long_lists():
    """sdm_update_sharded with more moving objects and removals than one frame block holds (a communicator of one rank: the
    count all-gather of every batch is issued): same result as the plain frame."""
    from tests.test_long_object_lists_gpu import N_TRACKS, frame as ll_frame, moves_of
    cfg, params = synth.CONFIGS["T0"], synth.PARAMS["vkitti2"]
    rng = np.random.default_rng(4)
    noise = synth.noise_table()
    a = binding.SdmMap(cfg, params, noise)
    b = binding.SdmMap(cfg, params, noise)
    b.comm_init(binding.comm_unique_id(), 1024)
    pos = np.zeros(3, np.float32)
    q = synth.yaw_quat(0.0).astype(np.float32)
    for t in range(5):
        depth, cloud = ll_frame(rng, cfg, params, pos.astype(np.float64), 0.0)
        mv = moves_of(rng, rng.permutation(np.arange(1, N_TRACKS + 1))[:130] if t >= 2 else [])
        remove = [int(x) for x in rng.permutation(np.arange(1, 401))[:200]] if t == 3 else None
        a.update(depth, cloud, pos, q, mv, remove, sync=True)
        b.update_sharded(depth, cloud, pos, q, mv, remove)
        b.synchronize()
    sa, sb = a.dump_state(), b.dump_state()
    for k in pu.STATE_KEYS:
        assert pu.diff_report(k, sa[k], sb[k]) is None
    assert np.array_equal(a.voxels(), b.voxels())
    assert a.stats()["n_moved"] == b.stats()["n_moved"] > 0
    a.close()
    b.close()


def test_native_ipc_single_rank():
    """sdm_ipc_create + sdm_ipc_connect + sdm_update_sharded with one shard: every exchange kernel is issued (member counts,
    export segments, ck parts, summed ck chunks) and the result is the plain frame's."""
    cfg, params, frames = synth.make_frames("T0", 5, "vkitti2", n_dynamic=2)
    noise = synth.noise_table()
    a = binding.SdmMap(cfg, params, noise)
    b = binding.SdmMap(cfg, params, noise)
    b.ipc_connect(b.ipc_create(1024))
    b.comm_timing(True)
    for depth, cloud, pos, q, moves in frames:
        a.update(depth, cloud, pos, q, moves, sync=True)
        b.update_sharded(depth, cloud, pos, q, moves)
        b.synchronize()
        ct = b.comm_times()
        assert ct["ck_alltoall"] > 0 and ct["ck_allgather"] == 0   # (the whole ck exchange is one launch, timed in the first slot)
        assert (ct["counts_allgather"] > 0) == (len(moves) > 0) == (ct["halo_alltoall"] > 0)
    sa, sb = a.dump_state(), b.dump_state()
    for k in pu.STATE_KEYS:
        assert pu.diff_report(k, sa[k], sb[k]) is None
    assert np.array_equal(a.voxels(), b.voxels())
    a.close()
    b.close()
