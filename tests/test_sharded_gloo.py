"""The multi-shard frame protocol (semantic_dsp_map_amd/sharded.py) on CPU: 2 processes, gloo backend.

What can be checked without a GPU: the order and content of the three exchanges of ShardedDriver (counts ->
exports -> ck images, the first two only when objects move), that every rank ends up with every shard's buffers in
shard order, the RCCL-id rendezvous (broadcast from rank 0), and the weak-scaling grid rule.  The engine is a CPU
stand-in that fills its buffers with rank-dependent patterns; the kernels behind the real engine are tested on the
GPU in tests/test_sharded_gpu.py."""
import os
import socket

import pytest

from semantic_dsp_map_amd import sharded


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class FakeEngine:
    def __init__(self, torch, rank, world, hw=24, cap=8):
        self.t, self.rank, self.world, self.hw = torch, rank, world, hw
        nb = sharded.HALO_HEADER_BYTES + cap * sharded.HALO_RECORD_BYTES
        self.counts_local = torch.zeros(sharded.HALO_OBJ, dtype=torch.int32)
        self.counts_all = torch.full((world * sharded.HALO_OBJ,), -1, dtype=torch.int32)
        self.halo_send = torch.zeros(nb, dtype=torch.uint8)
        self.halo_recv = torch.zeros(world * nb, dtype=torch.uint8)
        self.part = torch.zeros(hw, dtype=torch.float32)
        self.gathered = torch.zeros(world * hw, dtype=torch.float32)
        self.log = []

    def start(self, depth, cloud, pos, q, moves=None, remove_tracks=None, **kw):
        self.log.append("start")
        self.counts_local[:] = 0
        if moves is not None and len(moves):
            self.counts_local[:len(moves)] = self.t.arange(len(moves), dtype=self.t.int32) + 10 * (self.rank + 1)

    def moves(self):
        self.log.append("moves")
        self.halo_send[:] = self.rank + 1

    def predict(self):
        self.log.append("predict")
        self.part[:] = float(self.rank) + self.t.arange(self.hw, dtype=self.t.float32) / 100
        return self.part

    def finish(self, gathered, n_parts):
        self.log.append("finish%d" % n_parts)
        self.final = gathered.clone()


def worker(rank, world, port, results):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = FakeEngine(torch, rank, world)
        drv = sharded.ShardedDriver(eng, rank, world, dist)
        # frame 1: no moving object -> only the ck exchange
        drv.update(None, None, None, None, moves=[])
        assert eng.log == ["start", "moves", "predict", "finish%d" % world]
        assert int(eng.counts_all[0]) == -1 and int(eng.halo_recv.sum()) == 0      # untouched
        for r in range(world):
            want = float(r) + torch.arange(eng.hw, dtype=torch.float32) / 100
            assert torch.equal(eng.final[r * eng.hw:(r + 1) * eng.hw], want)        # shard order
        # frame 2: two moving objects -> all three exchanges
        drv.update(None, None, None, None, moves=[1, 2])
        for r in range(world):
            row = eng.counts_all[r * sharded.HALO_OBJ:(r + 1) * sharded.HALO_OBJ]
            assert int(row[0]) == 10 * (r + 1) and int(row[1]) == 10 * (r + 1) + 1 and int(row[2]) == 0
            nb = eng.halo_send.numel()
            assert torch.all(eng.halo_recv[r * nb:(r + 1) * nb] == r + 1)
        # rendezvous of the RCCL id: everybody gets rank 0's bytes
        from semantic_dsp_map_amd import binding
        binding.comm_unique_id = lambda: bytes([(7 * i + 3) % 256 for i in range(128)]) if rank == 0 else b"\0" * 128
        got = sharded.broadcast_unique_id(dist, rank)
        assert got == bytes([(7 * i + 3) % 256 for i in range(128)])
        results[rank] = "ok"
    except Exception as e:  # noqa: BLE001
        results[rank] = "FAILED: %r" % (e,)
    finally:
        dist.destroy_process_group()


def test_sharded_driver_two_processes_gloo():
    import torch.multiprocessing as mp
    world = 2
    port = free_port()
    mgr = mp.get_context("spawn").Manager()
    results = mgr.dict()
    mp.spawn(worker, args=(world, port, results), nprocs=world, join=True)
    assert dict(results) == {0: "ok", 1: "ok"}, dict(results)


def test_weak_scaled_config_matches_baseline_configs():
    from semantic_dsp_map_amd import synth
    c3 = synth.CONFIGS["C3"]
    assert sharded.weak_scaled_config(c3, 1) == c3
    c8 = sharded.weak_scaled_config(c3, 8)
    assert (c8["x_n"], c8["y_n"], c8["z_n"]) == (9, 9, 9) == tuple(synth.CONFIGS["C5"][k] for k in ("x_n", "y_n", "z_n"))
    for w in (1, 2, 4, 8):
        c = sharded.weak_scaled_config(c3, w)
        assert (1 << (c["x_n"] + c["y_n"] + c["z_n"])) // w == 1 << 24          # constant voxels per GPU
        assert (1 << c["z_n"]) % w == 0                                          # slabs divide the z axis
    with pytest.raises(AssertionError):
        sharded.weak_scaled_config(c3, 3)
