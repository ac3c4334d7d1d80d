"""The multi-shard frame protocol (semantic_dsp_map_amd/sharded.py) on CPU: 2 and 4 processes, gloo backend.

What can be checked without a GPU: the order and content of the exchanges of ShardedDriver (member counts: all-gather ->
export segments: all-to-all -> partial ck chunks: all-to-all -> summed chunks: all-gather; the first two only when
objects move), that every piece lands where the protocol says (segment / part s of a receive buffer comes from shard s and
was addressed to this shard), the RCCL-id rendezvous (broadcast from rank 0), and the weak-scaling grid rule.  The engine
is a CPU stand-in that fills its buffers with patterns that encode (source, destination); the kernels behind the real
engine are tested on the GPU in tests/test_sharded_gpu.py and tests/test_sharded_multiprocess_gpu.py."""
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
    def __init__(self, torch, rank, world, chunk=8, cap=2):
        self.t, self.rank, self.world, self.chunk = torch, rank, world, chunk
        self.seg = sharded.halo_segment_bytes(cap)
        self.counts_local = torch.zeros(sharded.HALO_OBJ, dtype=torch.int32)
        self.counts_all = torch.full((world * sharded.HALO_OBJ,), -1, dtype=torch.int32)
        self.halo_send = torch.zeros(world * self.seg, dtype=torch.uint8)
        self.halo_recv = torch.zeros(world * self.seg, dtype=torch.uint8)
        self.ck_part = torch.zeros(world * chunk, dtype=torch.float32)
        self.ck_stage = torch.zeros(world * chunk, dtype=torch.float32)
        self.ck_chunk = torch.zeros(chunk, dtype=torch.float32)
        self.ck_full = torch.zeros(world * chunk, dtype=torch.float32)
        self.ck_all = torch.zeros(world * world * chunk, dtype=torch.float32)
        self.log = []

    def start(self, depth, cloud, pos, q, moves=None, remove_tracks=None, **kw):
        self.log.append("start")
        self.counts_local[:] = 0
        if moves is not None and len(moves):
            self.counts_local[:len(moves)] = self.t.arange(len(moves), dtype=self.t.int32) + 10 * (self.rank + 1)

    def moves(self):
        self.log.append("moves")
        for d in range(self.world):   # segment d: "from rank to d"
            self.halo_send[d * self.seg:(d + 1) * self.seg] = 16 * self.rank + d

    def predict(self):
        self.log.append("predict")
        # partial sum of shard `rank` for pixel p: (rank + 1) * 1000 + p
        self.ck_part[:] = 1000.0 * (self.rank + 1) + self.t.arange(self.world * self.chunk, dtype=self.t.float32)

    def ck_reduce(self):
        self.log.append("reduce")
        st = self.ck_stage.reshape(self.world, self.chunk)
        acc = self.t.zeros(self.chunk)
        for s in range(self.world):   # slab order
            acc = acc + st[s]
        self.ck_chunk[:] = acc

    def finish(self):
        self.log.append("finish")
        self.final = self.ck_full.clone()

    def finish_images(self):
        self.log.append("finish_images")
        img = self.ck_all.reshape(self.world, self.world * self.chunk)
        acc = self.t.zeros(self.world * self.chunk)
        for s in range(self.world):   # slab order
            acc = acc + img[s]
        self.final = acc


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
        assert eng.log == ["start", "moves", "predict", "reduce", "finish"]
        assert int(eng.counts_all[0]) == -1 and int(eng.halo_recv.sum()) == 0      # untouched
        # part s of the stage buffer = shard s's sums for MY chunk of the pixels
        st = eng.ck_stage.reshape(world, eng.chunk)
        mine = torch.arange(rank * eng.chunk, (rank + 1) * eng.chunk, dtype=torch.float32)
        for s in range(world):
            assert torch.equal(st[s], 1000.0 * (s + 1) + mine)
        # every shard ends up with the whole summed image: sum_s (1000 (s + 1) + p)
        p = torch.arange(world * eng.chunk, dtype=torch.float32)
        assert torch.equal(eng.final, 1000.0 * world * (world + 1) / 2 + world * p)
        # frame 2: two moving objects -> all four exchanges
        drv.update(None, None, None, None, moves=[1, 2])
        for r in range(world):
            row = eng.counts_all[r * sharded.HALO_OBJ:(r + 1) * sharded.HALO_OBJ]
            assert int(row[0]) == 10 * (r + 1) and int(row[1]) == 10 * (r + 1) + 1 and int(row[2]) == 0
            # segment r of the receive buffer: from shard r, addressed to me
            assert torch.all(eng.halo_recv[r * eng.seg:(r + 1) * eng.seg] == 16 * r + rank)
        # the one-collective variant of the ck exchange: image s of the gathered buffer = shard s's whole partial image, and
        # the slab-ordered sum comes out the same
        eng2 = FakeEngine(torch, rank, world)
        sharded.ShardedDriver(eng2, rank, world, dist, ck_exchange="allgather").update(None, None, None, None, moves=[])
        assert eng2.log == ["start", "moves", "predict", "finish_images"]
        img = eng2.ck_all.reshape(world, world * eng2.chunk)
        for s in range(world):
            assert torch.equal(img[s], 1000.0 * (s + 1) + p)
        assert torch.equal(eng2.final, eng.final)
        # rendezvous of the RCCL id: everybody gets rank 0's bytes
        from semantic_dsp_map_amd import binding
        binding.comm_unique_id = lambda: bytes([(7 * i + 3) % 256 for i in range(128)]) if rank == 0 else b"\0" * 128
        got = sharded.broadcast_unique_id(dist, rank)
        assert got == bytes([(7 * i + 3) % 256 for i in range(128)])
        results[rank] = "ok"
    except Exception as e:  # noqa: BLE001
        import traceback
        results[rank] = "FAILED: %r %s" % (e, traceback.format_exc())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_driver_processes_gloo(world):
    import torch.multiprocessing as mp
    port = free_port()
    mgr = mp.get_context("spawn").Manager()
    results = mgr.dict()
    mp.spawn(worker, args=(world, port, results), nprocs=world, join=True)
    assert dict(results) == {r: "ok" for r in range(world)}, dict(results)


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
