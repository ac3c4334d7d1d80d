"""Z-slab sharding: one process per GPU.

The ring buffer is split by ring-z slab; a ring shift moves no data, so slab ownership never changes.  A frame
needs the other shards three times:
  1. per-object member counts        (SDM_HALO_OBJ int32 per shard)  -> global particle ranks for the noise cursor
  2. slab-crossing copies of moved particles (36-byte records)       -> imported by the slab that owns the target voxel
  3. partial ck images of pass 1 of the weight update (4*H*W bytes)  -> summed in slab order on every shard
Each is an all-gather.  On GPUs the library issues them itself with RCCL over xGMI on the map's stream
(sdm_comm_init / sdm_update_sharded); this module only does the rendezvous: rank 0 draws the RCCL id and it is
broadcast over torch.distributed's gloo backend (CPU only — torch's bundled HIP runtime is never initialised, the
process has exactly one HIP runtime, the system one libsdm_hip links against).

ShardedDriver is the same frame protocol over a generic engine and a torch.distributed-like module; it is what the
2-process gloo test drives with a CPU stand-in engine (tests/test_sharded_gloo.py).
"""
import numpy as np

HALO_OBJ = 64            # SDM_HALO_OBJ
HALO_RECORD_BYTES = 36   # SDM_HALO_RECORD_BYTES
HALO_HEADER_BYTES = 16   # SDM_HALO_HEADER_BYTES


def weak_scaled_config(base_cfg, world):
    """Grid for `world` GPUs with a constant 2^(x_n+y_n+z_n) voxels per GPU: z grows first, then x, then y
    (C3 256^3 at 1 GPU ... C5 512^3 at 8 GPUs, BASELINE.json configs)."""
    cfg = dict(base_cfg)
    extra = int(round(np.log2(world)))
    assert (1 << extra) == world, "world size must be a power of two"
    for i in range(extra):
        cfg[("z_n", "x_n", "y_n")[i % 3]] += 1
    return cfg


def broadcast_unique_id(dist, rank):
    """rank 0 draws the RCCL id (sdm_comm_unique_id), everybody receives it (gloo broadcast of 128 bytes)."""
    import torch
    from . import binding
    t = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        t = torch.frombuffer(bytearray(binding.comm_unique_id()), dtype=torch.uint8).clone()
    dist.broadcast(t, src=0)
    return bytes(t.numpy().tobytes())


class NativeShardedMap:
    """A libsdm_hip shard whose exchanges run inside the library on RCCL."""

    def __init__(self, cfg, params, rank, world, device, dist=None, noise_table=None, halo_cap=16384, max_visible=0):
        from . import binding
        self.rank, self.world = rank, world
        self.map = binding.SdmMap(cfg, params, noise_table, device=device, shard_rank=rank, shard_count=world,
                                  max_visible=max_visible)
        if world > 1:
            if dist is None:
                raise ValueError("world > 1 needs a torch.distributed module for the rendezvous")
            uid = broadcast_unique_id(dist, rank)
            # RCCL prints a version banner on stdout at communicator creation; keep stdout clean for the caller's JSON
            import os
            import sys
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                self.map.comm_init(uid, halo_cap)
            finally:
                os.dup2(saved, 1)
                os.close(saved)

    def update(self, depth, cloud, pos, q, moves=None, remove_tracks=None, on_device=True):
        if self.world == 1:
            self.map.update(depth, cloud, pos, q, moves, remove_tracks, on_device=on_device)
        else:
            self.map.update_sharded(depth, cloud, pos, q, moves, remove_tracks, on_device=on_device)

    def synchronize(self):
        self.map.synchronize()


class GlooShardEngine:
    """A real libsdm_hip shard driven through the split entry points (sdm_frame_start / _moves / _predict /
    sdm_update_finish) with the three exchanges carried by a CPU backend: the exchange buffers live on the device
    (sdm_set_halo_buffers, sdm_set_ck_buffer), `pull` copies this shard's part to a host tensor before an all-gather
    and `push` copies the gathered tensor back.  This is how two processes that share ONE GPU run the sharded engine
    (RCCL refuses two ranks on one device); on one GPU per process NativeShardedMap does the same with RCCL and no
    host copies."""

    def __init__(self, cfg, params, rank, world, device=0, noise_table=None, halo_cap=16384, max_visible=0):
        import torch
        from . import binding
        self.rank, self.world = rank, world
        self.map = m = binding.SdmMap(cfg, params, noise_table, device=device, shard_rank=rank, shard_count=world,
                                      max_visible=max_visible)
        self.hw = cfg["width"] * cfg["height"]
        self.hb = HALO_HEADER_BYTES + halo_cap * HALO_RECORD_BYTES
        self.d = {"counts_local": m.device_put(np.zeros(HALO_OBJ, np.int32)),
                  "counts_all": m.device_put(np.zeros(world * HALO_OBJ, np.int32)),
                  "halo_send": m.device_put(np.zeros(self.hb, np.uint8)),
                  "halo_recv": m.device_put(np.zeros(world * self.hb, np.uint8)),
                  "part": m.device_alloc(self.hw * 4),
                  "gathered": m.device_alloc(world * self.hw * 4)}
        m.set_ck_buffer(self.d["part"])
        m.set_halo_buffers(self.d["counts_local"], self.d["counts_all"], self.d["halo_send"], self.d["halo_recv"], halo_cap)
        self.counts_local = torch.zeros(HALO_OBJ, dtype=torch.int32)
        self.counts_all = torch.zeros(world * HALO_OBJ, dtype=torch.int32)
        self.halo_send = torch.zeros(self.hb, dtype=torch.uint8)
        self.halo_recv = torch.zeros(world * self.hb, dtype=torch.uint8)
        self.part = torch.zeros(self.hw, dtype=torch.float32)
        self.gathered = torch.zeros(world * self.hw, dtype=torch.float32)
        self.bytes_exchanged = {"counts": 0, "halo": 0, "halo_records": 0, "ck": 0}

    # exchange hooks of ShardedDriver: name -> (device source, host tensor) / (host tensor, device destination)
    def pull(self, name):
        src, dst = {"counts": ("counts_local", self.counts_local), "halo": ("halo_send", self.halo_send),
                    "ck": ("part", self.part)}[name]
        host = dst.numpy()
        host.view(np.uint8)[:] = self.map.device_download(self.d[src], host.nbytes)
        self.bytes_exchanged[name] += host.nbytes * (self.world - 1)          # what an all-gather receives per rank
        if name == "halo":
            self.bytes_exchanged["halo_records"] += int(host[:4].copy().view(np.uint32)[0])

    def push(self, name):
        src, dst = {"counts": (self.counts_all, "counts_all"), "halo": (self.halo_recv, "halo_recv"),
                    "ck": (self.gathered, "gathered")}[name]
        self.map.device_upload(self.d[dst], src.numpy())

    def start(self, depth, cloud, pos, q, moves=None, remove_tracks=None, **kw):
        self.map.frame_start(depth, cloud, pos, q, moves, remove_tracks, **kw)

    def moves(self):
        self.map.frame_moves()

    def predict(self):
        self.map.frame_predict()
        return self.part

    def finish(self, gathered, n_parts):
        self.map.update_finish(self.d["gathered"], n_parts)

    def close(self):
        self.map.close()


class ShardedDriver:
    """The frame protocol over a generic engine: start -> [counts] -> moves -> [exports] -> predict -> [ck images]
    -> finish, where [x] is dist.all_gather_into_tensor over the shards (the first two only when objects move).
    Engine attributes: counts_local/counts_all, halo_send/halo_recv, part/gathered (torch tensors on the engine's
    device); methods start, moves, predict, finish; optional pull(name) / push(name) around every exchange for engines
    whose buffers have to be staged (GlooShardEngine)."""

    def __init__(self, engine, rank, world, dist=None):
        self.engine, self.rank, self.world, self.dist = engine, rank, world, dist
        if world > 1 and dist is None:
            raise ValueError("world > 1 needs a torch.distributed module")

    def update(self, depth, cloud, pos, q, moves=None, remove_tracks=None, **kw):
        e = self.engine
        has_moves = moves is not None and len(moves) > 0   # replicated input: the same on every rank
        pull, push = getattr(e, "pull", lambda name: None), getattr(e, "push", lambda name: None)
        e.start(depth, cloud, pos, q, moves, remove_tracks, **kw)
        if has_moves and self.world > 1:
            pull("counts")
            self.dist.all_gather_into_tensor(e.counts_all, e.counts_local)
            push("counts")
        e.moves()
        if has_moves and self.world > 1:
            pull("halo")
            self.dist.all_gather_into_tensor(e.halo_recv, e.halo_send)
            push("halo")
        part = e.predict()
        pull("ck")
        if self.world > 1:
            # rank r's image lands at [r*HW, (r+1)*HW): slab order
            self.dist.all_gather_into_tensor(e.gathered, part)
        else:
            e.gathered[:part.numel()] = part
        push("ck")
        e.finish(e.gathered, self.world)
