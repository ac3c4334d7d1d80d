"""Z-slab sharding: one process per GPU.

The ring buffer is split by ring-z slab; a ring shift moves no data, so slab ownership never changes.  A frame
needs the other shards three times:
  1. per-object member counts (SDM_HALO_OBJ int32 per shard), all-gather   -> global particle ranks for the noise cursor
  2. slab-crossing copies of moved particles (36-byte records), all-to-all: every copy goes to the shard that owns its
     target voxel and to nobody else (one export segment per destination)
  3. partial ck images of pass 1 of the weight update, chunk-owner exchange: shard r owns 1/G of the pixels, receives
     every shard's partial sums for them (all-to-all), adds them in slab order, and the summed chunks are all-gathered:
     2 (G-1)/G images received per shard instead of G-1
On GPUs the library issues them itself with RCCL over xGMI on the map's streams (sdm_comm_init / sdm_update_sharded);
this module only does the rendezvous: rank 0 draws the RCCL id and it is broadcast over torch.distributed's gloo backend
(CPU only - torch's bundled HIP runtime is never initialised, the process has exactly one HIP runtime, the system one
libsdm_hip links against).

ShardedDriver is the same frame protocol over a generic engine and a torch.distributed-like module; it is what the
multi-process tests drive: with a CPU stand-in engine (tests/test_sharded_gloo.py) and with real libsdm_hip shards that
share one GPU (GlooShardEngine, tests/test_sharded_multiprocess_gpu.py).
"""
import numpy as np

HALO_OBJ = 64            # SDM_HALO_OBJ
HALO_RECORD_BYTES = 36   # SDM_HALO_RECORD_BYTES
HALO_HEADER_BYTES = 16   # SDM_HALO_HEADER_BYTES
HALO_DEFAULT_CAP = 4096  # SDM_HALO_DEFAULT_CAP: records per destination shard


def weak_scaled_config(base_cfg, world):
    """Grid for `world` GPUs with a constant 2^(x_n+y_n+z_n) voxels per GPU: z grows first, then x, then y
    (C3 256^3 at 1 GPU ... C5 512^3 at 8 GPUs, BASELINE.json configs)."""
    cfg = dict(base_cfg)
    extra = int(round(np.log2(world)))
    assert (1 << extra) == world, "world size must be a power of two"
    for i in range(extra):
        cfg[("z_n", "x_n", "y_n")[i % 3]] += 1
    return cfg


class stdout_to_stderr:
    """`with stdout_to_stderr():` - file descriptor 1 points at stderr inside the block, C-level buffers flushed on the
    way out.  RCCL (version banner at communicator creation) and gloo (connection summary at init_process_group) print
    on the process's stdout from C++; bench.py's stdout carries ONE JSON line and nothing else."""

    def __enter__(self):
        import os
        import sys
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import os
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def broadcast_unique_id(dist, rank):
    """rank 0 draws the RCCL id (sdm_comm_unique_id), everybody receives it (gloo broadcast of 128 bytes)."""
    import torch
    from . import binding
    t = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        t = torch.frombuffer(bytearray(binding.comm_unique_id()), dtype=torch.uint8).clone()
    dist.broadcast(t, src=0)
    return bytes(t.numpy().tobytes())


def gather_ipc_handles(dist, handle, world):
    """the 64-byte arena handles of all shards, in shard order (gloo all-gather)"""
    import torch
    mine = torch.frombuffer(bytearray(handle), dtype=torch.uint8).clone()
    allh = torch.zeros(64 * world, dtype=torch.uint8)
    dist.all_gather_into_tensor(allh, mine)
    return bytes(allh.numpy().tobytes())


class NativeShardedMap:
    """A libsdm_hip shard whose exchanges run inside the library: on RCCL (exchange="rccl", the default) or through the
    peers' arenas mapped with hipIpc (exchange="ipc": one small kernel per exchange, no collective launches)."""

    def __init__(self, cfg, params, rank, world, device, dist=None, noise_table=None, halo_cap=0, max_visible=0, force_comm=False,
                 exchange="rccl"):
        from . import binding
        self.rank, self.world = rank, world
        self.sharded = world > 1 or force_comm   # force_comm: the sharded frame with one rank (rehearsal)
        self.exchange = exchange
        self.map = binding.SdmMap(cfg, params, noise_table, device=device, shard_rank=rank, shard_count=world,
                                  max_visible=max_visible)
        if self.sharded:
            if dist is None and world > 1:
                raise ValueError("world > 1 needs a torch.distributed module for the rendezvous")
            if exchange == "ipc":
                handle = self.map.ipc_create(halo_cap)
                self.map.ipc_connect(gather_ipc_handles(dist, handle, world) if world > 1 else handle)
                if dist is not None:
                    dist.barrier()  # every shard has mapped every arena before the first frame writes into one
            else:
                # (a communicator of ONE rank needs no rendezvous: bench.py's `sharded_one_rank` leg runs without torch)
                uid = broadcast_unique_id(dist, rank) if dist is not None else binding.comm_unique_id()
                with stdout_to_stderr():  # RCCL prints a version banner on stdout at communicator creation
                    self.map.comm_init(uid, halo_cap)

    def update(self, depth, cloud, pos, q, moves=None, remove_tracks=None, on_device=True):
        if not self.sharded:
            self.map.update(depth, cloud, pos, q, moves, remove_tracks, on_device=on_device)
        else:
            self.map.update_sharded(depth, cloud, pos, q, moves, remove_tracks, on_device=on_device)

    def synchronize(self):
        self.map.synchronize()


def halo_segment_bytes(cap):
    return HALO_HEADER_BYTES + cap * HALO_RECORD_BYTES


class GlooShardEngine:
    """A real libsdm_hip shard driven through the split entry points (sdm_frame_start / _moves / _predict / sdm_ck_reduce
    / sdm_update_finish) with the exchanges carried by a CPU backend: the exchange buffers live on the device
    (sdm_set_halo_buffers, sdm_set_ck_buffer), `pull` copies this shard's part to a host tensor before a collective
    and `push` copies the received tensor back.  This is how several processes that share ONE GPU run the sharded engine
    (RCCL refuses two ranks on one device); on one GPU per process NativeShardedMap does the same with RCCL and no
    host copies."""

    def __init__(self, cfg, params, rank, world, device=0, noise_table=None, halo_cap=1024, max_visible=0):
        import torch
        from . import binding
        self.rank, self.world = rank, world
        self.map = m = binding.SdmMap(cfg, params, noise_table, device=device, shard_rank=rank, shard_count=world,
                                      max_visible=max_visible)
        self.hw = cfg["width"] * cfg["height"]
        self.chunk = m.ck_chunk_elems()
        self.seg = halo_segment_bytes(halo_cap)
        hb, ck = world * self.seg, world * self.chunk
        self.d = {"counts_local": m.device_put(np.zeros(HALO_OBJ, np.int32)),
                  "counts_all": m.device_put(np.zeros(world * HALO_OBJ, np.int32)),
                  "halo_send": m.device_put(np.zeros(hb, np.uint8)),
                  "halo_recv": m.device_put(np.zeros(hb, np.uint8)),
                  "ck_part": m.device_put(np.zeros(ck, np.float32)),
                  "ck_stage": m.device_put(np.zeros(ck, np.float32)),
                  "ck_full": m.device_put(np.zeros(ck, np.float32)),
                  "ck_all": m.device_put(np.zeros(world * self.hw, np.float32))}
        m.set_ck_buffer(self.d["ck_part"])
        m.set_halo_buffers(self.d["counts_local"], self.d["counts_all"], self.d["halo_send"], self.d["halo_recv"], halo_cap)
        self.counts_local = torch.zeros(HALO_OBJ, dtype=torch.int32)
        self.counts_all = torch.zeros(world * HALO_OBJ, dtype=torch.int32)
        self.halo_send = torch.zeros(hb, dtype=torch.uint8)
        self.halo_recv = torch.zeros(hb, dtype=torch.uint8)
        self.ck_part = torch.zeros(ck, dtype=torch.float32)
        self.ck_stage = torch.zeros(ck, dtype=torch.float32)
        self.ck_chunk = torch.zeros(self.chunk, dtype=torch.float32)
        self.ck_full = torch.zeros(ck, dtype=torch.float32)
        self.ck_all = torch.zeros(world * ck, dtype=torch.float32)   # ck_exchange "allgather": every shard's padded partial image
        # bytes RECEIVED from other shards per exchange, summed over the frames; halo_records = records this shard exported
        self.bytes_exchanged = {"counts": 0, "halo": 0, "halo_records": 0, "ck_alltoall": 0, "ck_allgather": 0, "ck_images": 0, "frames": 0}

    # exchange hooks of ShardedDriver: device source -> host tensor before a collective, host tensor -> device after it
    def pull(self, name):
        src, dst = {"counts": ("counts_local", self.counts_local), "halo": ("halo_send", self.halo_send),
                    "ck_part": ("ck_part", self.ck_part), "ck_image": ("ck_part", self.ck_part),
                    "ck_chunk": ("ck_full", self.ck_chunk)}[name]
        host = dst.numpy()
        off = self.rank * self.chunk * 4 if name == "ck_chunk" else 0
        host.view(np.uint8)[:] = self.map.device_download(self.d[src] + off, host.nbytes)
        w1 = self.world - 1
        if name == "counts":
            self.bytes_exchanged["counts"] += host.nbytes * w1
        elif name == "halo":
            self.bytes_exchanged["halo"] += self.seg * w1
            heads = host.view(np.uint8).reshape(self.world, self.seg)[:, :4].copy().view(np.uint32)
            self.bytes_exchanged["halo_records"] += int(heads.sum())
        elif name == "ck_part":
            self.bytes_exchanged["ck_alltoall"] += self.chunk * 4 * w1
        elif name == "ck_image":
            self.bytes_exchanged["ck_images"] += self.chunk * self.world * 4 * w1
        else:
            self.bytes_exchanged["ck_allgather"] += self.chunk * 4 * w1

    def push(self, name):
        if name == "ck_all":  # the gathered images are padded to whole chunks: sdm_update_finish takes them H*W floats apart
            img = self.ck_all.numpy().reshape(self.world, self.world * self.chunk)[:, :self.hw]
            self.map.device_upload(self.d["ck_all"], np.ascontiguousarray(img))
            return
        src, dst = {"counts": (self.counts_all, "counts_all"), "halo": (self.halo_recv, "halo_recv"),
                    "ck_stage": (self.ck_stage, "ck_stage"), "ck_full": (self.ck_full, "ck_full")}[name]
        self.map.device_upload(self.d[dst], src.numpy())

    def start(self, depth, cloud, pos, q, moves=None, remove_tracks=None, **kw):
        self.bytes_exchanged["frames"] += 1
        self.map.frame_start(depth, cloud, pos, q, moves, remove_tracks, **kw)

    def moves(self):
        self.map.frame_moves()

    def moves_pending(self):
        return self.map.frame_moves_pending()

    def predict(self):
        self.map.frame_predict()

    def ck_reduce(self):
        self.map.ck_reduce(self.d["ck_stage"], self.d["ck_full"])

    def finish(self):
        self.map.update_finish(self.d["ck_full"], 1)

    def finish_images(self):
        """ck_exchange "allgather": the shards' whole partial images, added in slab order by the library"""
        self.map.update_finish(self.d["ck_all"], self.world)

    def close(self):
        self.map.close()


class ShardedDriver:
    """The frame protocol over a generic engine:
        start -> [counts: all-gather] -> moves (-> [counts] -> moves, per further batch of a long object list)
              -> [export segments: all-to-all] -> predict
              -> [partial ck chunks: all-to-all] -> ck_reduce -> [summed chunks: all-gather] -> finish
    where [x] is a collective over the shards (the first two only when objects move).  Engine attributes (torch tensors
    on the engine's device): counts_local / counts_all, halo_send / halo_recv (world segments each), ck_part / ck_stage /
    ck_full (world chunks each) and ck_chunk (this shard's summed chunk); methods start, moves, predict, ck_reduce, finish;
    optional pull(name) / push(name) around every collective for engines whose buffers have to be staged (GlooShardEngine)."""

    def __init__(self, engine, rank, world, dist=None, ck_exchange="chunks"):
        """ck_exchange: "chunks" = chunk-owner reduction (two collectives, 2 (G-1)/G images received), "allgather" = ONE
        all-gather of the whole partial images, every shard adds them itself (sdm_comm_set_options, ck_exchange 1)"""
        self.engine, self.rank, self.world, self.dist = engine, rank, world, dist
        self.ck_exchange = ck_exchange
        if world > 1 and dist is None:
            raise ValueError("world > 1 needs a torch.distributed module")

    def update(self, depth, cloud, pos, q, moves=None, remove_tracks=None, **kw):
        e, d = self.engine, self.dist
        has_moves = moves is not None and len(moves) > 0   # replicated input: the same on every rank
        pull, push = getattr(e, "pull", lambda name: None), getattr(e, "push", lambda name: None)
        multi = self.world > 1
        e.start(depth, cloud, pos, q, moves, remove_tracks, **kw)
        if has_moves and multi:
            pull("counts")
            d.all_gather_into_tensor(e.counts_all, e.counts_local)
            push("counts")
        e.moves()
        # a list of more than SDM_MAX_MOVES objects: batch by batch, every batch's counts gathered before it is applied
        pending = getattr(e, "moves_pending", lambda: False)
        while has_moves and pending():
            if multi:
                pull("counts")
                d.all_gather_into_tensor(e.counts_all, e.counts_local)
                push("counts")
            e.moves()
        if has_moves and multi:
            pull("halo")
            d.all_to_all_single(e.halo_recv, e.halo_send)         # segment s of recv = segment `rank` of shard s's send
            push("halo")
        e.predict()
        if self.ck_exchange == "allgather":
            pull("ck_image")
            if multi:
                d.all_gather_into_tensor(e.ck_all, e.ck_part)   # image s = shard s's partial sums for every pixel
            else:
                e.ck_all[:e.ck_part.numel()] = e.ck_part
            push("ck_all")
            e.finish_images()
            return
        pull("ck_part")
        if multi:
            d.all_to_all_single(e.ck_stage, e.ck_part)            # part s of stage = shard s's sums for my pixels
        else:
            e.ck_stage.copy_(e.ck_part)
        push("ck_stage")
        e.ck_reduce()
        pull("ck_chunk")
        if multi:
            d.all_gather_into_tensor(e.ck_full, e.ck_chunk)
        else:
            e.ck_full[:e.ck_chunk.numel()] = e.ck_chunk
        push("ck_full")
        e.finish()
