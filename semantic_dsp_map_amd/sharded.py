"""Z-slab sharded frame driver: one process per GPU, torch.distributed as plumbing.

The ring buffer is split by ring-z slab; a ring shift moves no data, so slab ownership never changes.
The only cross-shard dependency of a static-scene frame is pass 1 of the SMC-PHD weight update: every
shard computes the partial ck image of its own visible particles, the images are all-gathered (RCCL over
xGMI with the "nccl" backend; 4*H*W bytes per shard, 1.86 MB at 1242x375) and each shard sums them in
slab order, which keeps the result identical on every rank and reproducible by the oracle (ck_slabs).

The driver is engine-agnostic so that its collective plumbing can be exercised on CPU with gloo
(tests/test_sharded_gloo.py); the product engine is HipEngine.
"""
import numpy as np


HALO_OBJ = 64            # SDM_HALO_OBJ
HALO_RECORD_BYTES = 36   # SDM_HALO_RECORD_BYTES
HALO_HEADER_BYTES = 16   # SDM_HALO_HEADER_BYTES


class HipEngine:
    """libsdm_hip shard on one GPU; tensors are torch CUDA tensors, work runs on torch's current stream."""

    def __init__(self, cfg, params, rank, world, device, noise_table=None, max_visible=0, halo_cap=16384):
        import torch
        from . import binding
        self.torch = torch
        self.map = binding.SdmMap(cfg, params, noise_table, device=device, shard_rank=rank, shard_count=world,
                                  max_visible=max_visible)
        self.device = torch.device("cuda", device)
        self.hw = cfg["width"] * cfg["height"]
        self.rank, self.world = rank, world
        self.gathered = torch.zeros(world * self.hw, dtype=torch.float32, device=self.device)
        self.part = torch.zeros(self.hw, dtype=torch.float32, device=self.device)
        self.map.set_ck_buffer(self.part.data_ptr())
        # move exchanges: per-object member counts, and the export buffer of slab-crossing copies
        self.counts_local = torch.zeros(HALO_OBJ, dtype=torch.int32, device=self.device)
        self.counts_all = torch.zeros(world * HALO_OBJ, dtype=torch.int32, device=self.device)
        nbytes = HALO_HEADER_BYTES + halo_cap * HALO_RECORD_BYTES
        self.halo_send = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        self.halo_recv = torch.zeros(world * nbytes, dtype=torch.uint8, device=self.device)
        if world > 1:
            self.map.set_halo_buffers(self.counts_local.data_ptr(), self.counts_all.data_ptr(), self.halo_send.data_ptr(),
                                      self.halo_recv.data_ptr(), halo_cap)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if stream:
            self.map.set_stream(stream)

    # whole frame without exchanges (single shard)
    def update(self, depth_ptr, cloud_ptr, pos, q, moves=None, remove_tracks=None, on_device=True):
        self.map.update(depth_ptr, cloud_ptr, pos, q, moves, remove_tracks, on_device=on_device)

    # the four steps of a sharded frame
    def start(self, depth_ptr, cloud_ptr, pos, q, moves=None, remove_tracks=None, on_device=True):
        self.map.frame_start(depth_ptr, cloud_ptr, pos, q, moves, remove_tracks, on_device=on_device)

    def moves(self):
        self.map.frame_moves()

    def predict(self):
        self.map.frame_predict()
        return self.part

    def finish(self, gathered, n_parts):
        self.map.update_finish(gathered.data_ptr() if n_parts > 1 else None, n_parts)

    def synchronize(self):
        self.map.synchronize()


class ShardedDriver:
    """One frame on every shard: start -> [counts] -> moves -> [exports] -> predict -> [ck images] -> finish,
    where [x] is an all-gather over the shards (skipped when the frame moves no object)."""

    def __init__(self, engine, rank, world, dist=None):
        self.engine, self.rank, self.world, self.dist = engine, rank, world, dist
        if world > 1 and dist is None:
            raise ValueError("world > 1 needs a torch.distributed module")

    def update(self, depth, cloud, pos, q, moves=None, remove_tracks=None, **kw):
        e = self.engine
        if self.world == 1:
            return e.update(depth, cloud, pos, q, moves, remove_tracks, **kw)
        has_moves = moves is not None and len(moves) > 0   # replicated input: the same on every rank
        e.start(depth, cloud, pos, q, moves, remove_tracks, **kw)
        if has_moves:
            self.dist.all_gather_into_tensor(e.counts_all, e.counts_local)
        e.moves()
        if has_moves:
            self.dist.all_gather_into_tensor(e.halo_recv, e.halo_send)
        part = e.predict()
        # all_gather_into_tensor: rank r's image lands at [r*HW, (r+1)*HW) = slab order
        self.dist.all_gather_into_tensor(e.gathered, part)
        e.finish(e.gathered, self.world)


def weak_scaled_config(base_cfg, world):
    """Grid for `world` GPUs with a constant 2^(x_n+y_n+z_n) voxels per GPU: z grows first, then x, then y
    (C3 256^3 at 1 GPU ... C5 512^3 at 8 GPUs, BASELINE.json configs)."""
    cfg = dict(base_cfg)
    extra = int(np.log2(world))
    assert (1 << extra) == world, "world size must be a power of two"
    for i in range(extra):
        cfg[("z_n", "x_n", "y_n")[i % 3]] += 1
    return cfg
