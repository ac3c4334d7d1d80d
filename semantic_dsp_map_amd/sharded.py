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


class HipEngine:
    """libsdm_hip shard on one GPU; tensors are torch CUDA tensors, work runs on torch's current stream."""

    def __init__(self, cfg, params, rank, world, device, noise_table=None, max_visible=0):
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
        self.map.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def begin(self, depth_ptr, cloud_ptr, pos, q, moves=None, remove_tracks=None, on_device=True):
        self.map.update_begin(depth_ptr, cloud_ptr, pos, q, moves, remove_tracks, on_device=on_device)
        return self.part

    def update(self, depth_ptr, cloud_ptr, pos, q, moves=None, remove_tracks=None, on_device=True):
        self.map.update(depth_ptr, cloud_ptr, pos, q, moves, remove_tracks, on_device=on_device)

    def finish(self, gathered, n_parts):
        self.map.update_finish(gathered.data_ptr() if n_parts > 1 else None, n_parts)

    def synchronize(self):
        self.map.synchronize()


class ShardedDriver:
    """Runs one frame on every shard: begin -> all-gather of the partial ck images -> finish."""

    def __init__(self, engine, rank, world, dist=None):
        self.engine, self.rank, self.world, self.dist = engine, rank, world, dist
        if world > 1 and dist is None:
            raise ValueError("world > 1 needs a torch.distributed module")

    def update(self, *frame, **kw):
        if self.world == 1 and hasattr(self.engine, "update"):
            return self.engine.update(*frame, **kw)   # no exchange: the fused single-GPU frame
        part = self.engine.begin(*frame, **kw)
        if self.world > 1:
            # all_gather_into_tensor: rank r's image lands at [r*HW, (r+1)*HW) = slab order
            self.dist.all_gather_into_tensor(self.engine.gathered, part)
        self.engine.finish(self.engine.gathered, self.world)


def weak_scaled_config(base_cfg, world):
    """Grid for `world` GPUs with a constant 2^(x_n+y_n+z_n) voxels per GPU: z grows first, then x, then y
    (C3 256^3 at 1 GPU ... C5 512^3 at 8 GPUs, BASELINE.json configs)."""
    cfg = dict(base_cfg)
    extra = int(np.log2(world))
    assert (1 << extra) == world, "world size must be a power of two"
    for i in range(extra):
        cfg[("z_n", "x_n", "y_n")[i % 3]] += 1
    return cfg
