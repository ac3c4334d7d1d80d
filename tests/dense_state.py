"""A random, densely populated map state for the tests of the non-incremental occupancy sweep: every slot of every voxel
drawn at random (with stretches of empty and of sparsely filled 64-voxel chunks), weights that trigger the clamp and
cull rules, guessed births, ties between tracks.  Plain numpy: used by the GPU parity test, by the golden fixture and by
its generator."""
import numpy as np

from semantic_dsp_map_amd import binding


def random_state(cfg, seed, run=1, kinds=(0.25, 0.25, 0.5)):
    """run: chunks of one kind come in aligned runs of this many (run = 8: whole 512-voxel groups - what one wave of the
    non-incremental sweep's kernels takes - are dense, sparse or empty, so that the sweep's group hints come into play);
    kinds: how often a run is empty / sparse / dense"""
    rng = np.random.default_rng(seed)
    NX, NY, NZ, S = 1 << cfg["x_n"], 1 << cfg["y_n"], 1 << cfg["z_n"], 1 << cfg["p_n"]
    V = NX * NY * NZ
    size = np.float32(cfg["voxel_size"])
    pmin = [-(N >> 1) * size for N in (NX, NY, NZ)]
    st = {k: np.zeros(V * S, dt) for k, dt in binding.STATE_FIELDS}
    st["owner"][:] = 0xFFFF
    vox = np.arange(V, dtype=np.int64)
    vx, vy, vz = vox & (NX - 1), (vox >> cfg["x_n"]) & (NY - 1), vox >> (cfg["x_n"] + cfg["y_n"])
    # per chunk of 64 voxels: 0 = empty, 1 = a few voxels hold something, 2 = every voxel does
    n_runs = ((V + 63) // 64 + run - 1) // run
    kind = rng.choice(3, size=n_runs, p=list(kinds))[(vox >> 6) // run]
    holds = (kind == 2) | ((kind == 1) & (rng.random(V) < 0.1))
    status = st["status"].reshape(V, S)
    status[:, 0] = 5                                              # TIMEPTC
    st["ts"].reshape(V, S)[:, 0] = rng.choice([0, 1, 2, 3, 3, 3], size=V)   # 0 = never observed
    weights = np.array([0.0, 0.0005, 0.001, 0.05, 0.3, 0.3, 0.7, 1.0, 1.2, 2.5], np.float32)   # ties, clamp and cull candidates
    for s in range(1, S):
        live = holds & (rng.random(V) < 0.8)
        idx = vox * S + s
        status[:, s] = np.where(live, rng.choice([1, 1, 1, 2, 3, 4], size=V), 0)   # UPDATED / REGULAR_BORN / GUESSED_BORN / COPIED
        st["w"][idx] = np.where(live, rng.choice(weights, size=V), 0).astype(np.float32)
        st["ts"][idx] = np.where(live, rng.choice([1, 2, 3], size=V), 0)
        st["track"][idx] = np.where(live, rng.choice([1, 2, 3, 65531, 65535], size=V), 0)
        st["label"][idx] = np.where(live, rng.integers(1, 20, size=V), 0)
        st["px"][idx] = (vx.astype(np.float32) + rng.random(V, np.float32)) * size + pmin[0]
        st["py"][idx] = (vy.astype(np.float32) + rng.random(V, np.float32)) * size + pmin[1]
        st["pz"][idx] = (vz.astype(np.float32) + rng.random(V, np.float32)) * size + pmin[2]
    return st


def stamps_for(o):
    """The initial stamp arrays of a map with a few slabs re-stamped at frame 2 and 3 (slots with an older stamp in them
    are stale) and the ring state at frame 3."""
    sx, sy, sz = (a.copy() for a in o.stamps())
    sx[3:6] = 2
    sy[10] = 3
    sz[20:22] = 2
    return (sx, sy, sz), dict(o.ring_state(), global_time_stamp=3)
