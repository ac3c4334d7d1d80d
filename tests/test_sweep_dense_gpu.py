"""The non-incremental occupancy sweep (k_occupancy_scan + k_occupancy_dense) on a random, densely populated state.

The clips of the other parity tests populate surfaces: most 64-voxel chunks hold nothing or a handful of voxels, so the
sweep's dense path - chunks with >= 16 voxels to evaluate, records fetched cooperatively through LDS - and the rarer
rules of the vote (clamp of weights above 1, cull of updated particles below the initial weight, guessed births,
stale slots behind a re-stamped slab, ties between tracks) see little of it.  Here every slot of every voxel is drawn
at random, with stretches of empty and of sparsely filled chunks in between, the state is loaded into both sides
(which makes the next sweep a non-incremental one) and one frame is run: state and every voxel result must agree bit
for bit."""
import numpy as np
import pytest

from semantic_dsp_map_amd import binding, synth
from tests import parity_utils as pu

pytestmark = pytest.mark.gpu


def random_state(cfg, seed):
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
    kind = rng.choice(3, size=(V + 63) // 64, p=[0.25, 0.25, 0.5])[vox >> 6]
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


@pytest.mark.parametrize("p_n,seed", [(3, 1), (3, 2), (2, 3), (4, 4), (1, 5)])
def test_non_incremental_sweep_on_a_dense_random_state(p_n, seed):
    cfg = dict(synth.CONFIGS["T0"], p_n=p_n)
    params = synth.PARAMS["vkitti2"]
    _, _, frames = synth.make_frames("T0", 2, "vkitti2", n_dynamic=0)
    o, g = pu.make_pair(cfg, params, synth.noise_table())
    S = 1 << p_n
    st = random_state(cfg, seed)
    ring = dict(o.ring_state(), global_time_stamp=3)
    sx, sy, sz = (a.copy() for a in o.stamps())
    sx[3:6] = 2        # slabs re-stamped at frame 2: slots with an older stamp in them are stale
    sy[10] = 3
    sz[20:22] = 2
    for m in (o, g):
        m.load_state(st)
        m.set_stamps(sx, sy, sz)
        m.set_ring_state(ring)
    for depth, cloud, pos, q, moves in frames:   # frame 1: the non-incremental sweep; frame 2: an incremental one on top
        o.update(depth, cloud, pos, q, moves)
        g.update(depth, cloud, pos, q, moves, sync=True)
        rep = pu.compare_maps(o, g, S, check_results=True)
        assert not rep, "\n".join(rep)
    vo = o.voxels()
    # the state does exercise the rules: occupied, guessed-occupied and empty results all occur
    assert (vo["occ"] == 1).any() and (vo["occ"] == 2).any() and (vo["occ"] == 0).any() and (vo["occ"] < 0).any()
    g.close()
