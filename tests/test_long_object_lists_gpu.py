"""Object lists longer than one block of kernel arguments holds: 200 moving objects and 300 removals in ONE frame.

The reference loops over whatever the object layer hands it (semantic_dsp_map.h:588-736) and takes every object's particles
out of the map before it re-inserts any, with one running noise cursor (mc_ring/operations.h:321-362).  The library's frame
block holds 48 motions and 128 removals; longer lists are worked off in batches inside the frame (member count + copy-out per
batch, ONE ordered re-insertion after the last, ranks running on from batch to batch).  Here a T0 map is seeded with
particles of 260 movable tracks in random, overlapping blobs - so that copies of different batches compete for the same
voxels and the same owner slots - then frames move 200 of them at once (rotations + translations) and remove up to 300
ids (born ones, moved ones, ids that never existed); both sides run free and must agree bit for bit after every frame."""
import numpy as np
import pytest

from semantic_dsp_map_amd import binding, synth
from tests import parity_utils as pu
from tests.test_fuzz_gpu import rot_y

pytestmark = pytest.mark.gpu

N_TRACKS = 260


def frame(rng, cfg, params, pos, yaw):
    """random depth; every pixel claimed by one of N_TRACKS movable tracks in small blobs"""
    W, H = cfg["width"], cfg["height"]
    depth = (0.8 + 7.0 * rng.random((H, W))).astype(np.float32)
    depth = (0.25 * (depth + np.roll(depth, 1, 0) + np.roll(depth, 1, 1) + np.roll(depth, (1, 1), (0, 1)))).astype(np.float32)
    jj, ii = np.meshgrid(np.arange(W), np.arange(H))
    xc = (jj - cfg["cx"]) / cfg["fx"] * depth
    yc = (ii - cfg["cy"]) / cfg["fy"] * depth
    pg = np.stack([xc, yc, depth], -1).reshape(-1, 3).astype(np.float64) @ rot_y(yaw).T + pos
    cloud = np.zeros(H * W, synth.LABELED_POINT)
    cloud["x"], cloud["y"], cloud["z"] = pg[:, 0], pg[:, 1], pg[:, 2]
    cloud["sigma"] = (params["depth_noise_zero_order"] + params["depth_noise_first_order"] * depth.reshape(-1)).astype(np.float32) \
        if params["if_consider_depth_noise"] else np.float32(0.1)
    # an 8 x 8 pixel checkerboard of track ids, shifted per frame: neighbouring blobs share voxels
    off = int(rng.integers(0, 8))
    track = (((ii + off) // 8) * ((W + 7) // 8 + 1) + (jj + off) // 8) % N_TRACKS + 1
    cloud["track_id"] = track.reshape(-1).astype(np.uint16)
    cloud["label_id"] = synth.LABEL_CAR
    cloud["is_valid"] = 1
    return depth, cloud


def moves_of(rng, ids):
    mv = np.zeros(len(ids), synth.OBJECT_MOVE)
    for k, trk in enumerate(ids):
        T = np.eye(4, dtype=np.float32)
        T[:3, :3] = rot_y(rng.normal(0, 0.04)).astype(np.float32)
        T[:3, 3] = rng.normal(0, 0.3, 3).astype(np.float32)
        mv[k]["track_id"], mv[k]["T"] = int(trk), T.reshape(-1)
    return mv


@pytest.mark.parametrize("params_name,seed", [("vkitti2", 1), ("noisy3", 2)])
def test_200_moving_objects_and_300_removals_in_one_frame(params_name, seed):
    cfg = synth.CONFIGS["T0"]
    params = synth.PARAMS[params_name]
    rng = np.random.default_rng(seed)
    o, g = pu.make_pair(cfg, params, synth.noise_table())
    S = 1 << cfg["p_n"]
    pos, yaw = np.zeros(3), 0.0
    n_moved_max = 0
    for t in range(10):
        pos = pos + rng.normal(0, 0.2, 3) * np.array([1.0, 0.1, 1.0])
        yaw += rng.normal(0, 0.03)
        depth, cloud = frame(rng, cfg, params, pos, yaw)
        if t < 2:            # populate
            mv, remove = moves_of(rng, []), None
        elif t % 3 == 2:     # 200 objects at once, in an order that is not the id order; a few ids twice never (the object layer moves an object once)
            ids = rng.permutation(np.arange(1, N_TRACKS + 1))[:200]
            mv, remove = moves_of(rng, ids), None
        elif t % 3 == 0:     # 300 removals: born ids, ids that never existed (up to 400)
            mv = moves_of(rng, rng.permutation(np.arange(1, N_TRACKS + 1))[:60])    # and a list of moves just over one batch
            remove = [int(x) for x in rng.permutation(np.arange(1, 401))[:300]]
        else:                # both long lists in one frame
            mv = moves_of(rng, rng.permutation(np.arange(1, N_TRACKS + 1))[:150])
            remove = [int(x) for x in rng.permutation(np.arange(1, 401))[:140]]
        q = synth.yaw_quat(yaw).astype(np.float32)
        p32 = pos.astype(np.float32)
        o.update(depth, cloud, p32, q, mv, remove)
        g.update(depth, cloud, p32, q, mv, remove, sync=True)
        rep = pu.compare_maps(o, g, S, tag="%s frame %d (%d moves, %d removals): " % (params_name, t, len(mv), len(remove or [])))
        assert not rep, "\n".join(rep)
        so, sg = o.stats(), g.stats()
        assert so["n_moved"] == sg["n_moved"], (t, so["n_moved"], sg["n_moved"])
        n_moved_max = max(n_moved_max, sg["n_moved"])
    assert n_moved_max > 2000, n_moved_max   # the long lists did move particles
    g.close()
