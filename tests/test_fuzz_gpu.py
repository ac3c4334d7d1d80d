"""Seeded random frames (not a scene): depth with holes, points that fall outside the map, pixels claimed by a handful
of movable track ids in random blobs, random rigid motions and removals of those tracks, an erratic camera.  Both
implementations run free for 120 frames; the inputs are built to hit the corners scenes rarely produce (objects that
overlap, tracks that are moved while empty, removals of tracks that were never born, re-used owner slots)."""
import numpy as np
import pytest

from semantic_dsp_map_amd import synth
from tests import parity_utils as pu

pytestmark = pytest.mark.gpu


def rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float64)


def random_frame(rng, cfg, params, t, pos, yaw):
    W, H = cfg["width"], cfg["height"]
    depth = (0.5 + 9.0 * rng.random((H, W))).astype(np.float32)
    # smooth it a little so that neighbouring pixels share voxels, then punch holes
    depth = (0.25 * (depth + np.roll(depth, 1, 0) + np.roll(depth, 1, 1) + np.roll(depth, (1, 1), (0, 1)))).astype(np.float32)
    depth[rng.random((H, W)) < 0.05] = np.nan
    depth[rng.random((H, W)) < 0.03] = np.float32(cfg["depth_max"] + 3.0)
    jj, ii = np.meshgrid(np.arange(W), np.arange(H))
    xc = (jj - cfg["cx"]) / cfg["fx"] * depth
    yc = (ii - cfg["cy"]) / cfg["fy"] * depth
    pc = np.stack([xc, yc, depth], -1).reshape(-1, 3).astype(np.float64)
    pg = pc @ rot_y(yaw).T + pos
    cloud = np.zeros(H * W, synth.LABELED_POINT)
    valid = np.isfinite(depth).reshape(-1) & (depth.reshape(-1) >= cfg["depth_min"]) & (depth.reshape(-1) <= cfg["depth_max"])
    cloud["x"], cloud["y"], cloud["z"] = pg[:, 0], pg[:, 1], pg[:, 2]
    if params["if_consider_depth_noise"]:
        cloud["sigma"] = (params["depth_noise_zero_order"]
                          + params["depth_noise_first_order"] * np.nan_to_num(depth.reshape(-1))).astype(np.float32)
    else:
        cloud["sigma"] = np.float32(0.1)
    track = np.full(H * W, synth.TRACK_BUILDING, np.uint16)
    label = np.full(H * W, synth.LABEL_BUILDING, np.uint8)
    for trk in range(1, 6):                      # blobs of movable tracks, overlapping on purpose
        if rng.random() < 0.7:
            ci, cj, rad = rng.integers(0, H), rng.integers(0, W), rng.integers(4, 18)
            blob = ((ii - ci) ** 2 + (jj - cj) ** 2 <= rad * rad).reshape(-1)
            track[blob] = trk
            label[blob] = synth.LABEL_CAR
    cloud["track_id"], cloud["label_id"], cloud["is_valid"] = track, label, valid
    cloud["x"][~valid] = 0
    cloud["y"][~valid] = 0
    cloud["z"][~valid] = 0
    moves = []
    for trk in range(1, 7):                      # 6 is never born: moving an object that does not exist
        if rng.random() < 0.5:
            T = np.eye(4, dtype=np.float32)
            T[:3, :3] = rot_y(rng.normal(0, 0.05)).astype(np.float32)
            T[:3, 3] = rng.normal(0, 0.25, 3).astype(np.float32)
            moves.append((trk, T.reshape(-1)))
    mv = np.zeros(len(moves), synth.OBJECT_MOVE)
    for k, (trk, T) in enumerate(moves):
        mv[k]["track_id"], mv[k]["T"] = trk, T
    remove = [int(rng.integers(1, 8))] if rng.random() < 0.15 else None
    return depth, cloud, mv, remove


@pytest.mark.parametrize("params_name,seed,p_n", [("vkitti2", 1, 3), ("noisy3", 2, 3), ("nodepthnoise", 3, 3), ("kitti360", 4, 3),
                                                  ("vkitti2", 5, 1), ("noisy3", 6, 2), ("zed2", 7, 4)])
def test_random_frames_free_running(params_name, seed, p_n):
    cfg = dict(synth.CONFIGS["T0"], p_n=p_n)
    params = synth.PARAMS[params_name]
    rng = np.random.default_rng(seed)
    o, g = pu.make_pair(cfg, params, synth.noise_table())
    S = 1 << cfg["p_n"]
    pos = np.zeros(3)
    yaw = 0.0
    for t in range(120):
        pos = pos + rng.normal(0, 0.35, 3) * np.array([1.0, 0.2, 1.0])   # crosses voxel borders on all axes, both ways
        if t == 60:
            pos = pos + np.array([9.0, 0.0, -7.0])                        # and once more than half the map at a time
        yaw += rng.normal(0, 0.08)
        depth, cloud, mv, remove = random_frame(rng, cfg, params, t, pos, yaw)
        q = synth.yaw_quat(yaw).astype(np.float32)
        p32 = pos.astype(np.float32)
        o.update(depth, cloud, p32, q, mv, remove)
        g.update(depth, cloud, p32, q, mv, remove)
        if t % 10 == 9:
            g.synchronize()
            rep = pu.compare_maps(o, g, S, tag="%s frame %d: " % (params_name, t))
            assert not rep, "\n".join(rep)
    assert g.stats(count_live=True)["live_particles"] > 0
    g.close()
