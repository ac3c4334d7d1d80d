"""The table of older owner-set memberships (State::alias: a particle slot that sits in two moving objects' sets at once -
the reference's ObjectParticleHashMap holds real sets, object_layer.h:20-52) with MANY entries, and its overflow.

How thousands of such slots come about in a few frames: every pixel of frame 0 belongs to movable track 1, so every
particle born joins set 1; the camera then jumps by more than the map's extent, which re-stamps every slab - all those
particles are stale, their slots vacant, but nothing removes a stale index from its set (only resampling and
removeObjectByTrackID do); frame 1's births, all of track 2, take the same storage voxels (a ring buffer) and the same
first vacant slots: each such slot is now in set 2 AND still in set 1.  Rounds 3-5 capped the table at 8192 entries."""
import numpy as np
import pytest

from semantic_dsp_map_amd import binding, synth
from tests import parity_utils as pu

pytestmark = pytest.mark.gpu


def frame(rng, cfg, params, pos, track):
    """a random depth image whose every valid pixel belongs to `track`; camera at pos, looking down +z"""
    W, H = cfg["width"], cfg["height"]
    depth = (0.5 + 5.5 * rng.random((H, W))).astype(np.float32)
    jj, ii = np.meshgrid(np.arange(W), np.arange(H))
    xc = (jj - cfg["cx"]) / cfg["fx"] * depth
    yc = (ii - cfg["cy"]) / cfg["fy"] * depth
    pg = np.stack([xc, yc, depth], -1).reshape(-1, 3).astype(np.float64) + np.asarray(pos, np.float64)
    cloud = np.zeros(H * W, synth.LABELED_POINT)
    cloud["x"], cloud["y"], cloud["z"] = pg[:, 0], pg[:, 1], pg[:, 2]
    cloud["sigma"] = (params["depth_noise_zero_order"] + params["depth_noise_first_order"] * depth.reshape(-1)).astype(np.float32)
    cloud["track_id"], cloud["label_id"], cloud["is_valid"] = track, synth.LABEL_CAR, 1
    return depth, cloud


def run(update, cfg, params, seed=3):
    rng = np.random.default_rng(seed)
    q = synth.yaw_quat(0.0).astype(np.float32)
    far = np.array([cfg["voxel_size"] * (1 << cfg["x_n"]) + 0.2, 0.0, 0.0], np.float32)  # more than the map's extent along x
    d, c = frame(rng, cfg, params, (0, 0, 0), 1)
    update(0, d, c, np.zeros(3, np.float32), q, None, None)
    d, c = frame(rng, cfg, params, far, 2)
    update(1, d, c, far, q, None, None)           # set 2 takes the slots set 1 still lists
    d, c = frame(rng, cfg, params, far, 2)
    update(2, d, c, far, q, None, None)
    d, c = frame(rng, cfg, params, far, 3)
    update(3, d, c, far, q, None, [1])            # set 1 is wiped: through the table, nearly all of it
    T = np.eye(4, dtype=np.float32)
    T[:3, 3] = (0.3, 0.0, 0.1)
    mv = np.zeros(1, synth.OBJECT_MOVE)
    mv[0]["track_id"], mv[0]["T"] = 3, T.reshape(-1)
    d, c = frame(rng, cfg, params, far, 2)
    update(4, d, c, far, q, mv, None)             # the youngest object moves (its chunks hold entries of set 2's slots)
    d, c = frame(rng, cfg, params, far, 3)
    update(5, d, c, far, q, None, [2])


def test_more_older_memberships_than_the_old_table_held():
    cfg, params = synth.CONFIGS["T1"], synth.PARAMS["noisy3"]
    o, g = pu.make_pair(cfg, params, synth.noise_table())
    S = 1 << cfg["p_n"]
    peak = [0]

    def update(t, depth, cloud, pos, q, mv, rm):
        o.update(depth, cloud, pos, q, mv, rm)
        g.update(depth, cloud, pos, q, mv, rm, sync=True)
        s = g.stats()
        assert not s["alias_overflowed"]
        peak[0] = max(peak[0], s["alias_entries"])
        rep = pu.compare_maps(o, g, S, tag="frame %d: " % t)
        assert not rep, "\n".join(rep)

    run(update, cfg, params)
    assert peak[0] > 8192, "the scenario no longer produces more entries than rounds 3-5 could hold: %d" % peak[0]
    g.close()


def test_an_overflow_of_the_table_is_reported_by_every_later_frame():
    """include/sdm.h: once entries were dropped the sets are incomplete and SDM_ERR_CAPACITY is reported at every
    synchronisation - also after frames with removals or births only, which never rank the sets - until sdm_clear."""
    cfg, params = synth.CONFIGS["T0"], synth.PARAMS["noisy3"]
    g = binding.SdmMap(cfg, params, synth.noise_table())
    g.set_alias_cap(64)
    errors = {}

    def update(t, depth, cloud, pos, q, mv, rm):
        g.update(depth, cloud, pos, q, mv, rm)
        try:
            g.synchronize()
            errors[t] = None
        except binding.SdmError as e:
            errors[t] = str(e)

    run(update, cfg, params)
    assert errors[0] is None
    for t in range(1, 6):  # frame 1 overflows; 2 births only, 3 and 5 a removal, 4 a move
        assert errors[t] is not None and "SDM_ERR_CAPACITY" in errors[t], (t, errors[t])
    assert g.stats_unchecked()["alias_overflowed"] == 1
    g.clear()
    g.synchronize()
    assert g.stats()["alias_overflowed"] == 0
    g.close()
