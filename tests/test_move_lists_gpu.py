"""Moved copies piling up in a few voxels: the lists k_move_replay works through, at every length.

moveParticlesInSetsByTransformations (mc_ring/operations.h:321-362) re-inserts the copies of all moved particles in
(object, index) order, each into the first vacant slot of the voxel it lands in, and drops what finds its voxel full.  The
library keeps the ranks of a voxel's first fourteen arrivals in a row of their own (one load, sorted in registers), chains
the later ones, keeps up to 96 ranks of a long list in LDS and goes through row and chain again per batch beyond that
(csrc/moves.hip, move_link / k_move_replay).  The clips of the other tests move rigid boxes: a voxel receives two copies
on average, seventeen at most on the 220-frame drive.  Here the motion is a matrix that sends EVERY particle of an object
to one point (the reference multiplies whatever 4 x 4 matrix it is handed, operations.h:336-343), and the table noise
(sigma = half a voxel) spreads the thousands of copies over the voxels around it: a thousand and more in the middle one,
hundreds in its face neighbours, dozens in the edge neighbours, a few further out.  Bit for bit against the oracle."""
import numpy as np
import pytest

from semantic_dsp_map_amd import synth
from tests import parity_utils as pu
from tests.test_alias_table_gpu import frame

pytestmark = pytest.mark.gpu


def collapse_to(point):
    T = np.zeros((4, 4), np.float32)
    T[:3, 3] = point
    T[3, 3] = 1.0
    return T


@pytest.mark.parametrize("name", ["T0", "T1"])
def test_an_object_collapses_into_the_voxels_round_one_point(name):
    cfg, params = synth.CONFIGS[name], synth.PARAMS["noisy3"]
    size = cfg["voxel_size"]
    o, g = pu.make_pair(cfg, params, synth.noise_table(stddev=0.5 * size))
    S = 1 << cfg["p_n"]
    rng = np.random.default_rng(11)
    q = synth.yaw_quat(0.0).astype(np.float32)
    pos = np.zeros(3, np.float32)
    centre = np.array([0.5 * size, 0.5 * size, 7.5 * size], np.float32)      # the middle of a voxel
    corner = np.array([-2.0 * size, 1.0 * size, 5.0 * size], np.float32)     # where eight voxels meet
    moved, kept = [], []

    def step(t, track, mv_track=None, T=None, rm=None):
        d, c = frame(rng, cfg, params, pos, track)
        mv = None
        if mv_track is not None:
            mv = np.zeros(1, synth.OBJECT_MOVE)
            mv[0]["track_id"], mv[0]["T"] = mv_track, T.reshape(-1)
        o.update(d, c, pos, q, mv, rm)
        g.update(d, c, pos, q, mv, rm, sync=True)
        s = g.stats()
        moved.append(s["n_moved"])
        kept.append(s["n_move_reinserted"])
        rep = pu.compare_maps(o, g, S, tag="frame %d: " % t)
        assert not rep, "\n".join(rep)

    step(0, 1)                                    # every pixel is object 1: thousands of members
    step(1, 2, 1, collapse_to(centre))            # all of them to the middle of one voxel
    step(2, 2, 2, collapse_to(corner))            # object 2 (born in frame 1) to a corner: eight voxels share the bulk
    T = np.eye(4, dtype=np.float32)
    T[:3, 3] = (size, 0.0, -size)
    step(3, 3, 1, T)                              # what is left of object 1 moves on, rigidly
    step(4, 3, 3, collapse_to(centre), rm=[2])    # object 3 into the voxels object 1 filled and left; object 2 is removed
    assert moved[1] > 1500 and moved[2] > 1500 and moved[4] > 500, moved
    # the voxels round the point are full after a few dozen copies: nearly everything is dropped
    for t in (1, 2, 4):
        assert 0 < kept[t] < moved[t] // 4, (moved, kept)
    g.close()
