"""sdm_clear on a used map against the oracle's RingBufferOperations::clear (mc_ring/operations.h:684-723): status, position,
weight and time stamp of EVERY slot are reset, track id, label and forget count stay, owner sets go.  The map is loaded
with a random dense state first (every field of every slot holds something), for every slot count the library has a clear
kernel for: S = 8 and 16 take the piece-linear kernel (k_clear_map), S = 2 and 4 the slot-per-thread one; a map smaller
than one workgroup's stretch of voxels and one that ends inside a stretch are among the sizes."""
import numpy as np
import pytest

from semantic_dsp_map_amd import synth
from tests import dense_state
from tests import parity_utils as pu

pytestmark = pytest.mark.gpu

SMALL = dict(synth.CONFIGS["T0"])
SIZES = {
    "S8_128_voxels": dict(SMALL, x_n=3, y_n=2, z_n=2, p_n=3),       # less than one stretch of 256 voxels
    "S8_T0": dict(SMALL, p_n=3),                                    # 32768 voxels
    "S8_2pow17": dict(SMALL, x_n=6, y_n=5, z_n=6, p_n=3),
    "S16_T0": dict(SMALL, p_n=4),
    "S16_128_voxels": dict(SMALL, x_n=3, y_n=2, z_n=2, p_n=4),
    "S4_T0": dict(SMALL, p_n=2),
    "S2_T0": dict(SMALL, p_n=1),
}


@pytest.mark.parametrize("name", list(SIZES))
def test_clear_of_a_dense_random_state(name):
    cfg = SIZES[name]
    params = synth.PARAMS["noisy3"]
    noise = synth.noise_table(seed=5)
    o, g = pu.make_pair(cfg, params, noise)
    S = 1 << cfg["p_n"]
    st = dense_state.random_state(cfg, seed=11)
    rng = np.random.default_rng(3)
    n = st["w"].size
    # every field holds something in every slot, the INVALID ones included: clear() must reset what it resets everywhere
    # and leave track id, label and forget count alone everywhere
    st["forget"][:] = rng.integers(0, 4, size=n)
    dead = st["status"] == 0
    st["track"][dead] = rng.integers(0, 65536, size=int(dead.sum()))
    st["label"][dead] = rng.integers(0, 256, size=int(dead.sum()))
    st["w"][dead] = rng.random(int(dead.sum()), np.float32)
    st["ts"][dead] = rng.integers(0, 4, size=int(dead.sum()))
    own = rng.random(n) < 0.2
    st["owner"][own] = rng.integers(1, 9, size=int(own.sum()))
    for m in (o, g):
        m.load_state(st)
    if min(cfg["x_n"], cfg["y_n"], cfg["z_n"]) >= 5:  # (the slabs stamps_for re-stamps exist)
        (sx, sy, sz), ring = dense_state.stamps_for(o)
        for m in (o, g):
            m.set_stamps(sx, sy, sz)
            m.set_ring_state(ring)
    rep = pu.compare_maps(o, g, S, check_results=False, tag="loaded: ")
    assert not rep, "\n".join(rep)
    o.clear()
    g.clear()
    rep = pu.compare_maps(o, g, S, check_results=False, tag="after clear: ")
    assert not rep, "\n".join(rep)
    sg = g.dump_state()
    assert np.array_equal(sg["track"], st["track"]) and np.array_equal(sg["label"], st["label"])
    assert np.array_equal(sg["forget"], st["forget"])
    assert not sg["w"].any() and not sg["ts"].any() and not sg["px"].any() and np.all(sg["owner"] == 0xFFFF)
    assert np.all(sg["status"].reshape(-1, S)[:, 0] == 5) and not sg["status"].reshape(-1, S)[:, 1:].any()
    # the map works after it: two frames, results included (the first sweep after a clear is the non-incremental one)
    sc = synth.Scene(cfg, n_dynamic=1, seed=4)
    for t in range(2 if cfg["x_n"] + cfg["y_n"] + cfg["z_n"] >= 15 else 0):
        depth, cloud, pos, q = sc.render(t, params)
        moves = sc.moves(t)
        o.update(depth, cloud, pos, q, moves)
        g.update(depth, cloud, pos, q, moves, sync=True)
        rep = pu.compare_maps(o, g, S, tag="frame %d after clear: " % t)
        assert not rep, "\n".join(rep)
    g.close()


def test_clear_after_frames_keeps_the_forget_counts_the_frames_left():
    """The forget counts live in a byte plane of their own (round 5) that the weight update, the births and the object moves
    write: a map those stages have worked on - not a loaded state - is cleared, and every field of every slot, the forget
    counts of live and dead slots included, must come out like the oracle's."""
    cfg = synth.CONFIGS["T0"]
    params = synth.PARAMS["zed2"]
    o, g = pu.make_pair(cfg, params, synth.noise_table())
    S = 1 << cfg["p_n"]
    sc = synth.Scene(cfg, n_dynamic=3, seed=21)
    for t in range(9):
        depth, cloud, pos, q = sc.render(t, params)
        moves = sc.moves(t)
        o.update(depth, cloud, pos, q, moves)
        g.update(depth, cloud, pos, q, moves, sync=True)
    so = o.dump_state()
    assert (so["forget"] > 0).sum() >= 40 and (so["forget"][so["status"] == 0] > 0).sum() >= 20, "the clip leaves forget counts behind, in dead slots too"
    rep = pu.compare_maps(o, g, S, tag="before clear: ")
    assert not rep, "\n".join(rep)
    o.clear()
    g.clear()
    rep = pu.compare_maps(o, g, S, check_results=False, tag="after clear: ")
    assert not rep, "\n".join(rep)
    assert np.array_equal(g.dump_state()["forget"], so["forget"])
    g.close()
