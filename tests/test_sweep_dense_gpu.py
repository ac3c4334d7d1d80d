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

from semantic_dsp_map_amd import synth
from tests import parity_utils as pu
from tests.dense_state import random_state, stamps_for

pytestmark = pytest.mark.gpu


# (x_n >= 6: a chunk of 64 voxels lies in one x row of the ring - the dense kernel's "rows" path for the slab stamps)
@pytest.mark.parametrize("p_n,seed,x_n", [(3, 1, 5), (3, 2, 5), (2, 3, 5), (4, 4, 5), (1, 5, 5), (3, 6, 6), (3, 7, 6), (3, 8, 7), (2, 9, 6)])
def test_non_incremental_sweep_on_a_dense_random_state(p_n, seed, x_n):
    cfg = dict(synth.CONFIGS["T0"], p_n=p_n, x_n=x_n)
    params = synth.PARAMS["vkitti2"]
    _, _, frames = synth.make_frames("T0", 2, "vkitti2", n_dynamic=0)
    o, g = pu.make_pair(cfg, params, synth.noise_table())
    S = 1 << p_n
    st = random_state(cfg, seed)
    (sx, sy, sz), ring = stamps_for(o)
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


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("seed,x_n", [(21, 5), (22, 6)])
def test_sparse_voxels_inline_and_through_lists(mode, seed, x_n):
    """A non-incremental sweep evaluates the voxels of its sparse chunks in its first launch (k_occupancy_scan) or hands
    them to per-tile lists and a launch of their own (k_occupancy_scan_lists, k_occupancy_listed); the library picks per
    sweep from what the sweep before found.  Here each way is forced for every sweep of a run - four frames, each ending
    in a non-incremental sweep, with births, moves and ring shifts in between - and held to the oracle bit for bit.
    (Tiles with a few listed voxels and tiles with hundreds: the random state has sparse runs of chunks.)"""
    cfg = dict(synth.CONFIGS["T0"], p_n=3, x_n=x_n)
    params = synth.PARAMS["vkitti2"]
    sc = synth.Scene(cfg, n_dynamic=2, seed=6)
    o, g = pu.make_pair(cfg, params, synth.noise_table())
    g.force_sweep_lists(mode)
    st = random_state(cfg, seed, run=8)
    (sx, sy, sz), ring = stamps_for(o)
    for m in (o, g):
        m.load_state(st)
        m.set_stamps(sx, sy, sz)
        m.set_ring_state(ring)
    for t in range(4):
        depth, cloud, pos, q = sc.render(t, params)
        pos = pos + np.array([0.0, 0.0, 0.45 * t], np.float32)
        for m in (o, g):
            m.set_params(params)     # the next sweep is a non-incremental one
        o.update(depth, cloud, pos, q, sc.moves(t))
        g.update(depth, cloud, pos, q, sc.moves(t), sync=True)
        rep = pu.compare_maps(o, g, 8, check_results=True, tag="lists %d, frame %d: " % (mode, t))
        assert not rep, "\n".join(rep)
    assert int((g.voxels()["occ"] > 0).sum()) > 1000
    g.close()


@pytest.mark.parametrize("seed,x_n", [(11, 5), (12, 6), (13, 7)])
def test_group_hints_repeated_non_incremental_sweeps(seed, x_n):
    """The second and later non-incremental sweeps of a map: groups of 512 voxels whose chunks were all dense in the sweep
    before are skipped by k_occupancy_scan and classified by k_occupancy_dense itself (State::grp_hint).  The hint is about
    speed only - so the sweeps must give the oracle's results whatever has happened to a hinted group in between:
    nothing, births, culls and object moves in it, ring shifts on z (every frame) and on x and y (frames 4-5) that re-stamp
    slabs through it (voxels of hinted groups become unobserved, are emptied by culls, get new particles).  Every
    frame here ends in a NON-incremental sweep (sdm_set_params before it: the threshold 'may have changed'), compared
    bit for bit: state, flags' effects and every voxel result."""
    cfg = dict(synth.CONFIGS["T0"], p_n=3, x_n=x_n)
    params = synth.PARAMS["vkitti2"]
    sc = synth.Scene(cfg, n_dynamic=2, seed=5)
    o, g = pu.make_pair(cfg, params, synth.noise_table())
    st = random_state(cfg, seed, run=8)      # whole groups dense / sparse / empty
    (sx, sy, sz), ring = stamps_for(o)
    for m in (o, g):
        m.load_state(st)
        m.set_stamps(sx, sy, sz)
        m.set_ring_state(ring)
    n_occ, n_hint = [], []
    for t in range(6):
        depth, cloud, pos, q = sc.render(t, params)
        pos = pos + np.array([0.0, 0.0, 0.45 * t], np.float32)   # (a ring shift on z every frame: slabs through hinted groups are re-stamped)
        if t >= 4:   # the camera jumps sideways and up: x and y slabs through EVERY group are re-stamped, part of each hinted group turns unobserved
            pos = pos + np.array([2.1, 0.9, 0.0], np.float32) * (t - 3)
        for m in (o, g):
            m.set_params(params)     # the next sweep is a non-incremental one
        o.update(depth, cloud, pos, q, sc.moves(t))
        g.update(depth, cloud, pos, q, sc.moves(t), sync=True)
        rep = pu.compare_maps(o, g, 8, check_results=True, tag="frame %d: " % t)
        assert not rep, "\n".join(rep)
        n_occ.append(int((g.voxels()["occ"] > 0).sum()))
        n_hint.append(g.hinted_groups())
    assert min(n_occ) > 1000, n_occ
    # the sweeps after the first one did find hinted groups (about half of the state's groups are dense), and the jumps took
    # some of the hints away again
    assert n_hint[0] > (1 << (x_n + 10)) // 512 // 8 and n_hint[-1] < n_hint[0], n_hint
    g.close()


@pytest.mark.parametrize("seed,x_n", [(31, 6), (32, 7)])
def test_a_map_of_dense_groups_is_swept_in_one_launch(seed, x_n):
    """When the classification launch of a non-incremental sweep found every group of 512 voxels hinted, the next such
    sweep leaves it out: k_occupancy_dense alone, every group taken as hinted whatever its byte says
    (launch_occupancy, OCC_SKIP_SCAN).  A matter of speed only - so: a state whose every group is dense, frames that each
    end in a non-incremental sweep; the first sweep classifies (no hints yet), the second finds nothing to classify, the
    third and fourth are one launch; then the camera moves on, ring shifts re-stamp slabs through the groups, the
    single launch finds groups that are not dense any more and says so, and the sweeps after it are two launches again.
    Bit for bit against the oracle after every frame, and the way each sweep was issued is checked."""
    cfg = dict(synth.CONFIGS["T0"], p_n=3, x_n=x_n)
    params = synth.PARAMS["vkitti2"]
    sc = synth.Scene(cfg, n_dynamic=2, seed=5)
    o, g = pu.make_pair(cfg, params, synth.noise_table())
    st = random_state(cfg, seed, run=8, kinds=(0.0, 0.0, 1.0))
    ring = dict(o.ring_state(), global_time_stamp=3)
    for m in (o, g):
        m.load_state(st)
        m.set_ring_state(ring)
    n_groups = (1 << (x_n + cfg["y_n"] + cfg["z_n"])) // 512
    modes, hints = [], []
    for t in range(8):
        depth, cloud, pos, q = sc.render(0 if t < 4 else t, params)
        if t >= 4:
            pos = pos + np.array([0.0, 0.0, 0.45 * (t - 3)], np.float32)
        if t >= 6:
            pos = pos + np.array([2.1, 0.9, 0.0], np.float32) * (t - 5)
        for m in (o, g):
            m.set_params(params)     # the next sweep is a non-incremental one
        modes.append(g.sweep_mode() >> 1)    # how this frame's sweep will be issued
        o.update(depth, cloud, pos, q, sc.moves(t))
        g.update(depth, cloud, pos, q, sc.moves(t), sync=True)
        rep = pu.compare_maps(o, g, 8, check_results=True, tag="frame %d: " % t)
        assert not rep, "\n".join(rep)
        hints.append(g.hinted_groups())
    assert hints[0] == n_groups, (hints, n_groups)
    assert modes[:4] == [0, 0, 1, 1], modes      # classify / nothing to classify / one launch / one launch
    assert modes[4] == 1 and hints[4] < n_groups, (modes, hints)   # the single launch meets the re-stamped slabs ...
    assert modes[5:] == [0, 0, 0], modes         # ... and the classification launch is back
    g.close()
