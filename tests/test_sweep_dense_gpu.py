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
