"""bench.py's `driven` workload with the oracle BESIDE the GPU from frame 0.

An EMPTY C3 map (256^3 voxels, 8 slots) and synth.DRIVEN_FRAMES = 220 frames of a 66 m drive down a cluttered street with
12 moving objects and three noisy births per point - more than the map is long, ring shifts on z (every frame or two)
and x.  Nothing is prefilled: the map's ~0.2 M live particles are the ones the filter puts there, and the other parity
tests at this grid size either start from a prefilled state or hand a GPU-grown state to the oracle.  Here both sides
start from nothing and run free:

  * canonical order (bin_order = 1), bit for bit: every voxel result every 20 frames, the whole particle state (every field
    of every slot, ring state, slab stamps) at frame 99 and at the frame where the orders part;
  * the last 5 frames against the oracle's LITERAL order (bin_order = 0: the reference's BFS push-order sums,
    semantic_dsp_map.h:1029, mc_ring/operations.h:1405-1407) started from that common state: identical integers in the
    particle state and the voxel results, POSITIONS bit for bit (a summation order cannot move a particle), weights within
    an absolute 1e-4 (north_star's bar), the un-clamped weight sums within 1e-4 relative where they exceed 1.

Reference: SemanticDSPMap::subObjectLevelUpdate, semantic_dsp_map.h:576-955.  About four minutes (220 oracle frames at
~0.4 s, 220 rendered frames, four comparisons of 134 M slots - round 5 made eleven, a minute more)."""
import numpy as np
import pytest

from semantic_dsp_map_amd import synth
from tests import parity_utils as pu

pytestmark = pytest.mark.gpu


def test_drive_from_an_empty_map_with_the_oracle_beside_the_gpu():
    cfg = synth.CONFIGS["C3"]
    params = synth.PARAMS[synth.DRIVEN_PARAMS]
    scene_kw = dict(synth.DRIVEN_SCENE)
    n, n_literal = synth.DRIVEN_FRAMES, 5
    S = 1 << cfg["p_n"]
    scene = synth.Scene(cfg, **scene_kw)
    rendered = synth.render_frames(cfg, params, scene_kw, range(n))
    noise = synth.noise_table()
    o, g = pu.make_pair(cfg, params, noise, bin_order=1)
    n_vis, moved = [], 0
    t_split = n - n_literal
    for t in range(t_split):
        depth, cloud, pos, q = rendered[t]
        moves = scene.moves(t)
        o.update(depth, cloud, pos, q, moves)
        g.update(depth, cloud, pos, q, moves, sync=True)
        so, sg = o.stats(), g.stats()
        assert so["n_visible"] == sg["n_visible"], "frame %d: visible particles %d (oracle) / %d (gpu)" % (t, so["n_visible"], sg["n_visible"])
        n_vis.append(sg["n_visible"])
        moved += sg.get("n_moved", 0)
        if t == 99 or t == t_split - 1:
            rep = pu.compare_maps(o, g, S, check_results=True, tag="frame %d: " % t)
            assert not rep, "\n".join(rep)
        elif t % 20 == 19:
            vo, vg = o.voxels(), g.voxels()
            for k in ("occ", "label", "track", "wsum"):
                r = pu.diff_report("frame %d: voxels.%s" % (t, k), vo[k], vg[k])
                assert not r, r
    # the drive did what it is meant to: a population the filter grew, objects that moved, the ring shifted on two axes
    live = g.stats(count_live=True)
    assert live["live_particles"] > 100000 and min(n_vis[60:]) > 20000, (live["live_particles"], n_vis[-5:])
    ring = g.ring_state()
    assert abs(ring["moved_steps"][2]) >= 300 and abs(ring["moved_steps"][0]) >= 3, ring
    assert moved > 0
    # the last frames against the literal order, from the common state
    # (a deep copy: the owner sets of the 12 objects hold indices twice where their particles have met - the reference's sets
    # are real sets - and a state dump carries one owner per index)
    lit = o.clone(bin_order=0)
    del o
    for t in range(t_split, n):
        depth, cloud, pos, q = rendered[t]
        moves = scene.moves(t)
        lit.update(depth, cloud, pos, q, moves)
        g.update(depth, cloud, pos, q, moves, sync=True)
        if t == n - 1:  # every field of every slot at the end of the literal phase (the results: every frame)
            so, sg = lit.dump_state(), g.dump_state()
            for k in ("status", "ts", "track", "label", "forget", "owner", "px", "py", "pz"):
                assert np.array_equal(pu.bits(so[k]), pu.bits(sg[k])), "frame %d: %s differs from the literal-order oracle" % (t, k)
            alive = so["status"] != 0
            a, b = so["w"][alive], sg["w"][alive]
            assert np.all(np.abs(a - b) <= 1e-4), "frame %d: w, max |diff| %g" % (t, float(np.max(np.abs(a - b))))
            del so, sg
        vo, vg = lit.voxels(), g.voxels()
        for k in ("occ", "label", "track"):
            assert np.array_equal(vo[k], vg[k]), "frame %d: voxels.%s differs from the literal-order oracle" % (t, k)
        # (a voxel's weight SUM is taken before the clamp and reaches a few hundred on surfaces seen for a hundred frames:
        # 1e-4 relative there - four float ulps at 400 are 1.2e-4 -, 1e-4 absolute where it is a probability)
        assert np.all(np.abs(vo["wsum"] - vg["wsum"]) <= 1e-4 * np.maximum(1.0, np.abs(vo["wsum"])))
        assert lit.stats()["n_visible"] == g.stats()["n_visible"]
    g.close()
