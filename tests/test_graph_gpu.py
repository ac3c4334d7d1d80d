"""The frame replayed as a hipGraph against the CPU oracle, bit for bit.

The default policy (SDM_GRAPH=2) replays plain frames from the branched graph, or from the chain on a host that is slow
at issuing launches, so a given box only ever sees one of the three ways a frame can be issued: these tests force each
of them (SDM_GRAPH=0: launch by launch; 1: frustum and birth chains as branches; 3: one chain of kernel nodes).  The
graph is captured from the very launches of the launch-by-launch frame; what changes from frame to frame travels in the
one kernel-node parameter that is updated before every replay."""
import os

import numpy as np
import pytest

from semantic_dsp_map_amd import binding, synth
from tests import parity_utils as pu

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["0", "1", "3", "4"], ids=["launches", "branched", "chain", "pieces"])
def graph_mode(request):
    old = os.environ.get("SDM_GRAPH")
    os.environ["SDM_GRAPH"] = request.param  # read when a map is created
    yield request.param
    if old is None:
        os.environ.pop("SDM_GRAPH", None)
    else:
        os.environ["SDM_GRAPH"] = old


@pytest.mark.parametrize("cfg_name,params_name,n_frames,scene_kw", [
    ("T0", "vkitti2", 8, dict(n_dynamic=3)),
    ("T1", "zed2", 6, dict(n_dynamic=2)),
    ("T0", "noisy3", 6, dict(n_dynamic=0)),
])
def test_graph_replay_matches_oracle(graph_mode, cfg_name, params_name, n_frames, scene_kw):
    cfg, params, frames = synth.make_frames(cfg_name, n_frames, params_name, **scene_kw)
    o, g = pu.make_pair(cfg, params, synth.noise_table())
    S = 1 << cfg["p_n"]
    for t, (depth, cloud, pos, q, moves) in enumerate(frames):
        rm = [2] if t == n_frames - 2 and scene_kw.get("n_dynamic", 0) >= 2 else None   # a removal inside a replayed frame
        mv = moves[moves["track_id"] != 2] if rm else moves
        o.update(depth, cloud, pos, q, mv, rm)
        g.update(depth, cloud, pos, q, mv, rm, sync=True)
        rep = pu.compare_maps(o, g, S, check_bins=True, tag="frame %d: " % t)
        assert not rep, "\n".join(rep)
    st = g.stats()
    # the first frame after creation is the non-incremental one and takes the launches; the others are replays
    assert st["graph_frames"] == (0 if graph_mode == "0" else n_frames - 1), st
    g.close()


def test_graph_is_recaptured_when_parameters_change(graph_mode):
    """sdm_set_params between frames: the filter constants are kernel arguments of the captured nodes, so the graph
    is captured again; the frame after the change is a non-incremental one (launch by launch), then replays resume."""
    cfg, params, frames = synth.make_frames("T0", 8, "vkitti2", n_dynamic=2)
    o, g = pu.make_pair(cfg, params, synth.noise_table())
    S = 1 << cfg["p_n"]
    p2 = dict(params, occupancy_threshold=params["occupancy_threshold"] * 0.5)
    for t, (depth, cloud, pos, q, moves) in enumerate(frames):
        if t == 4:
            o.set_params(p2)
            g.set_params(p2)
        o.update(depth, cloud, pos, q, moves)
        g.update(depth, cloud, pos, q, moves, sync=True)
        rep = pu.compare_maps(o, g, S, check_bins=True, tag="frame %d: " % t)
        assert not rep, "\n".join(rep)
    st = g.stats()
    assert (st["graph_frames"] == 0 and st["direct_frames"] == 8) if graph_mode == "0" else (st["graph_frames"] == 6 and st["direct_frames"] == 2), st
    g.close()


def test_graph_replay_without_synchronisation(graph_mode):
    """Frames issued back to back (no host synchronisation in between, inputs resident on the device): the replay of
    frame t+1 is enqueued while frame t runs."""
    cfg, params, frames = synth.make_frames("T0", 10, "vkitti2", n_dynamic=3)
    o, g = pu.make_pair(cfg, params, synth.noise_table())
    S = 1 << cfg["p_n"]
    dev = [(g.device_put(d), g.device_put(c)) for d, c, _, _, _ in frames]
    for (depth, cloud, pos, q, moves), (dd, dc) in zip(frames, dev):
        o.update(depth, cloud, pos, q, moves)
        g.update(dd, dc, pos, q, moves, on_device=True)
    g.synchronize()
    rep = pu.compare_maps(o, g, S, tag="after 10 frames: ")
    assert not rep, "\n".join(rep)
    assert g.stats()["graph_frames"] == (0 if graph_mode == "0" else 9)
    g.close()


def test_issue_mode_switched_between_frames():
    """sdm_set_issue_mode: the ways to issue a frame taken in turn inside one clip, three frames each (the graphs of the
    mode before are dropped, the next plain frame captures anew) - the map stays bit-exact against the oracle."""
    cfg, params, frames = synth.make_frames("T0", 16, "vkitti2", n_dynamic=3)
    o, g = pu.make_pair(cfg, params, synth.noise_table())
    S = 1 << cfg["p_n"]
    modes = [0, 4, 3, 1, 0]
    for t, (depth, cloud, pos, q, moves) in enumerate(frames):
        if t % 3 == 1:
            g.set_issue_mode(modes[(t // 3) % len(modes)])
        o.update(depth, cloud, pos, q, moves)
        g.update(depth, cloud, pos, q, moves, sync=True)
        rep = pu.compare_maps(o, g, S, tag="frame %d: " % t)
        assert not rep, "\n".join(rep)
    st = g.stats()
    assert st["graph_frames"] >= 6 and st["direct_frames"] >= 4, st
    with pytest.raises(Exception):
        g.set_issue_mode(7)
    g.close()
