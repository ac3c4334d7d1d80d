"""The sweep epoch (sdm_internal.h: next_epoch, 1..254 and over again) wraps under test.

The in-frame occupancy sweep only looks into tiles marked with the frame's epoch, in one of two mark arrays by epoch
parity; the epoch is advanced on the host, outside a captured graph.  A clip shorter than 254 frames never sees the
wrap (254 -> 1: the one place where two consecutive epochs are not n, n + 1), so these clips run past it - free-running
against the oracle, in every way a frame can be issued, and with sdm_set_params / sdm_clear landing on the frames around
the wrap (both force a non-incremental sweep, which takes the epoch from the host)."""
import os

import numpy as np
import pytest

from semantic_dsp_map_amd import synth
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


_CLIPS = {}


def _clip(n, seed=31):
    """(cfg, params, frames): rendered once per process - every test of this file replays the same few hundred frames"""
    cfg, params = synth.CONFIGS["T0"], synth.PARAMS["vkitti2"]
    if (n, seed) not in _CLIPS:
        sc = synth.Scene(cfg, n_dynamic=3, seed=seed, yaw_rate_deg=1.5)
        _CLIPS[(n, seed)] = [sc.render(t, params) + (sc.moves(t),) for t in range(n)]
    return cfg, params, _CLIPS[(n, seed)]


def test_free_running_clip_across_the_epoch_wrap(graph_mode):
    """320 frames, no synchronisation between them except where the maps are compared: every 16 frames, and after each of
    the frames 248..262 (the wrap falls in there whatever the first frames did to the count)."""
    cfg, params, frames = _clip(320)
    o, g = pu.make_pair(cfg, params, synth.noise_table())
    S = 1 << cfg["p_n"]
    for t in range(320):
        depth, cloud, pos, q, moves = frames[t]
        o.update(depth, cloud, pos, q, moves)
        g.update(depth, cloud, pos, q, moves)
        if t % 16 == 15 or 248 <= t <= 262 or t == 319:
            g.synchronize()
            rep = pu.compare_maps(o, g, S, tag="frame %d: " % t)
            assert not rep, "\n".join(rep)
    st = g.stats(count_live=True)
    assert st["live_particles"] > 0
    assert st["graph_frames"] == (0 if graph_mode == "0" else 319), st
    g.close()


@pytest.mark.parametrize("event_frame", [252, 253, 254, 255, 256])
@pytest.mark.parametrize("event", ["set_params", "clear"])
def test_non_incremental_sweep_on_the_wrap_frame(graph_mode, event, event_frame):
    """sdm_set_params / sdm_clear right before the frames around the wrap: that frame's sweep is the non-incremental one
    (every voxel, epoch given by the host), the frames after it are incremental again."""
    if graph_mode in ("1", "4") and event_frame not in (253, 254):
        pytest.skip("the branched graph and the pieces share the epoch handling of the chain: two event frames are enough")
    cfg, params, frames = _clip(264, seed=37)
    o, g = pu.make_pair(cfg, params, synth.noise_table())
    S = 1 << cfg["p_n"]
    p2 = dict(params, occupancy_threshold=params["occupancy_threshold"] * 0.5)
    for t in range(264):
        depth, cloud, pos, q, moves = frames[t]
        if t == event_frame:
            if event == "set_params":
                o.set_params(p2)
                g.set_params(p2)
            else:
                o.clear()
                g.clear()
                moves = None  # the owner sets went with the map
        o.update(depth, cloud, pos, q, moves)
        g.update(depth, cloud, pos, q, moves)
        if t % 32 == 31 or t >= event_frame - 2:
            g.synchronize()
            rep = pu.compare_maps(o, g, S, tag="%s at %d, frame %d: " % (event, event_frame, t))
            assert not rep, "\n".join(rep)
    g.close()
