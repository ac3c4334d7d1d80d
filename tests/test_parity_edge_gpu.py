"""Parity on the corners the main parity clips do not reach: the other slot counts (S = 2, 16: every kernel is a
template on S and the per-voxel record layout depends on it), the independent filter with moving objects, degenerate
frames (no valid pixel, everything out of range, no objects, many objects) and an ego jump longer than the map."""
import numpy as np
import pytest

from semantic_dsp_map_amd import synth
from tests import parity_utils as pu

pytestmark = pytest.mark.gpu


def noise():
    return synth.noise_table()


def run_clip(cfg, params, frames, check_bins=True, allow_alias=True):
    o, g = pu.make_pair(cfg, params, noise())
    S = 1 << cfg["p_n"]
    for t, (depth, cloud, pos, q, moves) in enumerate(frames):
        o.update(depth, cloud, pos, q, moves)
        g.update(depth, cloud, pos, q, moves, sync=True)
        rep = pu.compare_maps(o, g, S, check_bins=check_bins, tag="frame %d: " % t)
        assert not rep, "\n".join(rep)
    # an index in two owner sets at once (object_layer.h:20-52 are real sets): handled through State::alias
    assert allow_alias or o.stats()["alias_events"] == 0
    st = g.stats(count_live=True)
    g.close()
    return st


@pytest.mark.parametrize("p_n,params_name", [(1, "noisy3"), (1, "vkitti2"), (4, "noisy3"), (4, "nodepthnoise")])
def test_other_slot_counts(p_n, params_name):
    cfg = dict(synth.CONFIGS["T0"], p_n=p_n)
    params = synth.PARAMS[params_name]
    sc = synth.Scene(cfg, n_dynamic=2, seed=11)
    frames = []
    for t in range(7):
        depth, cloud, pos, q = sc.render(t, params)
        frames.append((depth, cloud, pos, q, sc.moves(t)))
    st = run_clip(cfg, params, frames)
    assert st["live_particles"] > 0


@pytest.mark.parametrize("window_half", [1, 2, 4, 6, 7])
def test_other_window_sizes(window_half):
    """The weight update is instantiated per window size class (<= 64, <= 128, <= 256 window pixels per particle: one,
    two or four rounds of 64 lanes); the presets only use window_half 3 and 5."""
    cfg = dict(synth.CONFIGS["T0"], window_half=window_half)
    params = synth.PARAMS["noisy3"]
    sc = synth.Scene(cfg, n_dynamic=2, seed=13)
    frames = []
    for t in range(6):
        depth, cloud, pos, q = sc.render(t, params)
        frames.append((depth, cloud, pos, q, sc.moves(t)))
    st = run_clip(cfg, params, frames)
    assert st["live_particles"] > 0


@pytest.mark.parametrize("width", [1280, 1281, 2047])
def test_wide_images(width):
    """k_bin_rows lays out a whole image row per workgroup: two pixels per thread up to 1280 columns, four beyond (up to the
    2047 the list entries have bits for); the presets stop at 1242."""
    cfg = dict(synth.CONFIGS["T0"], width=width, height=24, fx=width * 0.6, fy=width * 0.6, cx=width / 2.0, cy=12.0)
    params = synth.PARAMS["noisy3"]
    sc = synth.Scene(cfg, n_dynamic=1, seed=17)
    frames = []
    for t in range(4):
        depth, cloud, pos, q = sc.render(t, params)
        frames.append((depth, cloud, pos, q, sc.moves(t)))
    st = run_clip(cfg, params, frames)
    assert st["live_particles"] > 0


def test_independent_filter_with_moving_objects():
    cfg = synth.CONFIGS["T0"]
    params = synth.PARAMS["kitti360"]
    sc = synth.Scene(cfg, n_dynamic=3, seed=5)
    frames = []
    for t in range(6):
        depth, cloud, pos, q = sc.render(t, params)
        frames.append((depth, cloud, pos, q, sc.moves(t)))
    run_clip(cfg, params, frames)


def test_degenerate_frames():
    cfg = synth.CONFIGS["T0"]
    params = synth.PARAMS["vkitti2"]
    sc = synth.Scene(cfg, n_dynamic=2, seed=3)
    frames = []
    for t in range(8):
        depth, cloud, pos, q = sc.render(t, params)
        moves = sc.moves(t)
        if t == 2:      # nothing valid: NaN depth, every point invalid
            depth = np.full_like(depth, np.nan)
            cloud = cloud.copy()
            cloud["is_valid"] = 0
        elif t == 4:    # everything beyond the depth range
            depth = np.full_like(depth, cfg["depth_max"] + 5.0)
            cloud = cloud.copy()
            cloud["is_valid"] = 0
        elif t == 5:    # objects stand still this frame
            moves = moves[:0]
        frames.append((depth, cloud, pos, q, moves))
    run_clip(cfg, params, frames)


def test_many_objects():
    cfg = synth.CONFIGS["T1"]
    params = synth.PARAMS["zed2"]
    sc = synth.Scene(cfg, n_dynamic=12, seed=9)
    frames = []
    for t in range(5):
        depth, cloud, pos, q = sc.render(t, params)
        frames.append((depth, cloud, pos, q, sc.moves(t)))
    assert len(frames[1][4]) >= 10
    run_clip(cfg, params, frames, allow_alias=True)  # 12 objects in a 6 m room do collide; the states still agree bit for bit


def test_ego_jump_longer_than_the_map():
    """updateEgoCenterPos with a step larger than the grid (operations.h:68-96, 1111-1191): every slab is recycled."""
    cfg = synth.CONFIGS["T0"]
    params = synth.PARAMS["vkitti2"]
    sc = synth.Scene(cfg, n_dynamic=0, seed=2)
    frames = []
    extent = (1 << cfg["x_n"]) * cfg["voxel_size"]
    for t in range(6):
        depth, cloud, pos, q = sc.render(t, params)
        if t >= 3:      # the clouds are re-centred with the camera so that points still fall into the map
            shift = np.array([2.5 * extent, 0.0, -1.5 * extent], np.float32)
            pos = pos + shift
            cloud = cloud.copy()
            cloud["x"] += shift[0]
            cloud["z"] += shift[2]
        frames.append((depth, cloud, pos, q, None))
    st = run_clip(cfg, params, frames)
    assert st["live_particles"] > 0


def test_long_clip_stays_bit_exact():
    """150 frames free-running (ring shifts in all directions, objects entering and leaving, resampling every frame):
    the two implementations never see each other's state, any divergence would compound."""
    cfg = synth.CONFIGS["T0"]
    params = synth.PARAMS["vkitti2"]
    sc = synth.Scene(cfg, n_dynamic=3, seed=21, yaw_rate_deg=2.0)
    o, g = pu.make_pair(cfg, params, noise())
    S = 1 << cfg["p_n"]
    for t in range(150):
        depth, cloud, pos, q = sc.render(t, params)
        moves = sc.moves(t)
        o.update(depth, cloud, pos, q, moves)
        g.update(depth, cloud, pos, q, moves)
        if t % 10 == 9 or t == 149:
            g.synchronize()
            rep = pu.compare_maps(o, g, S, tag="frame %d: " % t)
            assert not rep, "\n".join(rep)
    assert g.stats(count_live=True)["live_particles"] > 0
    # the clip must have exercised the rare case: a slot that sits in two objects' sets at once (object_layer.h:20-52)
    assert o.stats()["alias_events"] > 0
    g.close()


@pytest.mark.parametrize("cfg_name,params_name,kw", [
    ("T1", "zed2", dict(n_dynamic=3, seed=4, yaw_rate_deg=3.0)),
    ("T0", "noisy3", dict(n_dynamic=2, seed=8, yaw_rate_deg=-2.0)),
    ("T0", "nodepthnoise", dict(n_dynamic=3, seed=13)),
    ("T0", "kitti360", dict(n_dynamic=2, seed=17, yaw_rate_deg=1.0)),
])
def test_long_clips_other_presets(cfg_name, params_name, kw):
    cfg = synth.CONFIGS[cfg_name]
    params = synth.PARAMS[params_name]
    sc = synth.Scene(cfg, **kw)
    o, g = pu.make_pair(cfg, params, noise())
    S = 1 << cfg["p_n"]
    for t in range(100):
        depth, cloud, pos, q = sc.render(t, params)
        moves = sc.moves(t)
        o.update(depth, cloud, pos, q, moves)
        g.update(depth, cloud, pos, q, moves)
        if t % 20 == 19:
            g.synchronize()
            rep = pu.compare_maps(o, g, S, tag="frame %d: " % t)
            assert not rep, "\n".join(rep)
    g.close()


def test_mid_size_preset_clip():
    """C2 (the reference's ZED2 BOOST-mode grid: 128^3 x 4 slots, 640 x 360, window 3), 30 frames free-running."""
    cfg = synth.CONFIGS["C2"]
    params = synth.PARAMS["zed2"]
    sc = synth.Scene(cfg, n_dynamic=3, seed=6, yaw_rate_deg=2.0)
    o, g = pu.make_pair(cfg, params, noise())
    S = 1 << cfg["p_n"]
    for t in range(30):
        depth, cloud, pos, q = sc.render(t, params)
        moves = sc.moves(t)
        o.update(depth, cloud, pos, q, moves)
        g.update(depth, cloud, pos, q, moves)
        if t % 10 == 9:
            g.synchronize()
            rep = pu.compare_maps(o, g, S, tag="frame %d: " % t)
            assert not rep, "\n".join(rep)
    assert g.stats(count_live=True)["live_particles"] > 1000
    g.close()


def test_clear_in_the_middle_of_a_clip():
    """SemanticDSPMap::clear (semantic_dsp_map.h:74-81, operations.h:684-723): particles, stamps, time stamp and object
    sets go, the ring offset and the noise-table cursors stay."""
    cfg = synth.CONFIGS["T0"]
    params = synth.PARAMS["noisy3"]
    sc = synth.Scene(cfg, n_dynamic=2, seed=12)
    o, g = pu.make_pair(cfg, params, noise())
    S = 1 << cfg["p_n"]
    for t in range(10):
        if t == 5:
            o.clear()
            g.clear()
            rep = pu.compare_maps(o, g, S, check_results=False, tag="after clear: ")
            assert not rep, "\n".join(rep)
            assert o.ring_state() == g.ring_state()
        depth, cloud, pos, q = sc.render(t, params)
        moves = sc.moves(t)
        o.update(depth, cloud, pos, q, moves)
        g.update(depth, cloud, pos, q, moves, sync=True)
        rep = pu.compare_maps(o, g, S, tag="frame %d: " % t)
        assert not rep, "\n".join(rep)
    g.close()


def test_parameters_changed_between_frames():
    """setMapParameters / setMapOptions / setDepthNoiseModelParameters in the middle of a clip (semantic_dsp_map.h:101-166).
    The forgetting table is built at the first update and keeps its forgetting_rate (basic_algorithms.h:32-48)."""
    cfg = synth.CONFIGS["T0"]
    seq = ["vkitti2", "noisy3", "kitti360", "nodepthnoise", "zed2"]
    sc = synth.Scene(cfg, n_dynamic=2, seed=14)
    o, g = pu.make_pair(cfg, synth.PARAMS[seq[0]], noise())
    S = 1 << cfg["p_n"]
    for t in range(15):
        params = synth.PARAMS[seq[t // 3]]
        if t % 3 == 0 and t > 0:
            o.set_params(params)
            g.set_params(params)
        depth, cloud, pos, q = sc.render(t, params)
        moves = sc.moves(t)
        o.update(depth, cloud, pos, q, moves)
        g.update(depth, cloud, pos, q, moves, sync=True)
        rep = pu.compare_maps(o, g, S, tag="frame %d (%s): " % (t, seq[t // 3]))
        assert not rep, "\n".join(rep)
    g.close()


def test_time_stamp_wraps_at_16_bits():
    """Particle time stamps are 16 bits wide (mc_ring/buffer.h:57-79), global_time_stamp and the slab stamps 32: past
    frame 65535 the two disagree in the reference.  Whatever it does then, both implementations must do the same."""
    cfg = synth.CONFIGS["T0"]
    params = synth.PARAMS["vkitti2"]
    sc = synth.Scene(cfg, n_dynamic=2, seed=15)
    o, g = pu.make_pair(cfg, params, noise())
    S = 1 << cfg["p_n"]
    for t in range(14):
        if t == 3:
            for m in (o, g):
                rs = m.ring_state()
                rs["global_time_stamp"] = 65529
                m.set_ring_state(rs)
        depth, cloud, pos, q = sc.render(t, params)
        moves = sc.moves(t)
        o.update(depth, cloud, pos, q, moves)
        g.update(depth, cloud, pos, q, moves, sync=True)
        rep = pu.compare_maps(o, g, S, tag="frame %d: " % t)
        assert not rep, "\n".join(rep)
    assert g.ring_state()["global_time_stamp"] > 65536
    g.close()
