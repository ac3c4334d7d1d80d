"""The per-image-row lists of the visible particles (k_visibility -> k_bin_rows) under a skewed load and a small
sdm_config.max_visible: a horizontal sheet of particles at the camera's height projects into ONE image row.  The lists of a
row are sized from what a row can hold, not from its share of max_visible, and a row's particles are spread over its
sub-lists by lane - a crowded row must not void a frame (SDM_ERR_CAPACITY) whose particles fit max_visible (round-4 review:
sub-list picked by workgroup, 64 entries each at this size - this state overflowed it)."""
import numpy as np
import pytest

from semantic_dsp_map_amd import binding, synth
from tests import parity_utils as pu

pytestmark = pytest.mark.gpu


def sheet_state(cfg):
    NX, NY, NZ, S = 1 << cfg["x_n"], 1 << cfg["y_n"], 1 << cfg["z_n"], 1 << cfg["p_n"]
    V = NX * NY * NZ
    size = np.float32(cfg["voxel_size"])
    st = {k: np.zeros(V * S, dt) for k, dt in binding.STATE_FIELDS}
    st["owner"][:] = 0xFFFF
    st["status"].reshape(V, S)[:, 0] = 5   # TIMEPTC
    rng = np.random.default_rng(3)
    iy = NY // 2                            # voxels whose lower face is the plane y = 0
    n = 0
    for iz in range(NZ // 2 + 3, NZ // 2 + 15):     # z = 1.2 m .. 6 m in front of the camera
        for ix in range(NX):
            v = (iz << (cfg["x_n"] + cfg["y_n"])) | (iy << cfg["x_n"]) | ix
            st["ts"][v * S] = 1
            for s in range(1, S):
                i = v * S + s
                st["status"][i] = 1
                st["ts"][i] = 1
                st["w"][i] = np.float32(0.3)
                st["track"][i] = 65535
                st["label"][i] = 3
                st["px"][i] = (np.float32(ix - NX // 2) + np.float32(rng.random())) * size
                st["py"][i] = np.float32(0.0005 + 0.003 * rng.random())       # a sheet 3 mm thick just below the optical axis: ONE image row
                st["pz"][i] = (np.float32(iz - NZ // 2) + np.float32(rng.random())) * size
                n += 1
    return st, n


def test_a_crowded_image_row_with_a_small_max_visible():
    cfg = synth.CONFIGS["T0"]
    params = synth.PARAMS["vkitti2"]
    st, n = sheet_state(cfg)
    o, g = pu.make_pair(cfg, params, synth.noise_table(), max_visible=4096)
    for m in (o, g):
        m.load_state(st)
        m.set_ring_state(dict(o.ring_state(), global_time_stamp=1))
    H, W = cfg["height"], cfg["width"]
    depth = np.full((H, W), 10.0, np.float32)     # a wall far behind the sheet: nothing is occluded
    sc = synth.Scene(cfg, n_static=0, n_dynamic=0)
    _, cloud, pos, q = sc.render(0, params)
    cloud = cloud.copy()
    cloud["is_valid"][:] = 0                       # no births: the frame is visibility + weight update + sweep
    o.update(depth, cloud, pos, q, sc.moves(0))
    g.update(depth, cloud, pos, q, sc.moves(0), sync=True)   # (raises SdmError on SDM_ERR_CAPACITY)
    n_vis = g.stats()["n_visible"]
    assert n_vis == o.stats()["n_visible"] and 1000 < n_vis <= 4096, (n_vis, n)
    cnt = g.bin_counts().reshape(H, W).sum(axis=1)
    assert cnt.max() > 0.9 * n_vis and cnt.max() > 8 * 64, "the sheet should fall into one image row: %r" % (cnt[cnt > 0],)
    rep = pu.compare_maps(o, g, 1 << cfg["p_n"], check_bins=True)
    assert not rep, "\n".join(rep)
    g.close()


def test_an_image_3000_pixels_wide():
    """The reference takes any g_image_width (settings/settings.h); the row kernel of the pixel bins holds one image row in LDS
    and took at most 2047 columns until round 6.  A 3000 x 40 camera (eight pixels per thread of k_bin_rows) on the T0 grid,
    a street scene with moving objects, six frames against the oracle with the bins compared."""
    cfg = dict(synth.CONFIGS["T0"], width=3000, height=40, fx=1200.0, fy=200.0, cx=1500.0, cy=20.0)
    params = synth.PARAMS["noisy3"]
    sc = synth.Scene(cfg, n_dynamic=2)
    o, g = pu.make_pair(cfg, params, synth.noise_table())
    n_vis = 0
    for t in range(6):
        depth, cloud, pos, q = sc.render(t, params)
        o.update(depth, cloud, pos, q, sc.moves(t))
        g.update(depth, cloud, pos, q, sc.moves(t), sync=True)
        rep = pu.compare_maps(o, g, 1 << cfg["p_n"], check_bins=True, tag="frame %d: " % t)
        assert not rep, "\n".join(rep)
        n_vis = max(n_vis, g.stats()["n_visible"])
    assert n_vis > 100, n_vis
    assert g.bin_counts().reshape(cfg["height"], cfg["width"])[:, 2048:].sum() > 0   # particles binned beyond the old limit's columns
    g.close()
    with pytest.raises(binding.SdmError):
        binding.SdmMap(dict(cfg, width=4096), params, synth.noise_table())
