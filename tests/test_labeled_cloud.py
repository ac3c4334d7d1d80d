"""generateLabeledPointCloud (utils/pointcloud_tools.h:88-310) — oracle restatement on the CPU,
device kernel through sdm_update_raw on the GPU."""
import numpy as np
import pytest

from oracle import oracle as orc
from semantic_dsp_map_amd import synth
from tests import parity_utils as pu


def frame_and_raw(cfg_name="T0", params_name="vkitti2", t=2, **kw):
    cfg = synth.CONFIGS[cfg_name]
    params = synth.PARAMS[params_name]
    sc = synth.Scene(cfg, **kw)
    depth, cloud, pos, q = sc.render(t, params)
    static_mask, objects = synth.raw_inputs(cfg, cloud, sc)
    pos64, q64 = sc.pose(t)
    return cfg, params, sc, depth, cloud, static_mask, objects, pos64, q64


@pytest.mark.parametrize("kw", [dict(n_dynamic=3), dict(n_dynamic=2, invalid_fraction=0.05), dict(n_dynamic=0)])
def test_oracle_cloud_matches_the_rendered_cloud(kw):
    cfg, params, sc, depth, cloud, static_mask, objects, pos64, q64 = frame_and_raw(**kw)
    o = orc.OracleMap(dict(cfg, bin_order=1), params)
    got = o.generate_cloud(depth, static_mask, synth.LABEL_TO_STATIC_INSTANCE, objects, pos64, q64)
    assert np.array_equal(got["is_valid"], cloud["is_valid"])
    v = cloud["is_valid"] > 0
    assert np.array_equal(got["track_id"][v], cloud["track_id"][v])
    assert np.array_equal(got["label_id"][v], cloud["label_id"][v])
    assert np.array_equal(got["sigma"], cloud["sigma"])
    for k in "xyz":   # the renderer back-projects along the ray, the reference through K^-1: same point up to rounding
        assert np.allclose(got[k][v], cloud[k][v], rtol=0, atol=2e-5)
    # a hand-checked pixel: identity pose, K^-1*(j,i,1)*d
    cfg0 = synth.CONFIGS["T0"]
    o2 = orc.OracleMap(dict(cfg0, bin_order=1), params)
    d = np.full((cfg0["height"], cfg0["width"]), 2.0, np.float32)
    c = o2.generate_cloud(d, None, synth.LABEL_TO_STATIC_INSTANCE, [], [0.5, 0, 0], [1, 0, 0, 0])
    i, j = 10, 100
    p = c[i * cfg0["width"] + j]
    assert p["x"] == np.float32((j / 80.0 - 64.0 / 80.0) * 2.0 + 0.5) and p["z"] == np.float32(2.0)
    assert p["track_id"] == 65535 and p["label_id"] == 0 and p["is_valid"] == 1


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_name,params_name,kw", [("T0", "vkitti2", dict(n_dynamic=3)),
                                                    ("T1", "zed2", dict(n_dynamic=2, invalid_fraction=0.05, yaw_rate_deg=5.0))])
def test_device_cloud_and_raw_frame_match_oracle(cfg_name, params_name, kw):
    from semantic_dsp_map_amd import binding
    cfg = synth.CONFIGS[cfg_name]
    params = synth.PARAMS[params_name]
    sc = synth.Scene(cfg, **kw)
    noise = synth.noise_table()
    o = orc.OracleMap(dict(cfg, bin_order=1), params, noise)
    g = binding.SdmMap(cfg, params, noise)
    S = 1 << cfg["p_n"]
    for t in range(4):
        depth, cloud, pos, q = sc.render(t, params)
        static_mask, objects = synth.raw_inputs(cfg, cloud, sc)
        pos64, q64 = sc.pose(t)
        want = o.generate_cloud(depth, static_mask, synth.LABEL_TO_STATIC_INSTANCE, objects, pos64, q64)
        g.update_raw(depth, static_mask, synth.LABEL_TO_STATIC_INSTANCE, objects, pos64, q64, sc.moves(t), sync=True)
        got = g.labeled_cloud()
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), "LabeledPoint image differs at frame %d" % t
        o.update(depth, want, pos64.astype(np.float32), q64.astype(np.float32), sc.moves(t))
        rep = pu.compare_maps(o, g, S, tag="frame %d: " % t)
        assert not rep, "\n".join(rep)
    g.close()


# ------------------------------------------------------------------ preset-specific parts (BOOST resize, ZED2 filters)
def boost_inputs(cfg, scale=0.5, seed=3):
    """Sensor-size inputs (2x the configured image) with recognisable content."""
    rng = np.random.default_rng(seed)
    W, H = cfg["width"], cfg["height"]
    sw, sh = int(W / scale) + 1, int(H / scale) + 1          # odd sizes: int(src * scale) still gives W x H
    assert int(sw * scale) == W and int(sh * scale) == H
    depth = (2.0 + 6.0 * rng.random((sh, sw))).astype(np.float32)
    depth[rng.random((sh, sw)) < 0.03] = np.nan
    static = rng.integers(2, 12, (sh, sw)).astype(np.uint8)
    m1 = np.zeros((sh, sw), np.uint8)
    m1[sh // 4: sh // 2, sw // 4: sw // 2] = 255
    m2 = np.zeros((sh, sw), np.uint8)
    m2[sh // 3: sh // 2 + 7, sw // 3: sw // 2 + 9] = 1
    return (sw, sh), depth, static, [(3, 14, m1), (9, 15, m2)]


def test_manual_resize_and_zed2_filters_in_the_oracle():
    cfg = synth.CONFIGS["T0"]
    params = synth.PARAMS["zed2"]
    W, H = cfg["width"], cfg["height"]
    o = orc.OracleMap(dict(cfg, bin_order=1), params)
    (sw, sh), depth, static, objects = boost_inputs(cfg)
    pos, q = [0.1, -0.2, 0.3], [1.0, 0.0, 0.0, 0.0]
    cloud, dres = o.generate_cloud_ex(depth, static, synth.LABEL_TO_STATIC_INSTANCE, objects, pos, q, src_size=(sw, sh), rescale=0.5)
    # manualResize (pointcloud_tools.h:1104-1133): dst(i, j) = src(int(i / scale), int(j / scale)), clamped
    si = np.minimum((np.arange(H, dtype=np.float32) * np.float32(2.0)).astype(np.int64), sh - 1)
    sj = np.minimum((np.arange(W, dtype=np.float32) * np.float32(2.0)).astype(np.int64), sw - 1)
    want = depth[np.ix_(si, sj)].reshape(-1)
    assert np.array_equal(np.isnan(want), np.isnan(dres)) and np.array_equal(want[~np.isnan(want)], dres[~np.isnan(dres)])
    # ... and the cloud is the plain generator applied to the resized images
    objs_r = [(t, l, m[np.ix_(si, sj)]) for t, l, m in objects]
    plain = o.generate_cloud(want.reshape(H, W), static[np.ix_(si, sj)], synth.LABEL_TO_STATIC_INSTANCE, objs_r, pos, q)
    assert np.array_equal(plain.view(np.uint8), cloud.view(np.uint8))
    # ZED2: sky pixels invalid; points of a movable instance outside their object's box become Background
    sky = int(synth.LABEL_TO_STATIC_INSTANCE[4])
    v = plain["is_valid"] > 0
    xs = np.unique(plain["x"][v & (plain["track_id"] == 3)].astype(np.float64))
    edge = 0.5 * (xs[len(xs) // 2] + xs[len(xs) // 2 + 1])    # between two float values: no point sits on the edge
    box3 = [xs[0] - 1, edge, -1e9, 1e9, -1e9, 1e9]            # keeps about half of object 3
    box9 = [-1e9, 1e9, -1e9, 1e9, -1e9, 1e9]                   # keeps all of object 9
    z, _ = o.generate_cloud_ex(want.reshape(H, W), static[np.ix_(si, sj)], synth.LABEL_TO_STATIC_INSTANCE, objs_r, pos, q,
                               sky_instance=sky, object_bbox=[box3, box9])
    was_sky = v & (plain["track_id"] == sky)
    assert was_sky.any() and not z["is_valid"][was_sky].any()
    t3 = v & (plain["track_id"] == 3)
    out = t3 & (plain["x"].astype(np.float64) > box3[1])
    assert out.any() and (t3 & ~out).any()
    assert np.all(z["track_id"][out] == 65535) and np.all(z["label_id"][out] == 0) and np.all(z["is_valid"][out] == 1)
    assert np.array_equal(z["x"][out], plain["x"][out]) and np.array_equal(z["sigma"][out], plain["sigma"][out])
    keep = v & ~was_sky & ~out
    assert np.array_equal(z[keep].view(np.uint8), plain[keep].view(np.uint8))


@pytest.mark.gpu
def test_device_preset_paths_match_oracle():
    from semantic_dsp_map_amd import binding
    cfg = synth.CONFIGS["T0"]
    params = synth.PARAMS["zed2"]
    W, H = cfg["width"], cfg["height"]
    noise = synth.noise_table()
    o = orc.OracleMap(dict(cfg, bin_order=1), params, noise)
    g = binding.SdmMap(cfg, params, noise)
    sky = int(synth.LABEL_TO_STATIC_INSTANCE[4])
    for t in range(3):
        (sw, sh), depth, static, objects = boost_inputs(cfg, seed=3 + t)
        pos, q = np.array([0.1 * t, -0.2, 0.3]), np.array([1.0, 0.0, 0.0, 0.0])
        boxes = [[-3.0, 0.4, -1e9, 1e9, 0.0, 6.5], [-1e9, 1e9, -0.3, 1e9, -1e9, 1e9]]
        want, dres = o.generate_cloud_ex(depth, static, synth.LABEL_TO_STATIC_INSTANCE, objects, pos, q, src_size=(sw, sh),
                                         rescale=0.5, sky_instance=sky, object_bbox=boxes)
        g.update_raw(depth, static, synth.LABEL_TO_STATIC_INSTANCE, objects, pos, q, sync=True, src_size=(sw, sh), rescale=0.5,
                     sky_instance=sky, object_bbox=boxes)
        got = g.labeled_cloud()
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), "LabeledPoint image differs at frame %d" % t
        assert (want["track_id"][want["is_valid"] > 0] == 65535).any()
        o.update(dres.reshape(H, W), want, pos.astype(np.float32), q.astype(np.float32), None)
        rep = pu.compare_maps(o, g, 1 << cfg["p_n"], tag="frame %d: " % t)
        assert not rep, "\n".join(rep)
    g.close()
