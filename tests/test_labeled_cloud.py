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
