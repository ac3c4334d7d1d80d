import numpy as np

from semantic_dsp_map_amd import synth


def test_frames_are_deterministic_and_well_formed():
    cfg, params, f1 = synth.make_frames("T0", 2, "vkitti2", n_dynamic=2)
    _, _, f2 = synth.make_frames("T0", 2, "vkitti2", n_dynamic=2)
    for a, b in zip(f1, f2):
        for x, y in zip(a, b):
            assert np.array_equal(np.asarray(x).view(np.uint8), np.asarray(y).view(np.uint8))
    depth, cloud, pos, q, moves = f1[1]
    assert depth.shape == (cfg["height"], cfg["width"]) and cloud.dtype.itemsize == 20 and cloud.size == depth.size
    v = cloud["is_valid"] > 0
    assert 0.5 < v.mean() <= 1.0
    assert np.all(depth.ravel()[v] >= cfg["depth_min"]) and np.all(depth.ravel()[v] <= cfg["depth_max"])
    assert len(moves) == 2 and set(moves["track_id"]) == {1, 2} and np.allclose(moves["T"][:, [0, 5, 10, 15]], 1)
    assert np.all(cloud["sigma"][v] == np.float32(0.2) + np.float32(0.01) * depth.ravel()[v])


def test_prefill_state_counts():
    cfg = synth.CONFIGS["T0"]
    sc = synth.Scene(cfg)
    st, ring, n = synth.prefill_state(cfg, sc, 7000)
    assert n == int((st["status"] == 1).sum()) and abs(n - 7000) < 8
    assert ring["global_time_stamp"] == 1
    st2, _, n2 = synth.prefill_state(cfg, sc, 7000, shard_rank=1, shard_count=2)
    assert len(st2["w"]) * 2 == len(st["w"])
