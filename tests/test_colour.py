"""colouring + packing of the emitted cloud (semantic_dsp_map.h:1274-1376).

CPU: known answers for the oracle's restatement of OpenCV's 8-bit RGB <-> HSV conversions (oracle/colour.py), worked out
by hand from the published integer algorithm (hsv_shift 12, cvRound tables) - the reference pins no OpenCV version and
the image has none, so this is what the restatement is held to.
GPU: k_emit_points_rgb against the restatement, bit for bit, for every colour branch: Background voxels through the jet
map, static labels, movable instances (both output formats), guessed-occupied white, the V x 0.7 dimming outside the
view, free space, camera-centred output."""
import numpy as np
import pytest

from oracle import colour as col
from semantic_dsp_map_amd import synth

LABEL_BGR = np.zeros((256, 3), np.uint8)
for _l, _c in {0: (0, 0, 0), 2: (200, 0, 210), 3: (255, 200, 90), 4: (0, 199, 0), 5: (0, 240, 90), 6: (140, 140, 140), 7: (100, 60, 100),
               8: (255, 100, 250), 9: (0, 255, 255), 10: (0, 200, 200), 11: (0, 130, 255), 12: (80, 80, 80), 13: (60, 60, 160),
               14: (80, 127, 255), 15: (139, 139, 0)}.items():   # utils/data_base.h:216-232
    LABEL_BGR[_l] = _c
PERM = np.random.default_rng(5).permutation(256).astype(np.uint8)


def test_rgb2hsv_known_answers():
    # primaries and secondaries: h = 30 * sector, full saturation and value
    rgb = np.array([[255, 0, 0], [255, 255, 0], [0, 255, 0], [0, 255, 255], [0, 0, 255], [255, 0, 255]], np.uint8)
    assert col.rgb2hsv_8u(rgb).tolist() == [[0, 255, 255], [30, 255, 255], [60, 255, 255], [90, 255, 255], [120, 255, 255], [150, 255, 255]]
    # greys: no hue, no saturation
    assert col.rgb2hsv_8u(np.array([[0, 0, 0], [128, 128, 128], [255, 255, 255]], np.uint8)).tolist() == [[0, 0, 0], [0, 0, 128], [0, 0, 255]]
    # (128, 64, 32): v = 128, diff = 96; sdiv[128] = 8160 -> s = (96 * 8160 + 2048) >> 12 = 191;
    # v == r -> h = g - b = 32; hdiv180[96] = cvRound(737280 / 576) = 1280 -> (32 * 1280 + 2048) >> 12 = 10
    assert col.rgb2hsv_8u(np.array([[128, 64, 32]], np.uint8)).tolist() == [[10, 191, 128]]
    # a negative hue wraps by 180: (200, 10, 60): v == r, h = g - b = -50, diff = 190, hdiv180[190] = cvRound(737280 / 1140) = 647
    # -> (-50 * 647 + 2048) >> 12 = floor(-30302 / 4096) = -8 -> 172; s = (190 * cvRound(1044480 / 200) + 2048) >> 12 = 242
    assert col.rgb2hsv_8u(np.array([[200, 10, 60]], np.uint8)).tolist() == [[172, 242, 200]]
    assert int(col.SDIV[200]) == 5222 and int(col.HDIV180[190]) == 647 and int(col.SDIV[1]) == 255 << 12


def test_hsv2rgb_known_answers_and_round_trip():
    hsv = np.array([[0, 255, 255], [30, 255, 255], [60, 255, 255], [90, 255, 255], [120, 255, 255], [150, 255, 255], [77, 0, 99]], np.uint8)
    assert col.hsv2rgb_8u(hsv).tolist() == [[255, 0, 0], [255, 255, 0], [0, 255, 0], [0, 255, 255], [0, 0, 255], [255, 0, 255], [99, 99, 99]]
    # h = 15 (half way through sector 0), s = v = 255: tab = (1, 0, 0.5, 0.5) -> r = 255, g = tab[3] = 0.5 -> 127.5 -> 128 (half to even), b = 0
    assert col.hsv2rgb_8u(np.array([[15, 255, 255]], np.uint8)).tolist() == [[255, 128, 0]]
    # the round trip is the identity on the fully saturated hues and on greys, and close elsewhere (h has 180 levels)
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, (4096, 3)).astype(np.uint8)
    back = col.hsv2rgb_8u(col.rgb2hsv_8u(rgb)).astype(int)
    assert np.abs(back - rgb.astype(int)).max() <= 6
    jet = col.jet_256().astype(np.uint8)
    assert np.abs(col.hsv2rgb_8u(col.rgb2hsv_8u(jet)).astype(int) - jet.astype(int)).max() <= 4


def test_jet_map_matches_the_reference_formula():
    jet = col.jet_256()
    assert jet[0].tolist() == [0, 0, 0] and jet[63].tolist() == [0, 0, 252] and jet[64].tolist() == [0, 0, 255]
    assert jet[128].tolist() == [0, 255, 255] and jet[191].tolist() == [252, 255, 3] and jet[255].tolist() == [255, 3, 0]


@pytest.mark.gpu
@pytest.mark.parametrize("evaluation_format,zero_center", [(False, False), (True, False), (False, True)])
def test_emitted_colours_match_the_oracle(evaluation_format, zero_center):
    from semantic_dsp_map_amd import binding
    cfg, params, frames = synth.make_frames("T1", 6, "vkitti2", n_dynamic=3)
    noise = synth.noise_table()
    g = binding.SdmMap(cfg, params, noise)
    for depth, cloud, pos, q, moves in frames[:5]:
        g.update(depth, cloud, pos, q, moves, sync=True)
    # hand-made voxels behind the camera (never in view): a guessed birth (-> occ 2), a Background particle (-> jet colour),
    # a static label, a movable instance
    st = g.dump_state()
    S = 1 << cfg["p_n"]
    ring = g.ring_state()
    gts = ring["global_time_stamp"]
    NX, NY = 1 << cfg["x_n"], 1 << cfg["y_n"]
    made = []
    for k, (status, w, track, label) in enumerate([(3, 0.1, 65535, 0), (1, 0.9, 65535, 0), (1, 0.9, 65529, 7), (1, 0.9, 3, 15),
                                                   (1, 0.9, 300, 14), (1, 0.9, 65535, 0)]):
        v = (2 + k) + NX * (3 + 2 * k) + NX * NY * 1          # ring voxel, low z row: behind the camera for this clip
        i = v * S + 1
        st["status"][i], st["w"][i], st["track"][i], st["label"][i], st["ts"][i] = status, w, track, label, gts
        st["px"][i] = st["py"][i] = st["pz"][i] = 0.0
        st["ts"][v * S] = gts                                   # observed
        made.append(v)
    g.load_state(st)
    g.set_ring_state(ring)
    depth, cloud, pos, q, moves = frames[5]
    g.update(depth, cloud, pos, q, moves, sync=True)
    g.set_colours(LABEL_BGR, PERM, background_label=0, evaluation_format=evaluation_format)
    pts, n = g.occupied(mark_fov=True, zero_center=zero_center)
    rgbp, n2 = g.occupied_rgb(zero_center=zero_center)
    assert n == n2 and n > 100
    assert np.array_equal(pts["x"].view(np.uint32), rgbp["x"].view(np.uint32)) and np.array_equal(pts["z"].view(np.uint32), rgbp["z"].view(np.uint32))
    assert np.all(rgbp["one"] == 1.0) and np.all(rgbp["a"] == 255)
    occ = pts["occ"] & 0x3f
    oof = (pts["occ"] & 0x40) != 0
    want = col.colour_points(pts["z"], pts["y"], pts["track"], pts["label"], occ, oof, LABEL_BGR, PERM, 0, cfg["max_movable_track"],
                             evaluation_format=evaluation_format)
    got = np.stack([rgbp["r"], rgbp["g"], rgbp["b"]], 1)
    bad = np.flatnonzero((want != got).any(1))
    assert bad.size == 0, "%d of %d colours differ, first: want %s got %s (track %d label %d occ %d oof %d)" % (
        bad.size, n, want[bad[0]], got[bad[0]], pts["track"][bad[0]], pts["label"][bad[0]], occ[bad[0]], oof[bad[0]])
    # every branch was exercised
    assert (occ == 2).any() and ((occ == 1) & (pts["label"] == 0)).any() and oof.any() and (~oof).any()
    assert ((pts["track"] > cfg["max_movable_track"]) & (pts["label"] != 0)).any() and (pts["track"] <= cfg["max_movable_track"]).any()
    free, nf = g.occupied_rgb(free=True)
    assert nf > 0 and np.all(free["g"] == 255) and np.all(free["r"] == 0) and np.all(free["b"] == 0)
    g.close()
