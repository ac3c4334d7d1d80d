"""GPU parity on the grids the reference actually ships (settings/settings.h:32-143, one per SETTING value) - the
presets include/semantic_dsp_map.h hands a drop-in user - each through the entry point the adapter uses for it
(sdm_update_raw_ex: depth + MONO8 masks in, generateLabeledPointCloud on the device), 6 frames against the oracle,
bit for bit on every state field, the ring state and the voxel results:

  SETTING 0  KITTI_360       256 x 256 x 256 x 8 slots, 0.15 m, 1408 x 376, static mode (no instances), options_kitti360.yaml
  SETTING 1  CODA            256 x 256 x 128 x 4 slots, 0.15 m,  960 x 540, depth <= 10 m, window 5, options.yaml
  SETTING 2  VIRTUAL_KITTI2  256 x 128 x 256 x 8 slots, 0.2 m,  1242 x 375, options_virtual_kitti2.yaml
  SETTING 3  ZED2            128 x  32 x 128 x 4 slots, 0.15 m, BOOST_MODE: 1280 x 720 inputs reduced to 640 x 360 by
                             manualResize, window 3, sky pixels dropped, object points clipped to their key-point boxes
                             (pointcloud_tools.h:101-196, 236-272), options_zed2.yaml
(BASELINE.json's cubes and the toy grids are covered in test_configs_gpu.py / test_parity_gpu.py.)"""
import numpy as np
import pytest

from oracle import oracle as orc
from semantic_dsp_map_amd import binding, synth
from tests import parity_utils as pu

pytestmark = pytest.mark.gpu
N_FRAMES = 6


def run_plain(cfg_name, scene_kw, instances):
    cfg = synth.CONFIGS[cfg_name]
    params = synth.PARAMS[synth.CONFIG_PARAMS[cfg_name]]
    sc = synth.Scene(cfg, **scene_kw)
    noise = synth.noise_table()
    o = orc.OracleMap(dict(cfg, bin_order=1), params, noise)
    g = binding.SdmMap(cfg, params, noise)
    S = 1 << cfg["p_n"]
    for t in range(N_FRAMES):
        depth, cloud, pos, q = sc.render(t, params)
        static_mask, objects = synth.raw_inputs(cfg, cloud, sc)
        pos64, q64 = sc.pose(t)
        moves = sc.moves(t) if instances else None
        want = o.generate_cloud(depth, static_mask, synth.LABEL_TO_STATIC_INSTANCE, objects, pos64, q64, consider_instance=instances)
        g.update_raw(depth, static_mask, synth.LABEL_TO_STATIC_INSTANCE, objects, pos64, q64, moves, sync=True,
                     flags=0 if instances else binding.NO_INSTANCES)
        got = g.labeled_cloud()
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), "LabeledPoint image differs at frame %d" % t
        o.update(depth, want, pos64.astype(np.float32), q64.astype(np.float32), moves)
        rep = pu.compare_maps(o, g, S, tag="%s frame %d: " % (cfg_name, t))
        assert not rep, "\n".join(rep)
    st = g.stats(count_live=True)
    assert st["n_visible"] > 0 and st["live_particles"] > 1000
    if instances:
        assert sum(g.object_particle_count(int(t)) for t in sc.dyn_tracks) > 0, "no particle is owned by a moving object"
    g.close()
    return st


def test_setting0_kitti360_256x256x256x8_static_mode():
    run_plain("REF_KITTI360", dict(n_static=32, n_dynamic=0), instances=False)


def test_setting1_coda_256x256x128x4():
    run_plain("REF_CODA", dict(n_static=32, n_dynamic=4, dyn_speed=(0.3, 0.8)), instances=True)


def test_setting2_virtual_kitti2_256x128x256x8():
    run_plain("REF_VKITTI2", dict(n_static=48, n_dynamic=6), instances=True)


def test_setting3_zed2_boost_128x32x128x4():
    """The shipped configuration: sensor-size inputs, BOOST reduction and the two ZED2 filters on the device."""
    cfg = synth.CONFIGS["REF_ZED2_BOOST"]
    sensor = synth.CONFIGS["REF_ZED2_SENSOR"]
    params = synth.PARAMS["zed2"]
    W, H = cfg["width"], cfg["height"]
    sw, sh = sensor["width"], sensor["height"]
    sc = synth.Scene(sensor, n_static=24, n_dynamic=3, dyn_speed=(0.3, 0.8))
    noise = synth.noise_table()
    o = orc.OracleMap(dict(cfg, bin_order=1), params, noise)
    g = binding.SdmMap(cfg, params, noise)
    # "Sky" of this test: the pole label's static instance (the street scene has no sky label) - its pixels must vanish
    sky = int(synth.LABEL_TO_STATIC_INSTANCE[synth.LABEL_POLE])
    clipped = 0
    for t in range(N_FRAMES):
        depth, cloud, pos, q = sc.render(t, params)                      # 1280 x 720
        static_mask, objects = synth.raw_inputs(sensor, cloud, sc)
        pos64, q64 = sc.pose(t)
        # boxes of the objects' current key points +- 1 m (pointcloud_tools.h:174-196); the first one is cut short in z so
        # that part of the object falls outside and turns into Background
        boxes = []
        for k, b in enumerate(sc.dyn_boxes(t)):
            zhi = b[5] + 1.0 if k else 0.5 * (b[2] + b[5])
            boxes.append([b[0] - 1.0, b[3] + 1.0, b[1] - 1.0, b[4] + 1.0, b[2] - 1.0, zhi])
        moves = sc.moves(t)
        want, dres = o.generate_cloud_ex(depth, static_mask, synth.LABEL_TO_STATIC_INSTANCE, objects, pos64, q64, src_size=(sw, sh),
                                         rescale=0.5, sky_instance=sky, object_bbox=boxes)
        g.update_raw(depth, static_mask, synth.LABEL_TO_STATIC_INSTANCE, objects, pos64, q64, moves, sync=True, src_size=(sw, sh),
                     rescale=0.5, sky_instance=sky, object_bbox=boxes)
        got = g.labeled_cloud()
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), "LabeledPoint image differs at frame %d" % t
        assert not (want["track_id"][want["is_valid"] > 0] == sky).any()
        clipped += int(((want["track_id"] == 65535) & (want["is_valid"] > 0)).sum())
        o.update(dres.reshape(H, W), want, pos64.astype(np.float32), q64.astype(np.float32), moves)
        rep = pu.compare_maps(o, g, 1 << cfg["p_n"], tag="ZED2 frame %d: " % t)
        assert not rep, "\n".join(rep)
    assert clipped > 0, "the bounding-box filter never fired"
    st = g.stats(count_live=True)
    assert st["n_visible"] > 0 and st["live_particles"] > 1000
    g.close()
