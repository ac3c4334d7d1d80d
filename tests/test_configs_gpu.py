"""BASELINE.json's configurations on the GPU (C1 .. C5), each at its full size.

  C1  64^3, static mode (kitti360 parameters, no instances), map prefilled to 100 k particles  - bit-exact vs the oracle
  C2  128^3, 4 slots, ZED2 parameters, BOOST window, prefilled to 500 k particles              - bit-exact vs the oracle
  C4  256^3, 8 M particles, Z-slab shards G = 2 / 4 / 8 side by side on one GPU (the test plays the three all-gathers)
      - the union of the shards bit-exact vs the single-map oracle
  C5  512^3 (2^30 slots), 30-frame stream: the replay harness (page-locked buffers, frames issued back to back) against
      the binding frame by frame, and the incremental sweeps of 30 frames against one non-incremental sweep at the end
      (the oracle would need minutes per frame and 26 GB at this size: size-independent properties instead)
C3 is the benchmark configuration: test_parity_gpu.py::test_mid_size_preset_clip, test_sharded_two_process_gpu.py, bench.py.
"""
import re
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc
from semantic_dsp_map_amd import binding, synth
from tests import parity_utils as pu

pytestmark = pytest.mark.gpu


def prefilled_pair(cfg_name, params_name, n_particles, scene_kw):
    cfg = synth.CONFIGS[cfg_name]
    params = synth.PARAMS[params_name]
    scene = synth.Scene(cfg, **scene_kw)
    noise = synth.noise_table()
    o, g = pu.make_pair(cfg, params, noise)
    st, ring, n_pre = synth.prefill_state(cfg, scene, n_particles)
    for m in (o, g):
        m.load_state(st)
        m.set_ring_state(ring)
    return cfg, params, scene, o, g, n_pre


def run_and_compare(cfg, params, scene, o, g, n_frames, moves=True):
    S = 1 << cfg["p_n"]
    for t in range(n_frames):
        depth, cloud, pos, q = scene.render(t, params)
        mv = scene.moves(t) if moves else None
        o.update(depth, cloud, pos, q, mv)
        g.update(depth, cloud, pos, q, mv)
        g.synchronize()
        rep = pu.compare_maps(o, g, S, tag="frame %d: " % t)
        assert not rep, "\n".join(rep)


def test_C1_static_64cube_100k_particles():
    cfg, params, scene, o, g, n_pre = prefilled_pair("C1", "kitti360", 100000, dict(n_static=16, n_dynamic=0))
    assert n_pre >= 90000
    run_and_compare(cfg, params, scene, o, g, 4, moves=False)
    st = g.stats(count_live=True)
    assert st["live_particles"] >= 80000 and st["n_visible"] > 0
    g.close()


def test_C2_zed2_128cube_500k_particles():
    cfg, params, scene, o, g, n_pre = prefilled_pair("C2", "zed2", 500000, dict(n_static=24, n_dynamic=3))
    assert n_pre >= 450000
    run_and_compare(cfg, params, scene, o, g, 4)
    st = g.stats(count_live=True)
    assert st["live_particles"] >= 400000 and st["n_visible"] > 0
    g.close()


@pytest.mark.parametrize("G", [2, 4, 8])
def test_C4_256cube_8M_particles_zslab_shards(G):
    from tests.test_sharded_gpu import Shard, run_frame, compare_union
    cfg = synth.CONFIGS["C4"]
    params = synth.PARAMS["vkitti2"]
    scene = synth.Scene(cfg, n_static=48, n_dynamic=6, seed=7, dyn_speed=(0.6, 1.2))
    noise = synth.noise_table()
    parts = [synth.prefill_state(cfg, scene, 8000000 // G, shard_rank=r, shard_count=G) for r in range(G)]
    ring = parts[0][1]
    assert sum(p[2] for p in parts) >= 7000000
    shards = [Shard(cfg, params, noise, r, G) for r in range(G)]
    for s, (st, _, _) in zip(shards, parts):
        s.m.load_state(st)
        s.m.set_ring_state(ring)
    o = orc.OracleMap(dict(cfg, bin_order=1, ck_slabs=G), params, noise)
    o.load_state({k: np.concatenate([p[0][k] for p in parts]) for k in parts[0][0]})
    o.set_ring_state(ring)
    del parts
    exported = 0
    n_frames = 3
    for t in range(n_frames):
        depth, cloud, pos, q = scene.render(t, params)
        frame = (depth, cloud, pos, q, scene.moves(t))
        o.update(*frame)
        exported += run_frame(shards, frame)
    compare_union(o, shards, n_frames - 1, 1 << cfg["p_n"])
    assert exported > 0, "no particle crossed a slab border"
    assert sum(s.m.stats(count_live=True)["live_particles"] for s in shards) >= 6500000
    for s in shards:
        s.m.close()


def wordsum(vox):
    return int(np.frombuffer(vox.tobytes(), "<u8").sum(dtype=np.uint64))


def test_C5_512cube_30_frame_stream(tmp_path):
    from tests.test_replay import build, EXE
    build()
    clip = str(tmp_path / "c5.bin")
    cfg, params, noise, frames = synth.write_clip(clip, "C5", 30, "vkitti2", n_static=48, n_dynamic=6)
    r = subprocess.run([EXE, clip, "1", "pinned", "pipelined"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"occupied (\d+)\s+checksum ([0-9a-f]{16})\s+wordsum ([0-9a-f]{16})", r.stdout)
    assert m, r.stdout
    g = binding.SdmMap(cfg, params, noise)
    for depth, static_mask, objects, pos64, q64, moves in frames:
        g.update_raw(depth, static_mask, synth.LABEL_TO_STATIC_INSTANCE, objects, pos64, q64, moves, sync=True)
    vox = g.voxels()
    n_occ = int((vox["occ"] > 0).sum())
    assert n_occ > 1000 and n_occ == int(m.group(1))
    assert wordsum(vox) == int(m.group(3), 16)                 # streamed back to back == frame by frame
    assert set(np.unique(vox["occ"]).tolist()) <= {-1, 0, 1, 2}
    # 30 incremental sweeps == a non-incremental sweep in frame 30: a second map runs the same clip and is told before
    # the last frame that no stored result is safe (sdm_set_params does that), so its last sweep evaluates every voxel
    # the way the reference does in every frame.  (A full sweep AFTER the clip would not do: evaluating a voxel twice
    # without a frame in between is not idempotent, clamped weights and culled slots change the second answer.)
    g2 = binding.SdmMap(cfg, params, noise)
    for k, (depth, static_mask, objects, pos64, q64, moves) in enumerate(frames):
        if k == len(frames) - 1:
            g2.set_params(params)
        g2.update_raw(depth, static_mask, synth.LABEL_TO_STATIC_INSTANCE, objects, pos64, q64, moves)
    g2.synchronize()
    assert g2.stats()["sweep_tiles"] == (1 << 27) // 2048          # every tile was looked into
    vox2 = g2.voxels()
    diff = np.flatnonzero(vox.view("<u8") != vox2.view("<u8"))
    assert diff.size == 0, "%d voxels differ between the incremental and the non-incremental sweep" % diff.size
    g2.close()
    st = g.stats(count_live=True)
    assert st["live_particles"] > 10000
    g.close()
