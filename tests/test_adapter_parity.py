"""The drop-in class itself against the oracle: SemanticDSPMap::update (include/semantic_dsp_map.h on libsdm_hip; reference
semantic_dsp_map.h:170-251) takes two committed clips frame by frame, and the clouds it appends to - occupied and free
voxels, xyz + rgb as pcl::PointXYZRGB, storage order - must equal, byte for byte, what tests/adapter_model.py (oracle.py +
object_layer.py + colour.py) produced for the same inputs and committed in tests/golden/adapter_clips.npz.

What only this path exercises: packRawInputs (mask layout, the ZED2 per-object boxes), the label -> instance tables, track-id
re-allocation above g_max_movable_object_instance_id (:179-186), the built-in object layer fed with the map's own owner-set
keys (sdm_tracks_with_particles; :712-736), BOOST-mode inputs reduced on the device, both colour formats."""
import json
import os
import subprocess

import numpy as np
import pytest

from semantic_dsp_map_amd import binding
from tests import adapter_clip, adapter_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "adapter_clips.npz")
CLIPS = ["vk2", "zed2b"]


def load(name):
    z = np.load(GOLDEN)
    preset = json.loads(str(z[name + "_preset"]))
    params = json.loads(str(z[name + "_params"]))
    frames = adapter_clip.frames_of(z, name)
    want = [(z["%s_%d_occ" % (name, t)], z["%s_%d_free" % (name, t)]) for t in range(len(frames))]
    return z, preset, params, tuple(z[name + "_bayes"]), bool(z[name + "_evaluation_format"]), frames, want


@pytest.mark.parametrize("name", CLIPS)
def test_fixture_is_what_the_oracle_model_emits(name):
    """(no GPU) the committed clouds are reproduced from the committed inputs by the oracle-side model: the fixture cannot
    drift away from the oracle unnoticed"""
    z, preset, params, bayes, evf, frames, want = load(name)
    model = adapter_model.AdapterModel(preset, params, z["noise"], bayes=bayes, evaluation_format=evf)
    n_occ = 0
    for t, fr in enumerate(frames):
        occ, free = model.update(fr["depth"], fr["seg"], fr["pos"], fr["q"], get_freespace=fr["free"], time_stamp=fr["ts"])
        assert occ.tobytes() == want[t][0].tobytes(), "%s frame %d: occupied cloud" % (name, t)
        assert (free.tobytes() if free is not None else b"") == want[t][1].tobytes(), "%s frame %d: free cloud" % (name, t)
        n_occ += len(occ)
    assert n_occ > 500
    if name == "vk2":  # the re-allocated id (:179-186) owns particles under its new number
        assert 3 in model.tracks_with_particles() and (3 + adapter_model.MAX_MOVABLE) in [s["track_id"] for s in frames[0]["seg"]]


def build_exe(exe):
    csrc = os.path.dirname(binding.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "tests", "mock_includes"), "-I",
                           os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "adapter_parity.cpp"), "-o", exe, "-L", csrc,
                           "-lsdm_hip", "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib"])


def test_parity_driver_builds(tmp_path):
    build_exe(str(tmp_path / "adapter_parity"))


@pytest.mark.gpu
@pytest.mark.parametrize("name", CLIPS)
def test_update_emits_the_oracles_clouds_byte_for_byte(name, tmp_path):
    z, preset, params, bayes, evf, frames, want = load(name)
    exe, clip, out = str(tmp_path / "adapter_parity"), str(tmp_path / "clip.bin"), str(tmp_path / "out.bin")
    build_exe(exe)
    adapter_clip.write_binary(clip, preset, params, bayes, z["noise"], frames, evf)
    r = subprocess.run([exe, clip, out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "adapter parity clip done" in r.stdout, r.stdout + r.stderr
    raw = open(out, "rb").read()
    at = 0
    rec = binding.POINT_XYZRGB.itemsize
    for t in range(len(frames)):
        for which, label in ((0, "occupied"), (1, "free")):
            n = int(np.frombuffer(raw, "<u4", 1, at)[0])
            at += 4
            got = np.frombuffer(raw, binding.POINT_XYZRGB, n, at)
            at += n * rec
            exp = want[t][which]
            assert n == len(exp), "%s frame %d: %d %s voxels, the oracle has %d" % (name, t, n, label, len(exp))
            if got.tobytes() != exp.tobytes():
                bad = np.flatnonzero([got[i].tobytes() != exp[i].tobytes() for i in range(n)])
                raise AssertionError("%s frame %d: %d of %d %s points differ, first [%d]: got %r, want %r"
                                     % (name, t, bad.size, n, label, bad[0], got[bad[0]], exp[bad[0]]))
    assert at == len(raw)
