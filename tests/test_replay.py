"""the ROS-free replay harness (tools/replay/replay.cpp, plain C ABI) streams a dumped clip through
sdm_update_raw; its result must equal what the Python binding produces from the same frames."""
import os
import re
import subprocess

import pytest

from semantic_dsp_map_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "replay", "replay.cpp")
EXE = os.path.join(ROOT, "tools", "replay", "replay")
LIBDIR = os.path.join(ROOT, "semantic_dsp_map_amd", "csrc")


def build():
    lib = os.path.join(LIBDIR, "libsdm_hip.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", LIBDIR])
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(SRC), os.path.getmtime(lib)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), SRC, "-o", EXE, lib,
                               "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"])


def fnv1a(b):
    h = 1469598103934665603
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_harness_builds_and_clip_round_trips(tmp_path):
    build()
    clip = str(tmp_path / "clip.bin")
    cfg, params, noise, frames = synth.write_clip(clip, "T0", 2, "vkitti2", n_dynamic=2)
    W, H = cfg["width"], cfg["height"]
    raw = open(clip, "rb").read()
    assert raw[:8] == b"SDMCLIP1"
    per_frame = 56 + 12 + 4 * W * H + W * H + sum(8 + W * H for _ in frames[0][2]) + frames[0][5].size * 68
    assert len(raw) == 8 + 80 + 52 + 4 + 4 * noise.size + 512 + 4 + sum(
        56 + 12 + 5 * W * H + len(fr[2]) * (8 + W * H) + fr[5].size * 68 for fr in frames)
    assert per_frame > 0
    # without a device the harness must fail loudly, not fall back to anything
    r = subprocess.run([EXE, clip], capture_output=True, text=True)
    if r.returncode != 0:
        assert "sdm_create" in r.stderr


@pytest.mark.gpu
def test_replay_matches_binding(tmp_path):
    from semantic_dsp_map_amd import binding
    build()
    clip = str(tmp_path / "clip.bin")
    cfg, params, noise, frames = synth.write_clip(clip, "T0", 5, "vkitti2", n_dynamic=3)
    r = subprocess.run([EXE, clip], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"occupied (\d+)\s+checksum ([0-9a-f]{16})", r.stdout)
    assert m, r.stdout
    g = binding.SdmMap(cfg, params, noise)
    for depth, static_mask, objects, pos64, q64, moves in frames:
        g.update_raw(depth, static_mask, synth.LABEL_TO_STATIC_INSTANCE, objects, pos64, q64, moves, sync=True)
    vox = g.voxels()
    assert int(m.group(1)) == int((vox["occ"] > 0).sum()) and int(m.group(1)) > 0
    assert int(m.group(2), 16) == fnv1a(vox.tobytes())
    # three passes with sdm_clear in between (the ring offset and the noise cursors survive a clear, operations.h:683,
    # so the passes differ), page-locked buffers, frames issued back to back: same as the binding doing the same
    r2 = subprocess.run([EXE, clip, "3", "pinned", "pipelined"], capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stdout + r2.stderr
    m2 = re.search(r"occupied (\d+)\s+checksum ([0-9a-f]{16})", r2.stdout)
    assert m2, r2.stdout
    for rep in range(2):
        g.clear()
        for depth, static_mask, objects, pos64, q64, moves in frames:
            g.update_raw(depth, static_mask, synth.LABEL_TO_STATIC_INSTANCE, objects, pos64, q64, moves)
    g.synchronize()
    assert int(m2.group(2), 16) == fnv1a(g.voxels().tobytes())
    g.close()
