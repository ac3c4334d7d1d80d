"""Generates the golden fixtures under tests/golden/ from the CPU oracle (canonical bin order).

Run from the repository root:  python tests/golden/make_golden.py
The reference ships no golden vectors, so these pin the oracle against regressions and give the
GPU tests expected outputs that do not need the oracle at run time.  Inputs come from the deterministic
generator semantic_dsp_map_amd/synth.py; a checksum of the inputs is stored next to the expected outputs.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402
from semantic_dsp_map_amd import synth  # noqa: E402

CASES = {
    "t0_vkitti2_dyn3": ("T0", "vkitti2", 6, dict(n_dynamic=3)),
    "t1_zed2_dyn2": ("T1", "zed2", 5, dict(n_dynamic=2)),
    "t0_noisy3_dyn2": ("T0", "noisy3", 5, dict(n_dynamic=2)),
    "t0_nodepthnoise": ("T0", "nodepthnoise", 4, dict(n_dynamic=2)),
    "t0_kitti360_static": ("T0", "kitti360", 4, dict(n_dynamic=0)),
}


def input_digest(frames):
    h = hashlib.sha256()
    for depth, cloud, pos, q, moves in frames:
        for a in (depth, cloud, pos, q, moves):
            h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def run_case(name):
    cfg_name, params_name, n, kw = CASES[name]
    cfg, params, frames = synth.make_frames(cfg_name, n, params_name, **kw)
    o = orc.OracleMap(dict(cfg, bin_order=1), params, synth.noise_table())
    n_vis = []
    for depth, cloud, pos, q, moves in frames:
        o.update(depth, cloud, pos, q, moves)
        n_vis.append(o.stats()["n_visible"])
    st = o.dump_state()
    vox = o.voxels()
    return cfg, params, frames, dict(
        input_sha256=np.frombuffer(bytes.fromhex(input_digest(frames)), np.uint8),
        n_visible=np.array(n_vis, np.int64),
        occ=vox["occ"], label=vox["label"], track=vox["track"], wsum=vox["wsum"],
        status=st["status"], ts=st["ts"], ptrack=st["track"], plabel=st["label"], forget=st["forget"],
        owner=st["owner"], w=st["w"], px=st["px"], py=st["py"], pz=st["pz"],
        ring=np.array([o.ring_state()[k] for k in ("global_time_stamp", "birth_cursor", "move_cursor")], np.int64))


if __name__ == "__main__":
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name in CASES:
        _, _, _, data = run_case(name)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **data)
        print(name, int((data["occ"] > 0).sum()), "occupied voxels,", int((data["status"] != 0).sum()), "slots set")
