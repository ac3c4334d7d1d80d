"""Generator of tests/golden/dense_sweep.npz: the oracle's voxel results and weights after ONE frame on the random,
densely populated state of tests/dense_state.py (T0 grid, 8 slots per voxel, seed 1, a few re-stamped slabs).  The frame's
sweep is the non-incremental one, over chunks of every density, with clamps, culls, guessed births, stale slots and
ties between tracks.

    python -m tests.golden.make_golden_dense      (re-)writes the fixture from the oracle
"""
import hashlib
import os

import numpy as np

from semantic_dsp_map_amd import synth
from tests.dense_state import random_state, stamps_for

HERE = os.path.dirname(os.path.abspath(__file__))
CFG_NAME, PARAMS_NAME, P_N, SEED = "T0", "vkitti2", 3, 1


def inputs():
    cfg = dict(synth.CONFIGS[CFG_NAME], p_n=P_N)
    params = synth.PARAMS[PARAMS_NAME]
    _, _, frames = synth.make_frames(CFG_NAME, 1, PARAMS_NAME, n_dynamic=0)
    st = random_state(cfg, SEED)
    h = hashlib.sha256()
    for k in sorted(st):
        h.update(np.ascontiguousarray(st[k]).tobytes())
    for a in frames[0][:2]:
        h.update(np.ascontiguousarray(a).tobytes())
    return cfg, params, frames[0], st, np.frombuffer(h.digest(), np.uint8)


def run(m, frame, st):
    """Load the state into map m (oracle or HIP binding: same interface), run the frame, return what the fixture holds."""
    (sx, sy, sz), ring = stamps_for(m)
    m.load_state(st)
    m.set_stamps(sx, sy, sz)
    m.set_ring_state(ring)
    depth, cloud, pos, q, moves = frame
    m.update(depth, cloud, pos, q, moves)
    vox, state = m.voxels(), m.dump_state()
    return {"occ": vox["occ"], "label": vox["label"], "track": vox["track"], "wsum": vox["wsum"],
            "w": state["w"], "status": state["status"]}


def main():
    from oracle import oracle as orc
    cfg, params, frame, st, digest = inputs()
    o = orc.OracleMap(dict(cfg, bin_order=1), params, synth.noise_table())
    data = run(o, frame, st)
    data["input_sha256"] = digest
    np.savez_compressed(os.path.join(HERE, "dense_sweep.npz"), **data)
    print("dense_sweep.npz:", {k: (v.shape, str(v.dtype)) for k, v in data.items()},
          "occupied %d, guessed %d, empty %d, unobserved %d" % ((data["occ"] == 1).sum(), (data["occ"] == 2).sum(),
                                                                (data["occ"] == 0).sum(), (data["occ"] < 0).sum()))


if __name__ == "__main__":
    main()
