"""Literal (bin_order = 0) against canonical (bin_order = 1) summation order on the `driven` workload, CPU only: the oracle
drives synth.DRIVEN_FRAMES frames from an empty C3 map in canonical order; at frames 59, 119 and 214 a DEEP copy of it
(oracle_clone: the owner sets with their double memberships - a state dump carries one owner per index) continues in literal
order for 5 frames beside it, and the two are compared after every frame: slots whose status differs, voxels whose occupancy
code / label / track differ, max |difference of the weight sums|.  About ten minutes on eight cores (rendering included).
  python tools/order_gap_driven.py > profiles/r05_order_gap_driven.txt"""
import os
import pickle
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as orc
from semantic_dsp_map_amd import synth

def main():
    cfg = synth.CONFIGS["C3"]; params = synth.PARAMS[synth.DRIVEN_PARAMS]; kw = dict(synth.DRIVEN_SCENE)
    n = synth.DRIVEN_FRAMES
    scene = synth.Scene(cfg, **kw)
    t0 = time.time()
    if os.path.exists("/tmp/sdm_driven_frames.pkl"):
        rendered = pickle.load(open("/tmp/sdm_driven_frames.pkl", "rb"))
    else:
        rendered = synth.render_frames(cfg, params, kw, range(n), workers=8)
        pickle.dump(rendered, open("/tmp/sdm_driven_frames.pkl", "wb"), protocol=4)
    print("rendered", time.time() - t0, flush=True)
    noise = synth.noise_table()
    o = orc.OracleMap(dict(cfg, bin_order=1), params, noise)
    lit = None
    S = 8
    checkpoints = {59: None, 119: None, 214: None}
    for t in range(n):
        d, c, p, q = rendered[t]
        o.update(d, c, p, q, scene.moves(t))
        if lit is not None:
            lit.update(d, c, p, q, scene.moves(t))
            so, sl = o.dump_state(), lit.dump_state()
            live_o, live_l = so["status"] != 0, sl["status"] != 0
            nd = int(np.count_nonzero(so["status"] != sl["status"]))
            vo, vl = o.voxels(), lit.voxels()
            occd = int(np.count_nonzero(vo["occ"] != vl["occ"]))
            labd = int(np.count_nonzero((vo["label"] != vl["label"]) | (vo["track"] != vl["track"])))
            nocc = int(np.count_nonzero(vo["occ"] > 0))
            both = (vo["occ"] > -1) & (vl["occ"] > -1)
            dws = float(np.max(np.abs(vo["wsum"][both] - vl["wsum"][both])))
            # weights near the clamp
            w = so["w"][live_o]
            near = int(np.count_nonzero(np.abs(w - 1.0) < 1e-5)); exact = int(np.count_nonzero(w == 1.0))
            print("frame %d (since split %d): live %d/%d, status differs in %d slots; voxels: occ differs in %d of %d occupied, label/track in %d, max|dwsum| %.3g; weights ==1: %d, within 1e-5 of 1: %d, n_vis %d/%d"
                  % (t, t - split, live_o.sum(), live_l.sum(), nd, occd, nocc, labd, dws, exact, near, o.stats()["n_visible"], lit.stats()["n_visible"]), flush=True)
            if t - split >= 5:
                lit = None
        if t in checkpoints:
            lit = o.clone(bin_order=0)
            split = t
            print("split at", t, time.time() - t0, flush=True)

if __name__ == "__main__":
    main()
