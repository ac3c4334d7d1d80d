#!/usr/bin/env python
"""Canonical vs literal summation order at benchmark size (CPU only, oracle in both modes, free running).

bin_order 0 = the reference's literal order (one running ck sum over each pixel's bin in BFS push order,
semantic_dsp_map.h:1029, mc_ring/operations.h:1405-1407), bin_order 1 = the canonical order the HIP path implements
(bins in ascending particle index, per-window-row partial sums).  Both maps start from the same prefilled state and
see the same frames; after every frame: slots whose status / time stamp differ, voxels whose occupancy code / label
differ, max |dw| over slots live in both, max relative d(ck+kappa) over valid pixels.  Writes a JSON summary.

"stress" = bench.py's busy scene (C3 grid, 200 static + 12 moving boxes, three noisy births per point, ~51 k visible
particles per frame): one map in canonical order runs the 14 warm-up frames from the prefilled state, both orders continue
from its state.

usage: python tools/order_gap.py C2|C3|stress [n_frames] [out.json]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402
from semantic_dsp_map_amd import synth  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C2"
    n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    stress = name == "stress"
    cfg = synth.CONFIGS["C3" if stress else name]
    params = synth.PARAMS["vkitti2_nb3" if stress else synth.CONFIG_PARAMS[name]]
    n_particles = {"C2": 500000, "C3": 2000000, "stress": 2000000}.get(name, 100000)
    if stress:
        scene = synth.Scene(cfg, n_static=200, n_dynamic=12, seed=11, yaw_rate_deg=1.5, lateral_extra=(0, 0.04))
    else:
        scene = synth.Scene(cfg, n_static=48 if name == "C3" else 24, n_dynamic=6 if name == "C3" else 3, seed=7)
    noise = synth.noise_table()
    st, ring, n_pre = synth.prefill_state(cfg, scene, n_particles)
    n_warm = 14 if stress else 0
    stamps = None
    if n_warm:
        w = orc.OracleMap(dict(cfg, bin_order=1), params, noise)
        w.load_state(st)
        w.set_ring_state(ring)
        for t in range(n_warm):
            depth, cloud, pos, q = scene.render(t, params)
            w.update(depth, cloud, pos, q, scene.moves(t))
        st, ring, stamps = w.dump_state(), w.ring_state(), w.stamps()
        del w
    maps = []
    for order in (0, 1):
        o = orc.OracleMap(dict(cfg, bin_order=order), params, noise)
        o.load_state(st)
        if stamps is not None:
            o.set_stamps(*stamps)
        o.set_ring_state(ring)
        maps.append(o)
    del st
    a, b = maps
    rows = []
    for t in range(n_warm, n_warm + n_frames):
        depth, cloud, pos, q = scene.render(t, params)
        mv = scene.moves(t)
        a.update(depth, cloud, pos, q, mv)
        b.update(depth, cloud, pos, q, mv)
        ca, cb = a.ck_kappa(), b.ck_kappa()
        valid = cloud["is_valid"].reshape(ca.shape) > 0
        rel = float(np.max(np.abs(ca[valid] - cb[valid]) / np.maximum(np.abs(ca[valid]), 1e-30))) if valid.any() else 0.0
        sa, sb = a.dump_state(), b.dump_state()
        same = (sa["status"] == sb["status"]) & (sa["ts"] == sb["ts"])
        both = same & (sa["status"] != 0)
        dw = float(np.max(np.abs(sa["w"][both] - sb["w"][both]))) if both.any() else 0.0
        va, vb = a.voxels(), b.voxels()
        row = {"frame": t, "slots_status_or_ts_differ": int((~same).sum()), "voxels_occ_differ": int((va["occ"] != vb["occ"]).sum()),
               "voxels_label_differ": int((va["label"] != vb["label"]).sum()), "max_abs_dw": dw, "max_rel_dck": rel,
               "visible": int(a.stats()["n_visible"]), "bins_equal": bool(np.array_equal(a.bin_counts(), b.bin_counts()))}
        rows.append(row)
        print(json.dumps(row), flush=True)
        del sa, sb, va, vb
    out = {"config": name, "prefilled_particles": int(n_pre), "frames": rows,
           "worst": {k: max(r[k] for r in rows) for k in ("slots_status_or_ts_differ", "voxels_occ_differ", "voxels_label_differ",
                                                         "max_abs_dw", "max_rel_dck")}}
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out["worst"]))


if __name__ == "__main__":
    main()
