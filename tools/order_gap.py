#!/usr/bin/env python
"""Canonical vs literal summation order at benchmark size (CPU only, oracle in both modes, free running).

bin_order 0 = the reference's literal order (one running ck sum over each pixel's bin in BFS push order,
semantic_dsp_map.h:1029, mc_ring/operations.h:1405-1407), bin_order 1 = the canonical order the HIP path implements
(bins in ascending particle index, per-window-row partial sums).  Both maps start from the same prefilled state and
see the same frames; after every frame: slots whose status / time stamp differ, voxels whose occupancy code / label
differ, max |dw| over slots live in both, max relative d(ck+kappa) over valid pixels.  Writes a JSON summary.

usage: python tools/order_gap.py C2|C3 [n_frames] [out.json]"""
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
    cfg = synth.CONFIGS[name]
    params = synth.PARAMS[synth.CONFIG_PARAMS[name]]
    n_particles = {"C2": 500000, "C3": 2000000}.get(name, 100000)
    scene = synth.Scene(cfg, n_static=48 if name == "C3" else 24, n_dynamic=6 if name == "C3" else 3, seed=7)
    noise = synth.noise_table()
    st, ring, n_pre = synth.prefill_state(cfg, scene, n_particles)
    maps = []
    for order in (0, 1):
        o = orc.OracleMap(dict(cfg, bin_order=order), params, noise)
        o.load_state(st)
        o.set_ring_state(ring)
        maps.append(o)
    del st
    a, b = maps
    rows = []
    for t in range(n_frames):
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
