"""What comes out of a drive through a street from an EMPTY map: live / visible particles and the frame time every 20
frames (GPU only).  usage: driven_probe.py <n_frames> '<json scene kwargs>'"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from semantic_dsp_map_amd import binding, synth  # noqa: E402

def main():
    n = int(sys.argv[1])
    kw = json.loads(sys.argv[2]) if len(sys.argv) > 2 else {}
    if "lateral_extra" in kw:
        kw["lateral_extra"] = tuple(kw["lateral_extra"])
    if "dyn_speed" in kw:
        kw["dyn_speed"] = tuple(kw["dyn_speed"])
    cfg, params = synth.CONFIGS["C3"], synth.PARAMS["vkitti2_nb3"]
    scene = synth.Scene(cfg, **kw)
    t0 = time.time()
    rendered = synth.render_frames(cfg, params, kw, range(n))
    print("rendered %d frames in %.1f s" % (n, time.time() - t0), flush=True)
    m = binding.SdmMap(cfg, params, None, device=0)
    m.generate_noise_table(seed=20250217)
    for t in range(n):
        depth, cloud, pos, q = rendered[t]
        t0 = time.perf_counter()
        m.update(depth, cloud, pos, q, scene.moves(t), sync=True)
        dt = time.perf_counter() - t0
        if t % 20 == 19 or t == n - 1:
            st = m.stats(count_live=True)
            print("frame %3d: live %7d particles in %6d voxels, visible %6d, births %6d, moved %6d (re-inserted %6d), alias %5d, valid px %6d, %.2f ms (host-synchronised)" % (
                t, st["live_particles"], st["live_voxels"], st["n_visible"], st.get("n_birth_success", -1), st.get("n_moved", -1),
                st.get("n_move_reinserted", -1), st.get("alias_entries", -1), int(cloud["is_valid"].sum()), dt * 1e3), flush=True)


if __name__ == "__main__":  # (render_frames spawns worker processes that import this module)
    main()
