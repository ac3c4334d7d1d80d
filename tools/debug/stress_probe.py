import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from semantic_dsp_map_amd import binding, synth
variants = {
 "a": (dict(n_static=120, n_dynamic=12, seed=11, yaw_rate_deg=0.5, lateral_extra=(0, 0.1)), "vkitti2_nb3"),
 "b": (dict(n_static=120, n_dynamic=12, seed=11, yaw_rate_deg=0.5, lateral_extra=(0, 0.1), speed=0.1), "vkitti2_nb3"),
 "c": (dict(n_static=200, n_dynamic=12, seed=11, yaw_rate_deg=0.5, lateral_extra=(0, 0.05), speed=0.1), "noisy3"),
}
cfg = synth.CONFIGS["C3"]
for name in sys.argv[1:]:
    kw, pname = variants[name]
    params = synth.PARAMS[pname]
    m = binding.SdmMap(cfg, params, None, device=0)
    m.generate_noise_table(seed=20250217)
    sc = synth.Scene(cfg, **kw)
    vis = []
    for t in range(30):
        depth, cloud, pos, q = sc.render(t, params)
        m.update(depth, cloud, pos, q, sc.moves(t))
        if t % 5 == 4:
            m.synchronize(); s = m.stats(count_live=True)
            vis.append((t, s["n_visible"], s["live_particles"], s["n_birth_success"], s["sweep_live_voxels"]))
    print(name, vis)
    m.close()
