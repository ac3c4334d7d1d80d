"""Debug: first frame where the GPU results differ from the oracle in the fuzz test, with details."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from semantic_dsp_map_amd import synth
from tests import parity_utils as pu
from tests.test_fuzz_gpu import random_frame

params_name, seed, p_n = "vkitti2", 1, 3
cfg = dict(synth.CONFIGS["T0"], p_n=p_n)
params = synth.PARAMS[params_name]
rng = np.random.default_rng(seed)
o, g = pu.make_pair(cfg, params, synth.noise_table())
S = 1 << p_n
pos = np.zeros(3); yaw = 0.0
prev_vg = None
for t in range(12):
    pos = pos + rng.normal(0, 0.35, 3) * np.array([1.0, 0.2, 1.0])
    yaw += rng.normal(0, 0.08)
    depth, cloud, mv, remove = random_frame(rng, cfg, params, t, pos, yaw)
    q = synth.yaw_quat(yaw).astype(np.float32); p32 = pos.astype(np.float32)
    o.update(depth, cloud, p32, q, mv, remove)
    g.update(depth, cloud, p32, q, mv, remove)
    g.synchronize()
    vo, vg = o.voxels(), g.voxels()
    bad = np.flatnonzero(vo["wsum"].view(np.uint32) != vg["wsum"].view(np.uint32))
    st = g.stats()
    print("frame", t, "bad", bad.size, "sweep_live", st["sweep_live_voxels"], "tiles", st["sweep_tiles"], "stamps", [int(x.max()) for x in g.stamps()])
    import ctypes as C
    from semantic_dsp_map_amd import binding
    L = binding.load_library()
    if hasattr(L, "sdm_debug_counters"):
        arr = (C.c_uint32 * 7)()
        L.sdm_debug_counters(g.h, arr)
        print("   dbg", [hex(x) for x in arr], "lv", arr[1])
    if bad.size:
        so = o.dump_state()
        sgd = g.dump_state()
        print(" gpu vts of bad:", sgd["ts"][bad * S][:40], " oracle:", so["ts"][bad * S][:40])
        for k in ("ts", "status", "w", "track"):
            print("  state diff", k, int((so[k] != sgd[k]).sum()))
        sx, sy, sz = o.stamps()
        for v in bad[:24]:
            rx, ry, rz = v & 31, (v >> 5) & 31, v >> 10
            smax = max(sx[rx], sy[ry], sz[rz])
            sl = slice(v * S, v * S + S)
            print(" v", v, (rx, ry, rz), "o", vo["wsum"][v], "g", vg["wsum"][v], "prev g", None if prev_vg is None else prev_vg["wsum"][v],
                  "smax", smax, "ts", so["ts"][sl], "st", so["status"][sl])
        print(" bad voxels", bad[:64])
        break
    prev_vg = vg
