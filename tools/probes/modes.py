"""Development aid: is the frame-time mode (0.27 / 0.295 ms on C3) a property of the process or of the map's streams?
Creates the C3 benchmark map several times in ONE process and times 20 back-to-back frames on each."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from semantic_dsp_map_amd import sharded, synth  # noqa: E402

cfg, params = synth.CONFIGS["C3"], synth.PARAMS["vkitti2"]
scene = synth.Scene(cfg, n_static=48, n_dynamic=6, seed=7)
st, ring, _ = synth.prefill_state(cfg, scene, 2000000)
host = [scene.render(t, params) + (scene.moves(t),) for t in range(25)]
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    eng = sharded.NativeShardedMap(cfg, params, 0, 1, 0)
    m = eng.map
    m.generate_noise_table(seed=20250217)
    frames = [(m.device_put(h[0]), m.device_put(h[1]), h[2], h[3], h[4]) for h in host]
    m.load_state(st)
    m.set_ring_state(ring)
    for t in range(5):
        eng.update(*frames[t])
    m.synchronize()
    m.device_synchronize()
    t0 = time.perf_counter()
    for t in range(5, 25):
        eng.update(*frames[t])
    m.synchronize()
    m.device_synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    import ctypes as C
    ov = []
    if hasattr(m.L, "sdm_debug_overlap"):
        m.L.sdm_debug_overlap.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
        for which in range(3):
            out = (C.c_double * 2)()
            m.L.sdm_debug_overlap(m.h, which, out)
            ov.append("%s %.0f/%.0f" % (("frustum", "birth", "moves")[which], out[0], out[1]))
    print("map %d: %.4f ms/frame   stamp on the side stream after .. us of a .. us spin on the main stream: %s" % (rep, ms, ", ".join(ov)), flush=True)
    m.close()
