"""Development aid: in-kernel wall-clock checkpoints (library built with tools/ab_build.sh timers -DSDM_AB_TIMERS=1,
SDM_LIB_PATH=build/ab/libsdm_timers.so): runs the benchmark frames one by one and prints g_dbg after each."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from semantic_dsp_map_amd import binding, synth  # noqa: E402

cfg, params = synth.CONFIGS["C3"], synth.PARAMS["vkitti2"]
m = binding.SdmMap(cfg, params, None, device=0)
m.generate_noise_table()
scene = synth.Scene(cfg, n_static=48, n_dynamic=6, seed=7)
st, ring, _ = synth.prefill_state(cfg, scene, 2000000)
m.load_state(st)
m.set_ring_state(ring)
L = m.L
L.sdm_debug_timers.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
out = np.zeros((4096, 4), np.uint64)
for t in range(12):
    depth, cloud, pos, q = scene.render(t, params)
    L.sdm_debug_timers(m.h, out.ctypes.data, 1)
    m.update(depth, cloud, pos, q, scene.moves(t), sync=True)
    L.sdm_debug_timers(m.h, out.ctypes.data, 0)
    o = out.astype(np.int64)
    ran = o[:, 0] > 0
    t0 = o[ran, 0].min()
    rep = ran & (o[:, 2] > 0)
    print("frame %2d: %d workgroups, starts spread over %.1f us | %d with heads: compaction %.1f us (max), replay avg %.1f / max %.1f us, "
          "last one ends at %.1f us | inserts of lane 0: max %d"
          % (t, ran.sum(), (o[ran, 0].max() - t0) / 100.0, rep.sum(), ((o[rep, 1] - o[rep, 0]).max()) / 100.0,
             ((o[rep, 2] - o[rep, 1]).mean()) / 100.0, ((o[rep, 2] - o[rep, 1]).max()) / 100.0, (o[rep, 2].max() - t0) / 100.0, o[rep, 3].max()))
