"""Development aid (library built with tools/ab_build.sh timers_birth -DSDM_AB_TIMERS=1 -DSDM_TIMERS_BIRTH=1): the first head lane
of every workgroup of k_birth_replay - the voxel's rows and the segment's keys have arrived, the closed form and the resampling
are done, the candidates' positions have arrived, end (the stores are issued, not waited for)."""
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
buf = np.zeros(6 * 8192 * 4 + 4 * 4096 * 4, np.uint64)
for t in range(10):
    depth, cloud, pos, q = scene.render(t, params)
    L.sdm_debug_timers(m.h, buf.ctypes.data, 1)
    m.update(depth, cloud, pos, q, scene.moves(t), sync=True)
    L.sdm_debug_timers(m.h, buf.ctypes.data, 0)
    if t < 7:
        continue
    b = buf[:6 * 8192 * 4].astype(np.int64).reshape(6, 8192, 4)[0]
    ran = (b[:, 0] > 0) & (b[:, 3] > 0) & (b[:, 1] > 0) & (b[:, 2] > 0)
    a = b[ran] / 100.0
    t0 = a[:, 0].min()
    dur = a[:, 3] - a[:, 0]
    order = np.argsort(dur)
    def row(sel, name):
        x = a[sel]
        print("   %-22s n %4d: rows arrived at +%.1f | closed form + resampling %.1f | positions arrived %.1f | stores issued %.1f | total from the rows %.1f us (means)"
              % (name, len(x), (x[:, 0] - t0).mean(), (x[:, 1] - x[:, 0]).mean(), (x[:, 2] - x[:, 1]).mean(), (x[:, 3] - x[:, 2]).mean(), (x[:, 3] - x[:, 0]).mean()))
    print("frame %d: %d workgroups with heads; last end %.1f us after the first start" % (t, ran.sum(), a[:, 3].max() - t0))
    row(order[: len(order) // 2], "faster half")
    row(order[len(order) // 2: -len(order) // 10], "50th-90th percentile")
    row(order[-len(order) // 10:], "slowest tenth")
    row(order[-5:], "slowest five")
m.close()
