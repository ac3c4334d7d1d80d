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
out = np.zeros((3, 4096, 4), np.uint64)   # k_birth_replay, k_move_apply, k_move_replay: [workgroup][checkpoint]
us = lambda x: x / 100.0
for t in range(12):
    depth, cloud, pos, q = scene.render(t, params)
    L.sdm_debug_timers(m.h, out.ctypes.data, 1)
    m.update(depth, cloud, pos, q, scene.moves(t), sync=True)
    L.sdm_debug_timers(m.h, out.ctypes.data, 0)
    o = out.astype(np.int64)
    b = o[0]
    ran = b[:, 0] > 0
    rep = ran & (b[:, 2] > 0)
    if rep.any():
        t0 = b[ran, 0].min()
        print("frame %2d birth_replay: %d workgroups, %d with heads: compaction %.1f us (max), replay avg %.1f / max %.1f us, ends at %.1f us"
              % (t, ran.sum(), rep.sum(), us((b[rep, 1] - b[rep, 0]).max()), us((b[rep, 2] - b[rep, 1]).mean()), us((b[rep, 2] - b[rep, 1]).max()),
                 us(b[rep, 2].max() - t0)))
    a = o[1]
    act = (a[:, 0] > 0) & (a[:, 3] > 0)
    if act.any():
        t0 = a[a[:, 0] > 0, 0].min()
        print("         move_apply: %d workgroups with members: ranking avg %.1f / max %.1f us, moves avg %.1f / max %.1f us, flush max %.1f us, ends at %.1f us"
              % (act.sum(), us((a[act, 1] - a[act, 0]).mean()), us((a[act, 1] - a[act, 0]).max()), us((a[act, 2] - a[act, 1]).mean()),
                 us((a[act, 2] - a[act, 1]).max()), us((a[act, 3] - a[act, 2]).max()), us(a[act, 3].max() - t0)))
    r = o[2]
    act = (r[:, 0] > 0) & (r[:, 2] > 0)
    if act.any():
        t0 = r[act, 0].min()
        print("         move_replay (thread 0 of each workgroup, its last voxel): %d workgroups: rows + list walk avg %.1f / max %.1f us, inserts avg %.1f / max %.1f us (max %d copies), ends at %.1f us"
              % (act.sum(), us((r[act, 1] - r[act, 0]).mean()), us((r[act, 1] - r[act, 0]).max()), us((r[act, 2] - r[act, 1]).mean()),
                 us((r[act, 2] - r[act, 1]).max()), r[act, 3].max(), us(r[act, 2].max() - t0)))
