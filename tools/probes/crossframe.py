"""Development aid (library built with -DSDM_AB_TIMERS=1): how long after the sweep of frame N does k_move_apply of frame
N+1 start when frames are issued back to back?  Frames 0..8 without a synchronisation, frame 9 stopped after its moves (so
that the sweep's clocks are still those of frame 8), then the in-kernel clocks are read.  Several maps per process."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from semantic_dsp_map_amd import sharded, synth  # noqa: E402

cfg, params = synth.CONFIGS["C3"], synth.PARAMS["vkitti2"]
scene = synth.Scene(cfg, n_static=48, n_dynamic=6, seed=7)
st, ring, _ = synth.prefill_state(cfg, scene, 2000000)
host = [scene.render(t, params) + (scene.moves(t),) for t in range(10)]
buf = np.zeros(6 * 8192 * 4 + 4 * 4096 * 4, np.uint64)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    eng = sharded.NativeShardedMap(cfg, params, 0, 1, 0)
    m = eng.map
    L = m.L
    L.sdm_debug_timers.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    m.generate_noise_table(seed=20250217)
    frames = [(m.device_put(h[0]), m.device_put(h[1]), h[2], h[3], h[4]) for h in host]
    m.load_state(st)
    m.set_ring_state(ring)
    for t in range(6):
        eng.update(*frames[t])
    m.synchronize()
    m.device_synchronize()
    import time
    tt = time.perf_counter()
    for rep2 in range(4):
        for t in range(1, 6):
            eng.update(*frames[t])
    m.synchronize()
    m.device_synchronize()
    ms = (time.perf_counter() - tt) / 20 * 1e3
    L.sdm_debug_timers(m.h, buf.ctypes.data, 1)
    import time
    t0 = time.perf_counter()
    for t in range(6, 9):
        eng.update(*frames[t])
    m.update(frames[9][0], frames[9][1], frames[9][2], frames[9][3], frames[9][4], stop_after=os.environ.get("CROSSFRAME_STOP", "move"), on_device=True)
    m.synchronize()
    m.device_synchronize()
    L.sdm_debug_timers(m.h, buf.ctypes.data, 0)
    k = buf[:6 * 8192 * 4].astype(np.int64).reshape(6, 8192, 4)
    mv = buf[6 * 8192 * 4:].astype(np.int64).reshape(4, 4096, 4)
    occ = k[5]
    occ_end = max(occ[:, 1].max(), occ[:, 3].max())
    occ_start = occ[occ[:, 0] > 0, 0].min()
    birth_end = k[0][:, 2].max()
    ap = mv[0]
    ap_start = ap[ap[:, 0] > 0, 0].min()
    def first(a, c=0):
        return a[a[:, c] > 0, c].min()
    rp_end = mv[1][:, 2].max()
    vis_start, vis_end = first(k[1]), k[1][:, 3].max()
    bsg_start = first(k[2])
    wt_end = k[4][:, 1].max()
    br_start = first(k[0])
    mm = mv[2]
    fb = mv[3]
    if (fb[:, 0] > 0).any():
        print("   k_frame_begin (9): first workgroup enters %.1f us after the sweep's end, last enters %.1f, duties done (last) %.1f"
              % ((first(fb) - occ_end) / 100.0, (fb[:, 0].max() - occ_end) / 100.0, (fb[:, 1].max() - occ_end) / 100.0), flush=True)
    if (mm[:, 0] > 0).any():
        print("   frame_begin + member count (9): first start %.1f us after the sweep's end, lists done %.1f, last workgroup done %.1f; move_apply ends %.1f, move_replay ends %.1f"
              % ((first(mm) - occ_end) / 100.0, (mm[:, 1].max() - occ_end) / 100.0, (mm[:, 3].max() - occ_end) / 100.0,
                 (ap[:, 3].max() - occ_end) / 100.0, (rp_end - occ_end) / 100.0), flush=True)
    stt = m.stats()
    print("   issue: host_enqueue_us %.1f, graph frames %d, direct frames %d" % (stt["host_enqueue_us"], stt["graph_frames"], stt["direct_frames"]), flush=True)
    print("map %d: %.4f ms/frame | sweep(8) %.1f us; births end -> sweep start %.1f; sweep end -> move_apply(9) start %.1f | frame 8: visibility end -> bin_sort_gather start %.1f, weight end -> birth_replay start %.1f us"
          % (rep, ms, (occ_end - occ_start) / 100.0, (occ_start - birth_end) / 100.0, (ap_start - occ_end) / 100.0,
             (bsg_start - vis_end) / 100.0, (br_start - wt_end) / 100.0), flush=True)
    m.close()
