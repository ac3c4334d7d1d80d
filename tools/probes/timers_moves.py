"""Development aid: where k_move_replay's time goes on the `driven` workload (library built with
tools/ab_build.sh timers -DSDM_AB_TIMERS=1, SDM_LIB_PATH=build/ab/libsdm_timers.so; SDM_DRIVEN_CACHE keeps the rendered
frames).  Per frame of the last stretch of the drive: the slowest head's list walk and whole replay (100 MHz wall clock),
the longest list, the number of lists, the copies re-inserted."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from semantic_dsp_map_amd import binding, synth  # noqa: E402


def main():
    cfg, params = synth.CONFIGS["C3"], synth.PARAMS[synth.DRIVEN_PARAMS]
    n = synth.DRIVEN_FRAMES + 20
    frames = synth.render_frames_cached(cfg, params, synth.DRIVEN_SCENE, range(n), os.environ.get("SDM_DRIVEN_CACHE"))
    scene = synth.Scene(cfg, **synth.DRIVEN_SCENE)
    m = binding.SdmMap(cfg, params, None, device=0)
    m.generate_noise_table(seed=20250217)
    L = m.L
    L.sdm_debug_timers.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    buf = np.zeros(6 * 8192 * 4 + 4 * 4096 * 4, np.uint64)
    for t, (depth, cloud, pos, q) in enumerate(frames):
        # (the last stretch: frames issued back to back, every fifth one waited for - its clocks are the ones read)
        first = int(os.environ.get("SDM_TIMERS_FROM", synth.DRIVEN_FRAMES - 10))   # (SDM_TIMERS_FROM=40: the middle of the drive too)
        read = t >= first and t % 5 == 4
        if t >= first and t % 5 == 0:   # (the reset waits for the device: before the stretch, not inside it; the maxima are over the five frames)
            L.sdm_debug_timers(m.h, buf.ctypes.data, 1)
        m.update(depth, cloud, pos, q, scene.moves(t), sync=read)
        if not read:
            continue
        L.sdm_debug_timers(m.h, buf.ctypes.data, 0)
        mv = buf[6 * 8192 * 4:].astype(np.int64).reshape(4, 4096, 4)
        s = mv[1].reshape(-1)[2048 * 4:]
        r = mv[1][:1024]
        ran = (r[:, 0] > 0) & (r[:, 2] > 0)
        span = (r[ran, 2].max() - r[ran, 0].min()) / 100.0 if ran.any() else 0.0
        fb, mm = mv[3], mv[2]
        ranf = (fb[:, 0] > 0) & (mm[:, 3] > 0)
        if ranf.any():
            t0 = fb[ranf, 0].min()
            print("   k_frame_begin: starts spread %.1f us | zeroing + slab marks done (avg) %.1f | chunk lists built (avg) %.1f, slowest %.1f | end avg %.1f, last %.1f us after the first start"
                  % ((fb[ranf, 0].max() - t0) / 100.0, (fb[ranf, 1] - t0).mean() / 100.0, (mm[ranf, 1] - t0).mean() / 100.0, (mm[ranf, 1] - t0).max() / 100.0,
                     (mm[ranf, 3] - t0).mean() / 100.0, (mm[ranf, 3] - t0).max() / 100.0))
        a = mv[0]
        rana = (a[:, 0] > 0) & (a[:, 3] > 0)
        if rana.any():
            t0 = a[rana, 0].min()
            print("   k_move_apply: workgroups with chunks %d | prefix done avg %.1f | first moves done avg %.1f | end avg %.1f, last %.1f us"
                  % (rana.sum(), (a[rana, 1] - a[rana, 0]).mean() / 100.0, (a[rana, 2] - a[rana, 0]).mean() / 100.0, (a[rana, 3] - t0).mean() / 100.0, (a[rana, 3] - t0).max() / 100.0))
        if ranf.any() and rana.any() and ran.any():
            print("   gaps: k_frame_begin's last end -> k_move_apply's first start %.1f us | k_move_apply's last end -> k_move_replay's first thread-0 stamp %.1f us"
                  % ((a[rana, 0].min() - mm[ranf, 3].max()) / 100.0, (r[ran, 0].min() - a[rana, 3].max()) / 100.0))
        st = m.stats()
        print("frame %d: replay span (thread 0 of the workgroups) %.1f us | slowest head: walk %.1f us, whole %.1f us | lists %d, longest %d, mean %.1f, most copies re-inserted by one head %d | moved %d re-inserted %d | largest per-head sums: walk + selection %.1f us, waiting for the batches' copies %.1f, insertions %.1f | older memberships in the table %d"
              % (t, span, s[0] / 100.0, s[1] / 100.0, s[4], s[2], s[3] / max(s[4], 1), s[5], st["n_moved"], st["n_move_reinserted"], s[6] / 100.0, s[7] / 100.0, s[8] / 100.0, st["alias_entries"]))
    m.close()


if __name__ == "__main__":
    main()
