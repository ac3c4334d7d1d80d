"""Development aid: where a frame's time goes BETWEEN its kernels, frames issued back to back as bench.py issues them
(library built with tools/ab_build.sh timers -DSDM_AB_TIMERS=1, SDM_LIB_PATH=build/ab/libsdm_timers.so).  The kernels of
the main stream stamp a 100 MHz wall clock per workgroup at their start and end; after a stretch of frames the stamps of
the LAST frame are read: per kernel the first start and the last end, the gap to the kernel before it.
usage: timers_frame_gaps.py [driven|c3]   (driven: SDM_DRIVEN_CACHE keeps the rendered frames)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from semantic_dsp_map_amd import sharded, synth  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "driven"
    cfg = synth.CONFIGS["C3"]
    steps = 20
    if mode == "driven":
        params = synth.PARAMS[synth.DRIVEN_PARAMS]
        scene = synth.Scene(cfg, **synth.DRIVEN_SCENE)
        n_grow = synth.DRIVEN_FRAMES
        rendered = synth.render_frames_cached(cfg, params, dict(synth.DRIVEN_SCENE), range(n_grow + steps), os.environ.get("SDM_DRIVEN_CACHE"))
        eng = sharded.NativeShardedMap(cfg, params, 0, 1, 0)
        m = eng.map
        m.generate_noise_table(seed=20250217)
        for t in range(n_grow):
            depth, cloud, pos, q = rendered[t]
            m.update(depth, cloud, pos, q, scene.moves(t), sync=(t % 8 == 7))
        m.synchronize()
        frames = [(rendered[t][2], rendered[t][3], scene.moves(t), m.device_put(rendered[t][0]), m.device_put(rendered[t][1])) for t in range(n_grow, n_grow + steps)]
    else:
        params = synth.PARAMS["vkitti2"]
        scene = synth.Scene(cfg, n_static=48, n_dynamic=6, seed=7)
        eng = sharded.NativeShardedMap(cfg, params, 0, 1, 0)
        m = eng.map
        m.generate_noise_table(seed=20250217)
        frames = []
        for t in range(steps + 5):
            depth, cloud, pos, q = scene.render(t, params)
            frames.append((pos, q, scene.moves(t), m.device_put(depth), m.device_put(cloud)))
        st, ring, _ = synth.prefill_state(cfg, scene, 2000000)
        m.load_state(st)
        m.set_ring_state(ring)
    L = m.L
    L.sdm_debug_timers.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    buf = np.zeros(6 * 8192 * 4 + 4 * 4096 * 4, np.uint64)
    for f in frames[:5]:
        eng.update(f[3], f[4], f[0], f[1], f[2])
    m.synchronize()
    L.sdm_debug_timers(m.h, buf.ctypes.data, 1)
    import time
    t_a = time.perf_counter()
    for f in frames[5:]:
        eng.update(f[3], f[4], f[0], f[1], f[2])
    m.synchronize()
    ms = (time.perf_counter() - t_a) * 1e3 / (len(frames) - 5)
    L.sdm_debug_timers(m.h, buf.ctypes.data, 0)
    print("%.4f ms per frame (wall clock over the stretch, the library with its clocks compiled in)" % ms)
    k = buf[:6 * 8192 * 4].astype(np.int64).reshape(6, 8192, 4)
    mv = buf[6 * 8192 * 4:].astype(np.int64).reshape(4, 4096, 4)
    fb0 = mv[3][:1024, 0]
    t0 = fb0[fb0 > 0].max() - 3000   # the last frame's k_frame_begin starts within 30 us of its latest workgroup
    chain = [("k_frame_begin", mv[3][:, 0], mv[2][:, 3]), ("k_move_apply", mv[0][:, 0], mv[0][:, 3]), ("k_move_replay", mv[1][:1024, 0], mv[1][:1024, 2]),
             ("k_visibility", k[1][:, 0], k[1][:, 3]), ("k_bin_rows", k[2][:, 0], k[2][:, 1]), ("k_ck", k[3][:, 0], k[3][:, 1]),
             ("k_weight", k[4][:, 0], k[4][:, 1]), ("k_birth_replay", k[0][:, 0], k[0][:, 2]), ("k_occupancy", k[5][:, 0], k[5][:, 3])]
    prev_end = None
    print("%s: the last of %d frames issued back to back (us from the first workgroup of its k_frame_begin)" % (mode, len(frames) - 5))
    base = None
    for name, s0, e0 in chain:
        s, e = s0[s0 >= t0], e0[e0 >= t0]
        if not len(s) or not len(e):
            print("  %-16s no stamps" % name)
            continue
        if base is None:
            base = s.min()
        a, b = (s.min() - base) / 100.0, (e.max() - base) / 100.0
        print("  %-16s first start %7.1f  last end %7.1f  (%5.1f us, %5d workgroups stamped)%s"
              % (name, a, b, b - a, len(s), "" if prev_end is None else "   gap to the kernel before: %5.1f us" % (a - prev_end)))
        prev_end = b
        both = (s0 >= t0) & (e0 >= t0) & (e0 >= s0)
        if both.any():
            life = (e0[both] - s0[both]) / 100.0
            st = (s0[both] - s0[both].min()) / 100.0
            print("  %-16s   a workgroup lives %.1f us on average (median %.1f, 99th percentile %.1f, longest %.1f); the starts spread over %.1f us (median start %.1f)"
                  % ("", life.mean(), np.median(life), np.percentile(life, 99), life.max(), st.max(), np.median(st)))
    # k_birth_replay, workgroup by workgroup (stamps: start, heads compacted, end; fourth word: thread 0's insertions)
    b = k[0]
    ok = (b[:, 0] >= t0) & (b[:, 2] >= b[:, 0]) & (b[:, 1] >= b[:, 0])
    if ok.any():
        life = (b[ok, 2] - b[ok, 0]) / 100.0
        front = (b[ok, 1] - b[ok, 0]) / 100.0
        order = np.argsort(life)
        print("  k_birth_replay by workgroup: start -> heads compacted %.1f us on average (99th percentile %.1f), compacted -> end %.1f (99th percentile %.1f)"
              % (front.mean(), np.percentile(front, 99), (life - front).mean(), np.percentile(life - front, 99)))
        slow = order[-12:]
        print("    the twelve slowest: " + ", ".join("%.1f = %.1f + %.1f (wg %d, start +%.1f)" % (life[i], front[i], life[i] - front[i], np.flatnonzero(ok)[i], (b[ok, 0][i] - b[ok, 0].min()) / 100.0) for i in slow))
    # k_visibility, workgroup by workgroup (stamps: start; masks of the LAST round known; its empty candidates done; end)
    vv = k[1][:2048]
    ok = (vv[:, 0] >= t0) & (vv[:, 3] >= vv[:, 0])
    if ok.any():
        life = (vv[ok, 3] - vv[ok, 0]) / 100.0
        print("  k_visibility by workgroup: lifetime percentiles 10 / 50 / 90 / 99 / 100: %s us" % " / ".join("%.1f" % np.percentile(life, q) for q in (10, 50, 90, 99, 100)))
        full = ok & (vv[:, 1] >= vv[:, 0]) & (vv[:, 2] >= vv[:, 1])
        if full.any():
            a0, a1, a2 = (vv[full, 1] - vv[full, 0]) / 100.0, (vv[full, 2] - vv[full, 1]) / 100.0, (vv[full, 3] - vv[full, 2]) / 100.0
            print("    workgroups whose last round had candidates (%d): start -> its masks %.1f us (99th percentile %.1f: earlier rounds are in here), empty candidates %.1f (%.1f), voxels that hold something %.1f (%.1f)"
                  % (full.sum(), a0.mean(), np.percentile(a0, 99), a1.mean(), np.percentile(a1, 99), a2.mean(), np.percentile(a2, 99)))
    m.close()


if __name__ == "__main__":
    main()
