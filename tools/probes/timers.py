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
buf = np.zeros(6 * 8192 * 4 + 4 * 4096 * 4, np.uint64)
us = lambda x: x / 100.0


def span(name, a, extra=""):
    """a: [workgroup][4] of one kernel, checkpoint 0 = start, 1 = end"""
    ran = (a[:, 0] > 0) & (a[:, 1] > 0)
    if not ran.any():
        return
    t0 = a[a[:, 0] > 0, 0].min()
    dur = a[ran, 1] - a[ran, 0]
    print("   %-24s %5d workgroups: starts spread %.1f us | duration avg %.1f / p90 %.1f / max %.1f us | last end %.1f us %s"
          % (name, ran.sum(), us(a[ran, 0].max() - t0), us(dur.mean()), us(np.percentile(dur, 90)), us(dur.max()), us(a[ran, 1].max() - t0), extra))


for t in range(10):
    depth, cloud, pos, q = scene.render(t, params)
    L.sdm_debug_timers(m.h, buf.ctypes.data, 1)
    m.update(depth, cloud, pos, q, scene.moves(t), sync=True)
    L.sdm_debug_timers(m.h, buf.ctypes.data, 0)
    if t < 7:
        continue
    k = buf[:6 * 8192 * 4].astype(np.int64).reshape(6, 8192, 4)
    mv = buf[6 * 8192 * 4:].astype(np.int64).reshape(4, 4096, 4)
    print("frame %d (thread 0 of every workgroup; 100 MHz wall clock)" % t)
    mm = mv[2].copy()
    span("move_members (lists)", mm, "| chunks: first done %.1f, all done %.1f us after the kernel's first start" % (us((mm[mm[:, 2] > 0, 2].min() if (mm[:, 2] > 0).any() else 0) - mm[mm[:, 0] > 0, 0].min()), us(mm[:, 3].max() - mm[mm[:, 0] > 0, 0].min())) if (mm[:, 0] > 0).any() else "")
    a = mv[0].copy()
    act = (a[:, 0] > 0) & (a[:, 2] > 0)
    extra = "| prefix %.1f, moves %.1f us (avg, chunks with members)" % (us((a[act, 1] - a[act, 0]).mean()), us((a[act, 2] - a[act, 1]).mean())) if act.any() else ""
    a[:, 1] = a[:, 3]
    span("move_apply", a, extra)
    r = mv[1].copy()
    hd = (r[:, 0] > 0) & (r[:, 1] > 0) & (r[:, 2] > 0)
    extra = "| heads (thread 0): list walked %.1f, copies + insertions %.1f us (avg), copies re-inserted avg %.1f" % (us((r[hd, 1] - r[hd, 0]).mean()), us((r[hd, 2] - r[hd, 1]).mean()), r[hd, 3].mean()) if hd.any() else ""
    r[:, 1] = r[:, 2]
    span("move_replay", r, extra)
    v = k[1]; w = v.copy(); w[:, 1] = w[:, 3]
    act = (v[:, 0] > 0) & (v[:, 3] > 0)
    span("visibility", w, "| masks %.1f, empty voxels %.1f, full voxels %.1f us (avg)" % (us((v[act, 1] - v[act, 0]).mean()), us((v[act, 2] - v[act, 1]).mean()), us((v[act, 3] - v[act, 2]).mean())) if act.any() else "")
    span("bin_rows", k[2])
    c = k[3]
    span("ck heavy part", c[:1536], "| batches per workgroup max %d" % c[:4096, 2].max())
    span("ck light part", c[1536:])
    span("weight", k[4])
    b = k[0].copy(); b[:, 1] = b[:, 2]
    span("birth_replay (heads)", b)
    o = k[5]
    early = o.copy(); early[o[:, 3] > 0] = 0
    span("occupancy (no tile)", early)
    full = o.copy(); full[:, 1] = full[:, 3]
    span("occupancy (one tile)", full)

    def first_last(a, end_col=1):
        ran = a[:, 0] > 0
        ends = a[:, end_col][a[:, end_col] > 0]
        return (a[ran, 0].min(), max(ends.max(), a[ran, 0].max())) if ran.any() and ends.size else None

    order = [("move_apply", mv[0], 3), ("move_replay", mv[1], 2), ("visibility", k[1], 3), ("bin_rows", k[2], 1), ("ck", k[3], 1),
             ("weight", k[4], 1), ("birth_replay", k[0], 2), ("occupancy", k[5], 3)]
    tl = [(n, first_last(a, c)) for n, a, c in order]
    tl = [(n, x) for n, x in tl if x]
    t0 = tl[0][1][0]
    print("   main stream, first workgroup start -> last recorded end, us from move_apply's start (k_ck_classify, not instrumented, sits between bin_rows and ck):")
    prev_end = None
    for n, (a0, a1) in tl:
        print("      %-16s %7.1f -> %7.1f%s" % (n, us(a0 - t0), us(a1 - t0), "" if prev_end is None else "   gap to the kernel before: %.1f us" % us(a0 - prev_end)))
        prev_end = a1
