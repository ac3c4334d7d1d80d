"""Development aid: two C3 benchmark maps alive in one process, timed in turn (20 back-to-back frames each): is the
slowdown of a process's second map a property of that map (its streams / memory) or of the process after a first map?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from semantic_dsp_map_amd import sharded, synth  # noqa: E402

cfg, params = synth.CONFIGS["C3"], synth.PARAMS["vkitti2"]
scene = synth.Scene(cfg, n_static=48, n_dynamic=6, seed=7)
st, ring, _ = synth.prefill_state(cfg, scene, 2000000)
host = [scene.render(t, params) + (scene.moves(t),) for t in range(25)]


def make():
    eng = sharded.NativeShardedMap(cfg, params, 0, 1, 0)
    m = eng.map
    m.generate_noise_table(seed=20250217)
    frames = [(m.device_put(h[0]), m.device_put(h[1]), h[2], h[3], h[4]) for h in host]
    return eng, m, frames


def timed(eng, m, frames):
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
    return (time.perf_counter() - t0) / 20 * 1e3


a = make()
print("map A alone: %.4f ms/frame" % timed(*a), flush=True)
b = make()
print("map B (A still alive): %.4f" % timed(*b), flush=True)
print("map A again: %.4f" % timed(*a), flush=True)
print("map B again: %.4f" % timed(*b), flush=True)
a[1].close()
print("map B after A was closed: %.4f" % timed(*b), flush=True)
c = make()
print("map C (created after A was closed): %.4f" % timed(*c), flush=True)
