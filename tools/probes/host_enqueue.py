"""Host time of sdm_update per frame, launch by launch and as a graph replay (SDM_HOST_TIMING prints the split at close)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["SDM_HOST_TIMING"] = "1"
from semantic_dsp_map_amd import binding, synth  # noqa: E402

cfg = synth.CONFIGS["C3"]
params = synth.PARAMS["vkitti2"]
m = binding.SdmMap(cfg, params, None, device=0)
m.generate_noise_table(seed=20250217)
scene = synth.Scene(cfg, n_static=48, n_dynamic=6, seed=7)
st, ring, _ = synth.prefill_state(cfg, scene, 2000000)
m.load_state(st)
m.set_ring_state(ring)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 10
K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
frames = []
for t in range(W + K):
    depth, cloud, pos, q = scene.render(t, params)
    frames.append((m.device_put(depth), m.device_put(cloud), pos, q, scene.moves(t)))
for f in frames[:W]:
    m.update(*f, on_device=True)
m.synchronize()
t0 = time.perf_counter()
for f in frames[W:]:
    m.update(*f, on_device=True)
t1 = time.perf_counter()
m.synchronize()
t2 = time.perf_counter()
print("mode", os.environ.get("SDM_GRAPH", "2"), "python-side enqueue %.1f us/frame, wall %.1f us/frame" % ((t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6), m.stats()["graph_frames"], m.stats()["direct_frames"])
m.close()
