"""Times the non-incremental sweep on the benchmark state (and the empty and dense states).  SDM_LIB_PATH selects the build."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from semantic_dsp_map_amd import binding, synth  # noqa: E402

cfg = synth.CONFIGS["C3"]
params = synth.PARAMS["vkitti2"]
m = binding.SdmMap(cfg, params, None, device=0)
out = {"empty_ms": round(m.time_occupancy_sweep(iters=10), 5)}
scene = synth.Scene(cfg, n_static=48, n_dynamic=6, seed=7)
st, ring, _ = synth.prefill_state(cfg, scene, 2000000)
m.load_state(st)
m.set_ring_state(ring)
out["full_ms"] = round(m.time_occupancy_sweep(iters=10), 5)
if "--dense" in sys.argv:
    m.fill_dense()
    out["dense_ms"] = round(m.time_occupancy_sweep(iters=10), 5)
print(os.path.basename(os.environ.get("SDM_LIB_PATH", "default")), json.dumps(out))
