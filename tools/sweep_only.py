"""Runs only the occupancy/semantic sweep kernel (k_occupancy) on the C3 benchmark map, for rocprofv3 PMC passes:

  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out_fetch -- python tools/sweep_only.py
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out_write -- python tools/sweep_only.py

(separate passes: FETCH_SIZE takes 3 and WRITE_SIZE 2 of the 4 TCC slots, MI355X_MICROARCH.md "rocprofv3 PMC slots")."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_dsp_map_amd import binding, synth  # noqa: E402

cfg = synth.CONFIGS["C3"]
params = synth.PARAMS["vkitti2"]
m = binding.SdmMap(cfg, params)
scene = synth.Scene(cfg, n_static=48, n_dynamic=6, seed=7)
st, ring, n = synth.prefill_state(cfg, scene, 2000000)
m.load_state(st)
m.set_ring_state(ring)
ms = m.time_occupancy_sweep(iters=int(sys.argv[1]) if len(sys.argv) > 1 else 10)
stt = m.stats(count_live=True)
print("sweep avg ms", ms, "live prefill", n, "live_voxels", stt["live_voxels"], "live_particles", stt["live_particles"])
