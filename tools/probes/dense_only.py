"""The dense case of the occupancy sweep alone (for rocprofv3 counter passes): C3 map, every slot live, N launches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from semantic_dsp_map_amd import binding, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
m = binding.SdmMap(synth.CONFIGS["C3"], synth.PARAMS["vkitti2"], None, device=0)
m.fill_dense()
print("dense_ms", m.time_occupancy_sweep(iters=n))
