"""The occupancy sweep on the dense case of SURVEY.md 8(d): every slot of every voxel of the C3 map live, every voxel
observed (sdm_debug_fill_dense).  Prints the average launch time and the bytes/s on 91 B/voxel
(2 B stamp + 1 B flag + 8 B result + status row + record = 11 + 10 S)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_dsp_map_amd import binding, synth  # noqa: E402

cfg = synth.CONFIGS["C3"]
m = binding.SdmMap(cfg, synth.PARAMS["vkitti2"])
m.fill_dense()
ms = m.time_occupancy_sweep(iters=int(sys.argv[1]) if len(sys.argv) > 1 else 10)
V, S = 1 << 24, 8
b = V * (11 + 10 * S)
print("dense sweep avg ms %.4f  bytes %d  %.0f GB/s  frac of 8 TB/s %.3f" % (ms, b, b / ms / 1e6, b / ms / 1e6 / 8000))
