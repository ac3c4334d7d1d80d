"""The dense case of the occupancy sweep alone (for rocprofv3 counter passes and quick A/B runs): C3 map, every slot
live, N launches; mode 0 = eight track ids drawn per slot, 1 = one per voxel (sdm_debug_fill_dense_ex)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from semantic_dsp_map_amd import binding, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
modes = [int(x) for x in sys.argv[2:]] or [0]
m = binding.SdmMap(synth.CONFIGS["C3"], synth.PARAMS["vkitti2"], None, device=0)
V = 1 << 24
for mode in modes:
    m.fill_dense_ex(mode)
    m.time_occupancy_sweep(iters=30)   # (the first sweeps set the group hints and bring the clocks up)
    ms = m.time_occupancy_sweep(iters=n)
    print("dense mode %d: %.4f ms  %.3f of 8 TB/s on layout bytes (81 B/voxel), %.3f on SURVEY 8(d) bytes (80)" % (mode, ms, V * 81 / ms / 1e6 / 8000.0, V * 80 / ms / 1e6 / 8000.0))
