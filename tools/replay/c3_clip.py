"""Write the C3 benchmark scene as a clip for tools/replay/replay (PCIe-inclusive frame time from plain host buffers)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from semantic_dsp_map_amd import synth  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/c3_clip.bin"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
synth.write_clip(out, "C3", n, "vkitti2", seed=7, n_static=48, n_dynamic=6)
print("wrote", out, os.path.getsize(out) >> 20, "MiB")
