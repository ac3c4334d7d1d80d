"""Times sdm_clear on a used C3 map (every slot of every voxel live: sdm_debug_fill_dense before each call, so that the
kernel has something to reset everywhere).  Wall clock around sdm_clear + sdm_synchronize, N calls; SDM_LIB_PATH selects the
build.  Bytes as bench.py's `roofline.clear` counts them."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from semantic_dsp_map_amd import binding, synth  # noqa: E402

cfg = synth.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "C3"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
m = binding.SdmMap(cfg, synth.PARAMS["vkitti2"], None, device=0)
V, S = m.v_count, m.S
ts = []
for i in range(n + 2):
    m.fill_dense()
    m.synchronize()
    t0 = time.perf_counter()
    m.clear()
    m.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
ts = sorted(ts[2:])
b = V * S * (16 + 2) + V * (S - 1) * (4 + 2 + 1) + V * (2 + 1 + 8)   # pos4, owner per slot; w, ts, status per particle slot; vts, vflag, res
med = ts[len(ts) // 2]
print(os.path.basename(os.environ.get("SDM_LIB_PATH", "default")),
      json.dumps({"clear_ms_median": round(med, 4), "min": round(ts[0], 4), "max": round(ts[-1], 4), "bytes": b,
                  "frac_of_8TBps": round(b / med / 1e6 / 8000, 4)}))
