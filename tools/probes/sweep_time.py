"""Times the occupancy sweep alone on the benchmark state (C3): in-frame launches (stage timers), the non-incremental
launch and the dense case.  For kernel experiments: SDM_LIB_PATH selects the library build under test."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from semantic_dsp_map_amd import binding, synth  # noqa: E402


def main():
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    cfg = synth.CONFIGS["C3"]
    params = synth.PARAMS["vkitti2"]
    m = binding.SdmMap(cfg, params, None, device=0)
    m.generate_noise_table(seed=20250217)
    scene = synth.Scene(cfg, n_static=48, n_dynamic=6, seed=7)
    st, ring, _ = synth.prefill_state(cfg, scene, 2000000)
    m.load_state(st)
    m.set_ring_state(ring)
    m.set_profiling(True)
    sweep, tiles, live, frame_ms = [], [], [], []
    for t in range(n_frames):
        depth, cloud, pos, q = scene.render(t, params)
        m.update(m.device_put(depth), m.device_put(cloud), pos, q, scene.moves(t), on_device=True)
        m.synchronize()
        s = m.stats()
        if t >= 2:
            sweep.append(s["stage_ms"][7])
            tiles.append(s["sweep_tiles"])
            live.append(s["sweep_live_voxels"])
            frame_ms.append(sum(s["stage_ms"][1:]))
    out = {"in_frame_ms": round(float(np.mean(sweep)), 5), "in_frame_min_ms": round(float(np.min(sweep)), 5),
           "tiles": float(np.mean(tiles)), "evaluated": float(np.mean(live)), "frame_gpu_ms": round(float(np.mean(frame_ms)), 4)}
    m.set_profiling(False)
    out["full_ms"] = round(m.time_occupancy_sweep(iters=10), 5)
    m.fill_dense()
    out["dense_ms"] = round(m.time_occupancy_sweep(iters=10), 5)
    print(os.environ.get("SDM_LIB_PATH", "default"), json.dumps(out))


if __name__ == "__main__":
    main()
