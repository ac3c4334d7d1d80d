"""The non-incremental sweep on a map the filter grew itself (150 frames of the `driven` scene): time per launch, and - run
under `rocprofv3 --kernel-trace` - its kernels.  SDM_DRIVEN_CACHE=<file>: the rendered frames are kept there (render
once unprofiled, profile the second run)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from semantic_dsp_map_amd import binding, synth  # noqa: E402


def main():
    cfg = synth.CONFIGS["C3"]
    params = synth.PARAMS[synth.DRIVEN_PARAMS]
    scene = synth.Scene(cfg, **synth.DRIVEN_SCENE)
    n = 150
    frames = synth.render_frames_cached(cfg, params, synth.DRIVEN_SCENE, range(n), os.environ.get("SDM_DRIVEN_CACHE"))
    m = binding.SdmMap(cfg, params, None, device=0)
    m.generate_noise_table(seed=20250217)
    out = {"empty_ms": round(m.time_occupancy_sweep(iters=10), 5)}
    for t, (depth, cloud, pos, q) in enumerate(frames):
        m.update(depth, cloud, pos, q, scene.moves(t))
    m.synchronize()
    s = m.stats(count_live=True)
    m.time_occupancy_sweep(iters=100)
    out["driven_ms"] = round(m.time_occupancy_sweep(iters=10), 5)
    out["live_voxels"] = int(s["live_voxels"])
    out["live_particles"] = int(s["live_particles"])
    out["hinted_groups"] = int(m.hinted_groups())
    print(os.path.basename(os.environ.get("SDM_LIB_PATH", "default")), json.dumps(out))
    m.close()


if __name__ == "__main__":
    main()
