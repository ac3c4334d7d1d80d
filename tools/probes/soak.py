"""Development aid: the `driven` drive over and over on one map - 240 frames down the street, then the camera is back at
the start (a jump of 70 m: every slab of the ring is recycled) and drives again -, no oracle beside it: does the library
keep running without an error code, do its counters stay sane (SDM_DRIVEN_CACHE keeps the rendered frames)?
usage: soak.py [passes]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from semantic_dsp_map_amd import binding, synth  # noqa: E402


def main():
    passes = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    cfg, params = synth.CONFIGS["C3"], synth.PARAMS[synth.DRIVEN_PARAMS]
    n = synth.DRIVEN_FRAMES + 20
    frames = synth.render_frames_cached(cfg, params, dict(synth.DRIVEN_SCENE), range(n), os.environ.get("SDM_DRIVEN_CACHE"))
    scene = synth.Scene(cfg, **synth.DRIVEN_SCENE)
    m = binding.SdmMap(cfg, params, None, device=0)
    m.generate_noise_table(seed=20250217)
    for p in range(passes):
        for t, (depth, cloud, pos, q) in enumerate(frames):
            rm = [65000] if t % 37 == 5 else None   # (a track nobody has: the removal path runs, nothing goes)
            m.update(depth, cloud, pos, q, scene.moves(t), rm, sync=(t % 40 == 39))
        m.synchronize()
        s = m.stats(count_live=True)   # raises on any error the frames left behind
        print("pass %d: %d frames, live particles %d in %d voxels, older memberships %d (overflowed %d), moved %d / re-inserted %d in the last frame, births %d"
              % (p, (p + 1) * n, s["live_particles"], s["live_voxels"], s["alias_entries"], s["alias_overflowed"], s["n_moved"], s["n_move_reinserted"], s["n_birth_success"]))
        assert s["alias_overflowed"] == 0
    m.close()
    print("soak ok")


if __name__ == "__main__":
    main()
