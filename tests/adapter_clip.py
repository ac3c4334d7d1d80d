"""TEST INFRASTRUCTURE: the clips of tests/test_adapter_parity.py - what the tracking node would hand SemanticDSPMap::update
per frame (depth image, "static" MONO8 mask, one MONO8 mask + key points per movable object, pose), as arrays in an .npz
and as the flat binary file tests/cpp/adapter_parity.cpp reads."""
import json
import struct

import numpy as np

PRESET_KEYS = ["x_n", "y_n", "z_n", "p_n", "voxel_size", "fx", "fy", "cx", "cy", "width", "height", "depth_min", "depth_max",
               "window_half", "consider_instance", "src_width", "src_height", "rescale", "zed2_filters", "object_mode"]
PARAM_KEYS = ["detection_probability", "noise_number", "nb_ptc_num_per_point", "occupancy_threshold", "max_obersevation_lost_time",
              "forgetting_rate", "max_forget_count", "match_score_threshold", "id_transition_probability", "if_consider_depth_noise",
              "if_use_independent_filter", "depth_noise_first_order", "depth_noise_zero_order"]
DEPTH_SCALE = 64.0  # depth is stored as uint16 = depth * 64 (the generator quantises it: exact in float32)


def frames_of(z, name):
    """list of frame dicts (depth f32, seg list, pos, q, ts, free) of clip `name` in the loaded npz z"""
    out = []
    for t in range(int(z[name + "_n_frames"])):
        k = "%s_%d_" % (name, t)
        meta = json.loads(str(z[k + "seg"]))
        depth = z[k + "depth_q"].astype(np.float32) / np.float32(DEPTH_SCALE)
        masks, cur, prev = z[k + "masks"], z[k + "kpts"], z[k + "prev"]
        seg, a = [], 0
        for i, m in enumerate(meta):
            n = m["n_kpts"]
            seg.append(dict(track_id=m["track_id"], label=m["label"], kpts_current=cur[a:a + n].copy(),
                            kpts_previous=prev[a:a + n].copy() if m["has_prev"] else None, mask=masks[i]))
            a += n
        pose = z[k + "pose"]
        out.append(dict(depth=depth, seg=seg, pos=pose[0:3].copy(), q=pose[3:7].copy(), ts=float(pose[7]), free=bool(pose[8] != 0)))
    return out


def write_binary(path, preset, params, bayes, noise, frames, evaluation_format):
    """the clip as tests/cpp/adapter_parity.cpp reads it (little endian, no padding)"""
    with open(path, "wb") as f:
        f.write(b"SDMADPT1")
        f.write(np.asarray([float(preset[k]) for k in PRESET_KEYS], "<f8").tobytes())
        f.write(np.asarray([float(params[k]) for k in PARAM_KEYS], "<f8").tobytes())
        f.write(struct.pack("<4d", *bayes))
        f.write(struct.pack("<i", 1 if evaluation_format else 0))
        noise = np.ascontiguousarray(noise, "<f4")
        f.write(struct.pack("<I", noise.size))
        f.write(noise.tobytes())
        f.write(struct.pack("<I", len(frames)))
        for fr in frames:
            f.write(np.asarray(list(fr["pos"]) + list(fr["q"]) + [fr["ts"]], "<f8").tobytes())
            d = np.ascontiguousarray(fr["depth"], "<f4")
            f.write(struct.pack("<3I", 1 if fr["free"] else 0, d.shape[0], d.shape[1]))
            f.write(d.tobytes())
            f.write(struct.pack("<I", len(fr["seg"])))
            for s in fr["seg"]:
                lab = s["label"].encode()
                cur = np.asarray(s["kpts_current"], "<f8").reshape(-1, 3)
                prev = s["kpts_previous"]
                f.write(struct.pack("<iI", s["track_id"], len(lab)))
                f.write(lab)
                f.write(struct.pack("<II", len(cur), 0 if prev is None else 1))
                f.write(cur.tobytes())
                if prev is not None:
                    f.write(np.asarray(prev, "<f8").reshape(-1, 3).tobytes())
                m = np.ascontiguousarray(s["mask"], np.uint8)
                f.write(struct.pack("<2I", m.shape[0], m.shape[1]))
                f.write(m.tobytes())
