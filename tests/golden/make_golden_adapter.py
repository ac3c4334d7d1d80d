"""Generator of tests/golden/adapter_clips.npz (run from the repository root: python tests/golden/make_golden_adapter.py).

Two short clips of the synthetic street scene in the form the tracking node hands them to SemanticDSPMap::update - depth
image, "static" MONO8 mask, one MONO8 mask + 3-D key points per movable object, pose - together with the occupied / free
clouds the oracle-side model of update() (tests/adapter_model.py: oracle.py + object_layer.py + colour.py) emits for every
frame.  tests/test_adapter_parity.py replays the inputs through the product's C++ class and compares byte for byte.

  vk2    VIRTUAL_KITTI2-style preset (object mode 2: matched key points -> seeded RANSAC), 8 slots per voxel; the third
         object reports a track id above g_max_movable_object_instance_id and is re-allocated (semantic_dsp_map.h:179-186)
  zed2b  ZED2 + BOOST_MODE-style preset (object mode 3: four box key points; inputs at twice the map's image size,
         reduced on the device; sky and per-object box filters), free space asked for, evaluation colour format

Depth is quantised to 1/64 m so that it is stored exactly as uint16; the Gaussian table is a short one (20011 draws)
committed with the clips."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from semantic_dsp_map_amd import synth  # noqa: E402
from tests import adapter_clip, adapter_model  # noqa: E402

PARAMS = dict(synth.PARAMS["vkitti2"])
BAYES = (0.1, 0.69, 0.2, 0.1)


def preset(x_n, y_n, z_n, p_n, voxel, fx, cx, cy, w, h, dmax, win, mode, boost=False, zed2=False):
    p = dict(x_n=x_n, y_n=y_n, z_n=z_n, p_n=p_n, voxel_size=voxel, fx=fx, fy=fx, cx=cx, cy=cy, width=w, height=h, depth_min=0.3,
             depth_max=dmax, window_half=win, consider_instance=True, src_width=0, src_height=0, rescale=1.0, zed2_filters=zed2,
             object_mode=mode)
    if boost:  # SdmGridPreset::Boosted
        p.update(src_width=w, src_height=h, rescale=0.5, fx=0.5 * fx, fy=0.5 * fx, cx=0.5 * cx, cy=0.5 * cy, width=w // 2, height=h // 2,
                 window_half=3)
    return p


def corners(box):
    lo, hi = box[0:3], box[3:6]
    return np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])], np.float64)


def make(name, pre, n_frames, realloc, evaluation_format, want_free, seed):
    # the scene is rendered at the size the sensor delivers (BOOST: twice the map's image)
    src_w = pre["src_width"] or pre["width"]
    src_h = pre["src_height"] or pre["height"]
    k = pre["rescale"]
    scfg = dict(x_n=pre["x_n"], y_n=pre["y_n"], z_n=pre["z_n"], p_n=pre["p_n"], voxel_size=pre["voxel_size"], fx=pre["fx"] / k,
                fy=pre["fy"] / k, cx=pre["cx"] / k, cy=pre["cy"] / k, width=src_w, height=src_h, depth_min=pre["depth_min"],
                depth_max=pre["depth_max"], window_half=pre["window_half"], max_movable_track=synth.MAX_MOVABLE_TRACK)
    sc = synth.Scene(scfg, n_dynamic=3, seed=seed)
    noise = synth.noise_table(n=20011)
    model = adapter_model.AdapterModel(pre, PARAMS, noise, bayes=BAYES, evaluation_format=evaluation_format)
    out = {name + "_preset": json.dumps(pre), name + "_params": json.dumps(PARAMS), name + "_bayes": np.array(BAYES),
           name + "_evaluation_format": np.int32(evaluation_format), name + "_n_frames": np.int32(n_frames)}
    n_moved = 0
    for t in range(n_frames):
        depth, cloud, _, _ = sc.render(t, PARAMS)
        depth_q = np.clip(np.rint(depth.astype(np.float64) * adapter_clip.DEPTH_SCALE), 0, 65535).astype(np.uint16)
        depth = depth_q.astype(np.float32) / np.float32(adapter_clip.DEPTH_SCALE)
        static_mask, objects = synth.raw_inputs(scfg, cloud, sc)
        pos, q = sc.pose(t)
        boxes, boxes_prev = sc.dyn_boxes(t), sc.dyn_boxes(t - 1 if t else 0)
        seg = [dict(track_id=65535, label="static", kpts_current=np.zeros((0, 3)), kpts_previous=None, mask=static_mask)]
        for i, (trk, _lab, mask) in enumerate(objects):
            tid = trk + adapter_model.MAX_MOVABLE if (realloc and i == 2) else trk
            if pre["object_mode"] == 3:  # four box key points, nothing matched
                lo, hi = boxes[i, 0:3], boxes[i, 3:6]
                cur = np.array([lo, [hi[0], lo[1], lo[2]], [lo[0], hi[1], lo[2]], [lo[0], lo[1], hi[2]]], np.float64)
                prev = None
            else:
                cur, prev = corners(boxes[i]), corners(boxes_prev[i])
            seg.append(dict(track_id=int(tid), label="Car", kpts_current=cur, kpts_previous=prev, mask=mask))
        free = bool(want_free and t == n_frames - 1)
        occ, fr = model.update(depth, seg, pos, q, get_freespace=free, time_stamp=0.1 * t)
        n_moved += model.o.stats()["n_moved"]
        kk = "%s_%d_" % (name, t)
        out[kk + "depth_q"] = depth_q
        out[kk + "pose"] = np.array(list(pos) + list(q) + [0.1 * t, 1.0 if free else 0.0], np.float64)
        out[kk + "seg"] = json.dumps([dict(track_id=s["track_id"], label=s["label"], n_kpts=len(s["kpts_current"]),
                                           has_prev=s["kpts_previous"] is not None) for s in seg])
        out[kk + "kpts"] = np.concatenate([np.asarray(s["kpts_current"], np.float64).reshape(-1, 3) for s in seg])
        out[kk + "prev"] = np.concatenate([np.asarray(s["kpts_previous"], np.float64).reshape(-1, 3) if s["kpts_previous"] is not None
                                           else np.zeros((len(s["kpts_current"]), 3)) for s in seg])
        out[kk + "masks"] = np.stack([s["mask"] for s in seg]).astype(np.uint8)
        out[kk + "occ"] = occ
        out[kk + "free"] = fr if fr is not None else np.zeros(0, occ.dtype)
        print("%s frame %d: %d occupied, %s free, %d particles moved so far, tracks with particles %s"
              % (name, t, len(occ), "-" if fr is None else len(fr), n_moved, model.tracks_with_particles()))
    assert model.o.stats()["alias_events"] == 0, "a slot in two owner sets: the per-slot owner export does not list every key"
    assert n_moved > 0, "no object was ever declared moving: the clip does not exercise the move path"
    return out, noise


if __name__ == "__main__":
    clips = {}
    a, noise = make("vk2", preset(5, 5, 5, 3, 0.4, 80.0, 64.0, 40.0, 128, 80, 12.0, 3, mode=2), 7, realloc=True, evaluation_format=False,
                    want_free=False, seed=11)
    clips.update(a)
    b, _ = make("zed2b", preset(5, 4, 5, 2, 0.4, 160.0, 128.0, 72.0, 256, 144, 12.0, 5, mode=3, boost=True, zed2=True), 7, realloc=False,
                evaluation_format=True, want_free=True, seed=5)
    clips.update(b)
    clips["noise"] = noise
    path = os.path.join(ROOT, "tests", "golden", "adapter_clips.npz")
    np.savez_compressed(path, **clips)
    print("wrote", path, os.path.getsize(path), "bytes")
