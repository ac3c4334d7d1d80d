"""ctypes view of include/sdm_objects.h (the object layer): host code inside libsdm_hip.so, no
GPU needed.  Mirrors the reference's calls: `update` = objectLevelUpdate (semantic_dsp_map.h:304-566), `collect` = the
object loop of the prediction step (:588-736), which yields the `moves` / `remove_tracks` arguments of SdmMap.update."""
import ctypes as C

import numpy as np

from . import binding

MODE_KITTI360, MODE_CODA, MODE_VKITTI2, MODE_ZED2 = 0, 1, 2, 3


class ObjectsConfig(C.Structure):
    _fields_ = [("mode", C.c_int32), ("max_movable_instance_id", C.c_int32),
                ("movement_distance_threshold", C.c_double), ("movement_probability_threshold", C.c_double),
                ("movement_increment", C.c_double), ("movement_decrement", C.c_double),
                ("map_half_size_scaled", C.c_double),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("image_width", C.c_int32), ("image_height", C.c_int32), ("seed", C.c_uint64)]


class Observation(C.Structure):
    _fields_ = [("track_id", C.c_int32), ("label_id", C.c_int32), ("is_static", C.c_int32), ("n_kpts", C.c_int32),
                ("kpts_current", C.c_void_p), ("kpts_previous", C.c_void_p)]


class ObjectMove(C.Structure):
    _fields_ = [("track_id", C.c_int32), ("T", C.c_float * 16)]


class ObjectInfo(C.Structure):
    _fields_ = [("exists", C.c_int32), ("label_id", C.c_int32), ("observation_time_step", C.c_int32),
                ("observation_count", C.c_int32), ("has_moved_flag", C.c_int32), ("moving", C.c_int32),
                ("to_match_with_previous", C.c_int32), ("prediction_available", C.c_int32),
                ("n_transformations", C.c_int32), ("has_t_matrix", C.c_int32),
                ("moved_probability", C.c_double), ("translation_velocity", C.c_double * 3), ("t_matrix", C.c_double * 16)]


_signed = False


def _lib():
    global _signed
    L = binding.load_library()
    if not _signed:
        vp, i32, u32, dbl = C.c_void_p, C.c_int32, C.c_uint32, C.c_double
        sig = {
            "sdm_objects_create": [C.POINTER(ObjectsConfig), C.POINTER(vp)],
            "sdm_objects_clear": [vp],
            "sdm_objects_set_bayes": [vp, dbl, dbl, dbl, dbl],
            "sdm_objects_update": [vp, vp, i32, vp, vp, dbl, u32],
            "sdm_objects_collect": [vp, u32, i32, vp, i32, vp, i32, C.POINTER(i32), vp, i32, C.POINTER(i32)],
            "sdm_objects_query": [vp, i32, C.POINTER(ObjectInfo)],
            "sdm_objects_count": [vp, C.POINTER(i32)],
            "sdm_objects_fit_rigid": [vp, vp, i32, vp],
            "sdm_objects_fit_rigid_ransac": [vp, vp, i32, i32, dbl, i32, C.c_uint64, vp, vp, C.POINTER(i32), C.POINTER(dbl)],
        }
        for name, argtypes in sig.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = C.c_int
        L.sdm_objects_destroy.argtypes = [vp]
        L.sdm_objects_destroy.restype = None
        _signed = True
    return L


def _check(rc, what):
    if rc != 0:
        raise binding.SdmError("%s failed: %s" % (what, binding.STATUS_NAMES.get(rc, rc)))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def fit_rigid(P, Q):
    """estimateTransformation (basic_algorithms.h:54-92).  P, Q: n x 3 -> 4 x 4."""
    P, Q = _f64(P).reshape(-1, 3), _f64(Q).reshape(-1, 3)
    T = np.zeros(16)
    _check(_lib().sdm_objects_fit_rigid(P.ctypes.data, Q.ctypes.data, len(P), T.ctypes.data), "sdm_objects_fit_rigid")
    return T.reshape(4, 4)


def fit_rigid_ransac(P, Q, max_iterations=100, threshold=0.5, recompute_with_inliers=False, seed=0):
    """estimateTransformationRANSAC (basic_algorithms.h:104-195), seeded -> (T, inlier indices, mse of the inliers)."""
    P, Q = _f64(P).reshape(-1, 3), _f64(Q).reshape(-1, 3)
    T = np.zeros(16)
    inl = np.zeros(len(P), np.int32)
    n_in, mse = C.c_int32(0), C.c_double(0)
    _check(_lib().sdm_objects_fit_rigid_ransac(P.ctypes.data, Q.ctypes.data, len(P), max_iterations, threshold,
                                               1 if recompute_with_inliers else 0, seed, T.ctypes.data, inl.ctypes.data,
                                               C.byref(n_in), C.byref(mse)), "sdm_objects_fit_rigid_ransac")
    return T.reshape(4, 4), inl[:n_in.value].tolist(), mse.value


class ObjectLayer:
    """cfg: dict with the fields of sdm_objects_config."""

    def __init__(self, cfg):
        self.L = _lib()
        c = ObjectsConfig(**{k: cfg[k] for k, _ in ObjectsConfig._fields_})
        self.h = C.c_void_p()
        _check(self.L.sdm_objects_create(C.byref(c), C.byref(self.h)), "sdm_objects_create")

    def close(self):
        if self.h:
            self.L.sdm_objects_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear(self):
        _check(self.L.sdm_objects_clear(self.h), "sdm_objects_clear")

    def set_bayes(self, distance_threshold, probability_threshold, increment, decrement):
        _check(self.L.sdm_objects_set_bayes(self.h, distance_threshold, probability_threshold, increment, decrement),
               "sdm_objects_set_bayes")

    def update(self, observations, cam_pos, cam_q, time_stamp, global_time_stamp):
        """observations: dicts with track_id, label_id, is_static, kpts_current (n x 3), kpts_previous (n x 3 / None)."""
        n = len(observations)
        arr = (Observation * max(n, 1))()
        keep = []
        for i, ob in enumerate(observations):
            cur = _f64(ob["kpts_current"]).reshape(-1, 3)
            prev = None if ob.get("kpts_previous") is None else _f64(ob["kpts_previous"]).reshape(-1, 3)
            keep += [cur, prev]
            arr[i] = Observation(ob["track_id"], ob["label_id"], 1 if ob["is_static"] else 0, len(cur),
                                 cur.ctypes.data if len(cur) else None, prev.ctypes.data if prev is not None and len(prev) else None)
        p, q = _f64(cam_pos), _f64(cam_q)
        _check(self.L.sdm_objects_update(self.h, C.cast(arr, C.c_void_p), n, p.ctypes.data, q.ctypes.data, time_stamp,
                                         global_time_stamp), "sdm_objects_update")

    def collect(self, global_time_stamp, max_obersevation_lost_time, present_tracks=(), cap=1024):
        """-> (moves: list of (track_id, 4 x 4 float32)), remove_tracks: list of int)."""
        moves = (ObjectMove * cap)()
        rem = np.zeros(cap, np.int32)
        pres = np.ascontiguousarray(list(present_tracks), dtype=np.int32)
        n_m, n_r = C.c_int32(0), C.c_int32(0)
        _check(self.L.sdm_objects_collect(self.h, global_time_stamp, max_obersevation_lost_time,
                                          pres.ctypes.data if len(pres) else None, len(pres), C.cast(moves, C.c_void_p), cap,
                                          C.byref(n_m), rem.ctypes.data, cap, C.byref(n_r)), "sdm_objects_collect")
        out = [(moves[k].track_id, np.array(moves[k].T, dtype=np.float32).reshape(4, 4)) for k in range(n_m.value)]
        return out, rem[:n_r.value].tolist()

    def query(self, track_id):
        info = ObjectInfo()
        _check(self.L.sdm_objects_query(self.h, track_id, C.byref(info)), "sdm_objects_query")
        d = {k: getattr(info, k) for k, _ in ObjectInfo._fields_ if k not in ("translation_velocity", "t_matrix")}
        d["translation_velocity"] = np.array(info.translation_velocity)
        d["t_matrix"] = np.array(info.t_matrix).reshape(4, 4)
        return d

    def count(self):
        n = C.c_int32(0)
        _check(self.L.sdm_objects_count(self.h, C.byref(n)), "sdm_objects_count")
        return n.value


def write_keypoint_clip(path, cfg, frames, max_obersevation_lost_time=5, present=None):
    """Dump what the tracking node hands the object layer per frame (track ids, labels, 3-D keypoints, pose, time stamp)
    in the format tools/replay/track.cpp reads.  frames: list of (observations, cam_pos, cam_q, time_stamp) as taken by
    ObjectLayer.update; present[t]: track ids that own particles in the map at frame t (optional)."""
    import struct
    c = ObjectsConfig(**{k: cfg[k] for k, _ in ObjectsConfig._fields_})
    with open(path, "wb") as f:
        f.write(b"SDMKPTS1")
        f.write(bytes(c))
        f.write(struct.pack("<iI", max_obersevation_lost_time, len(frames)))
        for t, (obs, pos, q, ts) in enumerate(frames):
            pres = list(present[t]) if present else []
            f.write(_f64(pos).tobytes())
            f.write(_f64(q).tobytes())
            f.write(struct.pack("<dII", ts, len(obs), len(pres)))
            f.write(np.asarray(pres, "<i4").tobytes())
            for ob in obs:
                cur = _f64(ob["kpts_current"]).reshape(-1, 3)
                prev = None if ob.get("kpts_previous") is None else _f64(ob["kpts_previous"]).reshape(-1, 3)
                has_prev = prev is not None and len(prev) == len(cur) and len(cur) > 0
                f.write(struct.pack("<iiiii", ob["track_id"], ob["label_id"], 1 if ob["is_static"] else 0, len(cur), 1 if has_prev else 0))
                f.write(cur.tobytes())
                if has_prev:
                    f.write(prev.tobytes())
