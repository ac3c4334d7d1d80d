"""ctypes wrapper of the CPU oracle (oracle/cpu_ref.cpp).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never from the product package.
PARITY UNPINNED (see cpu_ref.h).

The shared library is compiled on demand with -march=native; one build per
host CPU (keyed by the cpuinfo flags) so that a library built in the CPU
container is never executed on a different GPU-box CPU.
"""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def _cpu_tag():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return hashlib.sha1(line.encode()).hexdigest()[:10]
    except OSError:
        pass
    return "generic"


def build(force=False):
    out = os.path.join(_HERE, "_build", "libsdm_oracle_%s.so" % _cpu_tag())
    src = [os.path.join(_HERE, "cpu_ref.cpp"), os.path.join(_HERE, "cpu_ref.h")]
    stale = (not os.path.exists(out)) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "OUT=" + os.path.relpath(out, _HERE)])
    return out


class Config(C.Structure):
    _fields_ = [("x_n", C.c_int32), ("y_n", C.c_int32), ("z_n", C.c_int32), ("p_n", C.c_int32),
                ("voxel_size", C.c_float),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("width", C.c_int32), ("height", C.c_int32),
                ("depth_min", C.c_float), ("depth_max", C.c_float),
                ("window_half", C.c_int32), ("max_movable_track", C.c_int32),
                ("bin_order", C.c_int32), ("ck_slabs", C.c_int32)]


class Params(C.Structure):
    _fields_ = [("detection_probability", C.c_float), ("noise_number", C.c_float),
                ("nb_ptc_num_per_point", C.c_int32), ("occupancy_threshold", C.c_float),
                ("max_obersevation_lost_time", C.c_int32), ("forgetting_rate", C.c_float),
                ("max_forget_count", C.c_int32), ("match_score_threshold", C.c_float),
                ("id_transition_probability", C.c_float),
                ("if_consider_depth_noise", C.c_int32), ("if_use_independent_filter", C.c_int32),
                ("depth_noise_first_order", C.c_float), ("depth_noise_zero_order", C.c_float)]


class RingState(C.Structure):
    _fields_ = [("global_time_stamp", C.c_uint32), ("moved_steps", C.c_int32 * 3), ("eq_steps", C.c_int32 * 3),
                ("map_center", C.c_float * 3), ("last_pos", C.c_float * 3),
                ("birth_cursor", C.c_int32), ("move_cursor", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("live_particles", C.c_int64), ("n_visible", C.c_int64), ("n_birth_attempts", C.c_int64),
                ("n_birth_success", C.c_int64), ("n_resampled_voxels", C.c_int64), ("n_moved", C.c_int64),
                ("n_move_reinserted", C.c_int64), ("n_frustum_voxels", C.c_int64), ("n_occupied", C.c_int64),
                ("alias_events", C.c_int64), ("bfs_start_in_frustum", C.c_int64), ("stage_ms", C.c_double * 8)]


LABELED_POINT = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("sigma", "<f4"),
                          ("track_id", "<u2"), ("label_id", "u1"), ("is_valid", "u1")])
OBJECT_MOVE = np.dtype([("track_id", "<i4"), ("T", "<f4", (16,))])
VOXEL_RESULT = np.dtype([("wsum", "<f4"), ("track", "<u2"), ("label", "u1"), ("occ", "i1")])
assert LABELED_POINT.itemsize == 20 and OBJECT_MOVE.itemsize == 68 and VOXEL_RESULT.itemsize == 8

STATE_FIELDS = [("px", np.float32), ("py", np.float32), ("pz", np.float32), ("w", np.float32),
                ("ts", np.uint16), ("track", np.uint16), ("label", np.uint8), ("status", np.uint8),
                ("forget", np.uint8), ("owner", np.uint16)]

STAGES = {"all": 0, "ego": 1, "move": 2, "remove": 3, "visibility": 4, "weight": 5, "birth": 6, "occupancy": 7}

_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        vp = C.c_void_p
        L.oracle_create.restype = vp
        L.oracle_create.argtypes = [C.POINTER(Config)]
        L.oracle_destroy.argtypes = [vp]
        L.oracle_clone.restype = vp
        L.oracle_clone.argtypes = [vp, C.c_int32]
        L.oracle_clear.argtypes = [vp]
        L.oracle_set_params.argtypes = [vp, C.POINTER(Params)]
        L.oracle_set_noise_table.argtypes = [vp, vp, C.c_int32]
        L.oracle_update.restype = C.c_int
        L.oracle_update.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int32, vp, C.c_int32, C.c_int32]
        L.oracle_get_voxels.argtypes = [vp, vp]
        L.oracle_get_stats.argtypes = [vp, C.POINTER(Stats)]
        L.oracle_get_ring_state.argtypes = [vp, C.POINTER(RingState)]
        L.oracle_set_ring_state.argtypes = [vp, C.POINTER(RingState)]
        L.oracle_get_stamps.argtypes = [vp, vp, vp, vp]
        L.oracle_set_stamps.argtypes = [vp, vp, vp, vp]
        L.oracle_dump_state.argtypes = [vp] + [vp] * 10
        L.oracle_load_state.argtypes = [vp] + [vp] * 10
        L.oracle_get_ck_kappa.argtypes = [vp, vp]
        L.oracle_get_bin_counts.argtypes = [vp, vp]
        L.oracle_get_bins.restype = C.c_int64
        L.oracle_get_bins.argtypes = [vp, vp, C.c_int64]
        L.oracle_get_extrinsic.argtypes = [vp, vp]
        L.oracle_get_pdf_table.argtypes = [vp, vp]
        L.oracle_point_in_frustum.argtypes = [vp, C.c_float, C.c_float, C.c_float]
        L.oracle_point_in_frustum.restype = C.c_int32
        L.oracle_generate_cloud.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int32, vp, vp, C.c_int32, vp]
        L.oracle_generate_cloud_ex.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int32, vp, vp, C.c_int32, C.c_int32, C.c_int32,
                                               C.c_float, C.c_int32, vp, vp, vp]
        L.oracle_pos_to_voxel.restype = C.c_uint32
        L.oracle_pos_to_voxel.argtypes = [vp, C.c_float, C.c_float, C.c_float]
        L.oracle_voxel_to_pos.argtypes = [vp, C.c_uint32, vp]
        L.oracle_query_pdf.restype = C.c_float
        L.oracle_query_pdf.argtypes = [vp, C.c_float, C.c_float, C.c_float]
        L.oracle_forgetting_factor.restype = C.c_float
        L.oracle_forgetting_factor.argtypes = [vp, C.c_int32]
        L.oracle_add_particle.restype = C.c_uint32
        L.oracle_add_particle.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_uint8, C.c_uint16]
        L.oracle_resample_voxel.restype = C.c_int32
        L.oracle_resample_voxel.argtypes = [vp, C.c_uint32]
        L.oracle_set_global_time_stamp.argtypes = [vp, C.c_uint32]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleMap:
    """One reference-equivalent map (the reference allows exactly one per process; this does not)."""

    def __init__(self, cfg, params=None, noise_table=None):
        """cfg: dict with the Config fields (see sdm config helpers in tests/scenes.py)."""
        self.L = lib()
        c = Config()
        for k, _ in Config._fields_:
            if k in cfg:
                setattr(c, k, cfg[k])
        if not c.ck_slabs:
            c.ck_slabs = 1
        self.cfg = c
        self.h = self.L.oracle_create(C.byref(c))
        if not self.h:
            raise ValueError("oracle_create rejected the configuration")
        self.V = 1 << (c.x_n + c.y_n + c.z_n)
        self.S = 1 << c.p_n
        self.W, self.H = c.width, c.height
        if params is not None:
            self.set_params(params)
        if noise_table is not None:
            self.set_noise_table(noise_table)

    def clone(self, bin_order=None):
        """A deep copy of the whole map (the owner sets with their double memberships included, which dump_state /
        load_state cannot carry), optionally with the other summation order from here on."""
        o = OracleMap.__new__(OracleMap)
        o.L, o.cfg, o.V, o.S, o.W, o.H = self.L, self.cfg, self.V, self.S, self.W, self.H
        o.h = self.L.oracle_clone(self.h, self.cfg.bin_order if bin_order is None else bin_order)
        if not o.h:
            raise MemoryError("oracle_clone")
        return o

    def __del__(self):
        if getattr(self, "h", None):
            self.L.oracle_destroy(self.h)
            self.h = None

    def clear(self):
        self.L.oracle_clear(self.h)

    def set_params(self, params):
        p = Params()
        for k, _ in Params._fields_:
            setattr(p, k, params[k])
        self.L.oracle_set_params(self.h, C.byref(p))

    def set_noise_table(self, table):
        t = np.ascontiguousarray(table, dtype=np.float32)
        self.L.oracle_set_noise_table(self.h, _ptr(t), t.size)

    def update(self, depth, cloud, cam_pos, cam_q, moves=None, remove_tracks=None, stop_after="all"):
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        cloud = np.ascontiguousarray(cloud, dtype=LABELED_POINT)
        assert depth.size == self.W * self.H and cloud.size == self.W * self.H
        pos = np.ascontiguousarray(cam_pos, dtype=np.float32)
        q = np.ascontiguousarray(cam_q, dtype=np.float32)
        mv = np.ascontiguousarray(moves if moves is not None else np.zeros(0, OBJECT_MOVE), dtype=OBJECT_MOVE)
        rm = np.ascontiguousarray(remove_tracks if remove_tracks is not None else [], dtype=np.int32)
        return self.L.oracle_update(self.h, _ptr(depth), _ptr(cloud), _ptr(pos), _ptr(q), _ptr(mv), mv.size,
                                    _ptr(rm), rm.size, STAGES[stop_after] if isinstance(stop_after, str) else stop_after)

    def voxels(self):
        out = np.empty(self.V, VOXEL_RESULT)
        self.L.oracle_get_voxels(self.h, _ptr(out))
        return out

    def stats(self):
        s = Stats()
        self.L.oracle_get_stats(self.h, C.byref(s))
        d = {k: getattr(s, k) for k, _ in Stats._fields_ if k != "stage_ms"}
        d["stage_ms"] = list(s.stage_ms)
        return d

    def ring_state(self):
        r = RingState()
        self.L.oracle_get_ring_state(self.h, C.byref(r))
        return {"global_time_stamp": r.global_time_stamp, "moved_steps": list(r.moved_steps),
                "eq_steps": list(r.eq_steps), "map_center": list(r.map_center), "last_pos": list(r.last_pos),
                "birth_cursor": r.birth_cursor, "move_cursor": r.move_cursor}

    def set_ring_state(self, d):
        r = RingState()
        r.global_time_stamp = d["global_time_stamp"]
        for i in range(3):
            r.moved_steps[i] = d["moved_steps"][i]
            r.eq_steps[i] = d["eq_steps"][i]
            r.map_center[i] = d["map_center"][i]
            r.last_pos[i] = d["last_pos"][i]
        r.birth_cursor = d["birth_cursor"]
        r.move_cursor = d["move_cursor"]
        self.L.oracle_set_ring_state(self.h, C.byref(r))

    def stamps(self):
        c = self.cfg
        sx = np.empty(1 << c.x_n, np.uint32)
        sy = np.empty(1 << c.y_n, np.uint32)
        sz = np.empty(1 << c.z_n, np.uint32)
        self.L.oracle_get_stamps(self.h, _ptr(sx), _ptr(sy), _ptr(sz))
        return sx, sy, sz

    def set_stamps(self, sx, sy, sz):
        sx, sy, sz = (np.ascontiguousarray(a, dtype=np.uint32) for a in (sx, sy, sz))
        self.L.oracle_set_stamps(self.h, _ptr(sx), _ptr(sy), _ptr(sz))

    def dump_state(self):
        n = self.V * self.S
        st = {k: np.empty(n, dt) for k, dt in STATE_FIELDS}
        self.L.oracle_dump_state(self.h, *[_ptr(st[k]) for k, _ in STATE_FIELDS])
        return st

    def load_state(self, st):
        arrs = [np.ascontiguousarray(st[k], dtype=dt) for k, dt in STATE_FIELDS]
        self.L.oracle_load_state(self.h, *[_ptr(a) for a in arrs])

    def ck_kappa(self):
        out = np.empty(self.W * self.H, np.float32)
        self.L.oracle_get_ck_kappa(self.h, _ptr(out))
        return out.reshape(self.H, self.W)

    def bin_counts(self):
        out = np.empty(self.W * self.H, np.uint32)
        self.L.oracle_get_bin_counts(self.h, _ptr(out))
        return out.reshape(self.H, self.W)

    def bins(self):
        n = int(self.bin_counts().sum())
        out = np.empty(max(n, 1), np.uint32)
        m = self.L.oracle_get_bins(self.h, _ptr(out), n)
        assert m == n
        return out[:n]

    def extrinsic(self):
        out = np.empty(16, np.float32)
        self.L.oracle_get_extrinsic(self.h, _ptr(out))
        return out.reshape(4, 4)

    def generate_cloud_ex(self, depth, static_mask, label_to_inst, objects, cam_pos, cam_q, consider_instance=True,
                          src_size=None, rescale=1.0, sky_instance=-1, object_bbox=None):
        """generate_cloud with the preset-specific parts: BOOST-mode inputs of src_size = (width, height) reduced by
        `rescale`, ZED2 sky exclusion and per-object boxes (n_objects x 6 doubles).  Returns (cloud, resized depth)."""
        hw = self.W * self.H
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        sm = None if static_mask is None else np.ascontiguousarray(static_mask, dtype=np.uint8)
        tab = np.ascontiguousarray(label_to_inst, dtype=np.uint16)
        tr = np.array([o[0] for o in objects], np.int32)
        lb = np.array([o[1] for o in objects], np.int32)
        masks = np.ascontiguousarray(np.concatenate([np.asarray(o[2], np.uint8).reshape(-1) for o in objects])
                                     if objects else np.zeros(1, np.uint8))
        pos = np.ascontiguousarray(cam_pos, dtype=np.float64)
        q = np.ascontiguousarray(cam_q, dtype=np.float64)
        bb = None if object_bbox is None else np.ascontiguousarray(object_bbox, dtype=np.float64)
        out = np.zeros(hw, LABELED_POINT)
        dout = np.zeros(hw, np.float32)
        sw, sh = src_size if src_size else (0, 0)
        self.L.oracle_generate_cloud_ex(self.h, _ptr(depth), _ptr(sm), _ptr(tab), _ptr(tr), _ptr(lb), _ptr(masks), len(objects),
                                        _ptr(pos), _ptr(q), 1 if consider_instance else 0, sw, sh, float(rescale),
                                        int(sky_instance), _ptr(bb), _ptr(out), _ptr(dout))
        return out, dout

    def generate_cloud(self, depth, static_mask, label_to_inst, objects, cam_pos, cam_q, consider_instance=True):
        """objects: list of (track_id, label_id, mask HxW uint8).  Returns the LabeledPoint image."""
        hw = self.W * self.H
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        sm = None if static_mask is None else np.ascontiguousarray(static_mask, dtype=np.uint8)
        tab = np.ascontiguousarray(label_to_inst, dtype=np.uint16)
        tr = np.array([o[0] for o in objects], np.int32)
        lb = np.array([o[1] for o in objects], np.int32)
        masks = np.ascontiguousarray(np.concatenate([np.asarray(o[2], np.uint8).reshape(-1) for o in objects])
                                     if objects else np.zeros(1, np.uint8))
        pos = np.ascontiguousarray(cam_pos, dtype=np.float64)
        q = np.ascontiguousarray(cam_q, dtype=np.float64)
        out = np.zeros(hw, LABELED_POINT)
        self.L.oracle_generate_cloud(self.h, _ptr(depth), _ptr(sm), _ptr(tab), _ptr(tr), _ptr(lb), _ptr(masks), len(objects),
                                     _ptr(pos), _ptr(q), 1 if consider_instance else 0, _ptr(out))
        return out

    def point_in_frustum(self, x, y, z):
        return bool(self.L.oracle_point_in_frustum(self.h, float(x), float(y), float(z)))

    def pdf_table(self):
        out = np.empty(20000, np.float32)
        self.L.oracle_get_pdf_table(self.h, _ptr(out))
        return out

    # known-answer helpers
    def pos_to_voxel(self, x, y, z):
        return int(self.L.oracle_pos_to_voxel(self.h, x, y, z))

    def voxel_to_pos(self, v):
        out = np.empty(3, np.float32)
        self.L.oracle_voxel_to_pos(self.h, v, _ptr(out))
        return out

    def query_pdf(self, x, mu, sigma):
        return float(self.L.oracle_query_pdf(self.h, x, mu, sigma))

    def forgetting_factor(self, c):
        return float(self.L.oracle_forgetting_factor(self.h, c))

    def add_particle(self, x, y, z, label=0, track=65535):
        return int(self.L.oracle_add_particle(self.h, x, y, z, label, track))

    def resample_voxel(self, v):
        return int(self.L.oracle_resample_voxel(self.h, v))

    def set_global_time_stamp(self, t):
        self.L.oracle_set_global_time_stamp(self.h, t)
