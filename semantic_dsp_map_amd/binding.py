"""ctypes binding of libsdm_hip (include/sdm.h) — plumbing for tests and bench.py.

The product is the C-ABI library; this module only marshals numpy / torch
buffers into it.  There is no CPU path: if the HIP library is missing or no
GPU is visible, loading / creating a map fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDM_LIB_PATH") or os.path.join(_HERE, "csrc", "libsdm_hip.so")  # override: debug builds only

LABELED_POINT = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("sigma", "<f4"),
                          ("track_id", "<u2"), ("label_id", "u1"), ("is_valid", "u1")])
OBJECT_MOVE = np.dtype([("track_id", "<i4"), ("T", "<f4", (16,))])
VOXEL_RESULT = np.dtype([("wsum", "<f4"), ("track", "<u2"), ("label", "u1"), ("occ", "i1")])
POINT = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("track", "<u2"), ("label", "u1"), ("occ", "i1")])
# sdm_point_xyzrgb = pcl::PointXYZRGB's 32 bytes
POINT_XYZRGB = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("one", "<f4"), ("b", "u1"), ("g", "u1"), ("r", "u1"), ("a", "u1"),
                         ("pad", "<u4", (3,))])
assert LABELED_POINT.itemsize == 20 and OBJECT_MOVE.itemsize == 68 and VOXEL_RESULT.itemsize == 8 and POINT.itemsize == 16

STATE_FIELDS = [("px", np.float32), ("py", np.float32), ("pz", np.float32), ("w", np.float32),
                ("ts", np.uint16), ("track", np.uint16), ("label", np.uint8), ("status", np.uint8),
                ("forget", np.uint8), ("owner", np.uint16)]

STAGES = {"all": 0, "ego": 1, "move": 2, "remove": 3, "visibility": 4, "weight": 5, "birth": 6, "occupancy": 7}
INPUT_ON_DEVICE = 0x1
SKIP_OCCUPANCY = 0x2
NO_INSTANCES = 0x4     # sdm_update_raw: ignore the object masks (g_consider_instance == false)

STATUS_NAMES = {0: "SDM_OK", 1: "SDM_ERR_INVALID_ARGUMENT", 2: "SDM_ERR_NO_DEVICE", 3: "SDM_ERR_HIP",
                4: "SDM_ERR_CAPACITY", 5: "SDM_ERR_NOT_CONVERGED", 6: "SDM_ERR_COMM"}


class Config(C.Structure):
    _fields_ = [("x_n", C.c_int32), ("y_n", C.c_int32), ("z_n", C.c_int32), ("p_n", C.c_int32),
                ("voxel_size", C.c_float),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("width", C.c_int32), ("height", C.c_int32),
                ("depth_min", C.c_float), ("depth_max", C.c_float),
                ("window_half", C.c_int32), ("max_movable_track", C.c_int32),
                ("device", C.c_int32), ("shard_rank", C.c_int32), ("shard_count", C.c_int32),
                ("max_visible", C.c_int64)]


class Params(C.Structure):
    _fields_ = [("detection_probability", C.c_float), ("noise_number", C.c_float),
                ("nb_ptc_num_per_point", C.c_int32), ("occupancy_threshold", C.c_float),
                ("max_obersevation_lost_time", C.c_int32), ("forgetting_rate", C.c_float),
                ("max_forget_count", C.c_int32), ("match_score_threshold", C.c_float),
                ("id_transition_probability", C.c_float),
                ("if_consider_depth_noise", C.c_int32), ("if_use_independent_filter", C.c_int32),
                ("depth_noise_first_order", C.c_float), ("depth_noise_zero_order", C.c_float)]


class InstanceMask(C.Structure):
    _fields_ = [("track_id", C.c_int32), ("label_id", C.c_int32), ("mask", C.c_void_p)]


class RawOptions(C.Structure):
    _fields_ = [("src_width", C.c_int32), ("src_height", C.c_int32), ("rescale", C.c_float), ("sky_instance", C.c_int32),
                ("object_bbox", C.c_void_p)]


class RingState(C.Structure):
    _fields_ = [("global_time_stamp", C.c_uint32), ("moved_steps", C.c_int32 * 3), ("eq_steps", C.c_int32 * 3),
                ("map_center", C.c_float * 3), ("last_pos", C.c_float * 3),
                ("birth_cursor", C.c_int32), ("move_cursor", C.c_int32)]


class ColourConfig(C.Structure):
    _fields_ = [("label_bgr", (C.c_uint8 * 3) * 256), ("perm", C.c_uint8 * 256), ("background_label", C.c_int32),
                ("colour_by_label", C.c_int32), ("jet_axis", C.c_int32), ("evaluation_format", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("live_particles", C.c_int64), ("n_visible", C.c_int64), ("n_birth_attempts", C.c_int64),
                ("n_birth_success", C.c_int64), ("n_resampled_voxels", C.c_int64), ("n_moved", C.c_int64),
                ("n_move_reinserted", C.c_int64), ("n_frustum_voxels", C.c_int64), ("n_occupied", C.c_int64),
                ("flood_rounds", C.c_int64), ("bfs_start_in_frustum", C.c_int64), ("live_voxels", C.c_int64), ("sweep_live_voxels", C.c_int64),
                ("sweep_tiles", C.c_int64),
                ("stage_ms", C.c_double * 8), ("restamped_slabs", C.c_int64 * 3),
                ("graph_frames", C.c_int64), ("direct_frames", C.c_int64), ("host_enqueue_us", C.c_double),
                ("halo_dropped", C.c_int64), ("alias_entries", C.c_int64), ("alias_overflowed", C.c_int64)]


class SdmError(RuntimeError):
    pass


_lib = None


HOST_NUMA_NODE = None  # what sdm_bind_host_thread answered when the library was loaded (-1: nothing done)


def load_library():
    """Load libsdm_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SdmError("libsdm_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "or `make -C semantic_dsp_map_amd/csrc`.  There is no CPU path.")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    i32, u32, i64 = C.c_int32, C.c_uint32, C.c_int64
    sig = {
        "sdm_create": [C.POINTER(Config), C.POINTER(vp)],
        "sdm_destroy": [vp],
        "sdm_clear": [vp],
        "sdm_set_params": [vp, C.POINTER(Params)],
        "sdm_generate_noise_table": [vp, C.c_uint64, i32, C.c_float],
        "sdm_upload_noise_table": [vp, vp, i32],
        "sdm_download_noise_table": [vp, vp, i32],
        "sdm_download_pdf_table": [vp, vp, i32],
        "sdm_update": [vp, vp, vp, vp, vp, vp, i32, vp, i32, u32, i32],
        "sdm_update_raw": [vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, vp, i32, u32, i32],
        "sdm_update_raw_ex": [vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, vp, i32, u32, i32, vp],
        "sdm_get_labeled_cloud": [vp, vp],
        "sdm_host_alloc": [C.c_size_t, C.POINTER(vp)],
        "sdm_host_free": [vp],
        "sdm_update_begin": [vp, vp, vp, vp, vp, vp, i32, vp, i32, u32, i32, C.POINTER(vp)],
        "sdm_update_finish": [vp, vp, i32, u32, i32],
        "sdm_frame_start": [vp, vp, vp, vp, vp, vp, i32, vp, i32, u32, i32],
        "sdm_frame_moves": [vp],
        "sdm_frame_moves_pending": [vp, C.POINTER(C.c_int32)],
        "sdm_frame_predict": [vp, C.POINTER(vp)],
        "sdm_set_halo_buffers": [vp, vp, vp, vp, vp, i32],
        "sdm_comm_unique_id": [vp],
        "sdm_comm_init": [vp, vp, i32],
        "sdm_ipc_create": [vp, i32, vp],
        "sdm_ipc_connect": [vp, vp],
        "sdm_update_sharded": [vp, vp, vp, vp, vp, vp, i32, vp, i32, u32],
        "sdm_ck_chunk_elems": [vp, C.POINTER(i64)],
        "sdm_ck_reduce": [vp, vp, vp],
        "sdm_comm_timing": [vp, i32],
        "sdm_get_comm_times": [vp, vp],
        "sdm_device_alloc": [vp, C.c_size_t, C.POINTER(vp)],
        "sdm_device_free": [vp, vp],
        "sdm_device_upload": [vp, vp, vp, C.c_size_t],
        "sdm_device_download": [vp, vp, vp, C.c_size_t],
        "sdm_device_synchronize": [vp],
        "sdm_stream": [vp, C.POINTER(vp)],
        "sdm_set_stream": [vp, vp],
        "sdm_set_ck_buffer": [vp, vp],
        "sdm_synchronize": [vp],
        "sdm_get_voxels": [vp, vp],
        "sdm_get_occupied": [vp, vp, C.c_size_t, C.POINTER(C.c_size_t), i32],
        "sdm_get_occupied_rgb": [vp, vp, C.c_size_t, C.POINTER(C.c_size_t), i32],
        "sdm_get_freespace_rgb": [vp, vp, C.c_size_t, C.POINTER(C.c_size_t), i32],
        "sdm_set_colours": [vp, vp],
        "sdm_get_freespace": [vp, vp, C.c_size_t, C.POINTER(C.c_size_t), i32],
        "sdm_voxels_device_ptr": [vp, C.POINTER(vp)],
        "sdm_object_particle_count": [vp, i32, C.POINTER(i64)],
        "sdm_tracks_with_particles": [vp, vp, i32, C.POINTER(i32)],
        "sdm_comm_set_options": [vp, i32, i32],
        "sdm_get_stats": [vp, C.POINTER(Stats), i32],
        "sdm_set_profiling": [vp, i32],
        "sdm_debug_force_generic_flood": [vp, i32],
        "sdm_get_ring_state": [vp, C.POINTER(RingState)],
        "sdm_set_ring_state": [vp, C.POINTER(RingState)],
        "sdm_get_stamps": [vp, vp, vp, vp],
        "sdm_set_stamps": [vp, vp, vp, vp],
        "sdm_dump_state": [vp] + [vp] * 10,
        "sdm_load_state": [vp] + [vp] * 10,
        "sdm_get_ck_kappa": [vp, vp],
        "sdm_get_bin_counts": [vp, vp],
        "sdm_get_bins": [vp, vp, i64, C.POINTER(i64)],
        "sdm_get_extrinsic": [vp, vp],
        "sdm_time_occupancy_sweep": [vp, i32, C.POINTER(C.c_float)],
        "sdm_set_issue_mode": [vp, i32],
        "sdm_bind_host_thread": [i32],
        "sdm_host_numa_node_early": [i32],
        "sdm_debug_fill_dense": [vp],
        "sdm_debug_fill_dense_ex": [vp, i32],
        "sdm_debug_hinted_groups": [vp, C.POINTER(C.c_int64)],
        "sdm_debug_sweep_lists": [vp, i32],
        "sdm_debug_sweep_mode": [vp, C.POINTER(i32)],
        "sdm_debug_alias_cap": [vp, i32],
        "sdm_test_scan": [vp, vp, i64],
        "sdm_test_sort_pairs": [vp, vp, vp, vp, i64, i32],
    }
    for name, argtypes in sig.items():
        fn = getattr(L, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    L.sdm_last_error.restype = C.c_char_p
    L.sdm_version.restype = C.c_char_p
    # The process belongs on the NUMA node of its GPU before the HIP runtime makes its first allocations (sdm.h,
    # sdm_bind_host_thread; SDM_NUMA_BIND=0 switches it off): nothing of HIP has been called yet at this point.
    global HOST_NUMA_NODE
    HOST_NUMA_NODE = L.sdm_bind_host_thread(int(os.environ.get("LOCAL_RANK", "0")))
    _lib = L
    return L


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return a.ctypes.data_as(C.c_void_p)


def _check(L, rc, what):
    if rc != 0:
        raise SdmError("%s failed: %s (%s)" % (what, STATUS_NAMES.get(rc, rc), L.sdm_last_error().decode()))


class SdmMap:
    """One map on one GPU (or one Z-slab shard of a map)."""

    def __init__(self, cfg, params=None, noise_table=None, device=0, shard_rank=0, shard_count=1, max_visible=0):
        self.L = load_library()
        c = Config()
        for k, _ in Config._fields_:
            if k in cfg:
                setattr(c, k, cfg[k])
        c.device, c.shard_rank, c.shard_count, c.max_visible = device, shard_rank, shard_count, max_visible
        self.cfg = c
        h = C.c_void_p()
        _check(self.L, self.L.sdm_create(C.byref(c), C.byref(h)), "sdm_create")
        self.h = h
        self.V = 1 << (c.x_n + c.y_n + c.z_n)
        self.S = 1 << c.p_n
        self.v_count = self.V // shard_count
        self.W, self.H = c.width, c.height
        if params is not None:
            self.set_params(params)
        if noise_table is not None:
            self.upload_noise_table(noise_table)

    def close(self):
        if getattr(self, "h", None):
            self.L.sdm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear(self):
        _check(self.L, self.L.sdm_clear(self.h), "sdm_clear")

    def set_params(self, params):
        p = Params()
        for k, _ in Params._fields_:
            setattr(p, k, params[k])
        _check(self.L, self.L.sdm_set_params(self.h, C.byref(p)), "sdm_set_params")

    def generate_noise_table(self, seed=20250217, n=1000000, stddev=0.05):
        _check(self.L, self.L.sdm_generate_noise_table(self.h, seed, n, stddev), "sdm_generate_noise_table")

    def upload_noise_table(self, table):
        t = np.ascontiguousarray(table, dtype=np.float32)
        _check(self.L, self.L.sdm_upload_noise_table(self.h, _ptr(t), t.size), "sdm_upload_noise_table")

    def download_noise_table(self, n=1000000):
        t = np.empty(n, np.float32)
        _check(self.L, self.L.sdm_download_noise_table(self.h, _ptr(t), n), "sdm_download_noise_table")
        return t

    def download_pdf_table(self):
        t = np.empty(20000, np.float32)
        _check(self.L, self.L.sdm_download_pdf_table(self.h, _ptr(t), 20000), "sdm_download_pdf_table")
        return t

    def _frame_args(self, depth, cloud, cam_pos, cam_q, moves, remove_tracks, on_device):
        keep = []
        if on_device:
            dp, cp = _ptr(int(depth)), _ptr(int(cloud))
        else:
            depth = np.ascontiguousarray(depth, dtype=np.float32)
            cloud = np.ascontiguousarray(cloud, dtype=LABELED_POINT)
            assert depth.size == self.W * self.H and cloud.size == self.W * self.H
            keep += [depth, cloud]
            dp, cp = _ptr(depth), _ptr(cloud)
        pos = np.ascontiguousarray(cam_pos, dtype=np.float32)
        q = np.ascontiguousarray(cam_q, dtype=np.float32)
        mv = np.ascontiguousarray(moves if moves is not None else np.zeros(0, OBJECT_MOVE), dtype=OBJECT_MOVE)
        rm = np.ascontiguousarray(remove_tracks if remove_tracks is not None else [], dtype=np.int32)
        keep += [pos, q, mv, rm]
        return keep, (dp, cp, _ptr(pos), _ptr(q), _ptr(mv) if mv.size else None, mv.size,
                      _ptr(rm) if rm.size else None, rm.size)

    def update(self, depth, cloud, cam_pos, cam_q, moves=None, remove_tracks=None, stop_after="all",
               on_device=False, flags=0, sync=False):
        keep, args = self._frame_args(depth, cloud, cam_pos, cam_q, moves, remove_tracks, on_device)
        fl = flags | (INPUT_ON_DEVICE if on_device else 0)
        st = STAGES[stop_after] if isinstance(stop_after, str) else stop_after
        _check(self.L, self.L.sdm_update(self.h, *args, fl, st), "sdm_update")
        if sync:
            self.synchronize()

    def update_raw(self, depth, static_mask, label_to_inst, objects, cam_pos, cam_q, moves=None, remove_tracks=None,
                   stop_after="all", flags=0, sync=False, src_size=None, rescale=1.0, sky_instance=-1, object_bbox=None):
        """SURVEY row N1 on the device.  objects: list of (track_id, label_id, mask HxW uint8); pose in double."""
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        sm = None if static_mask is None else np.ascontiguousarray(static_mask, dtype=np.uint8)
        tab = np.ascontiguousarray(label_to_inst, dtype=np.uint16)
        assert tab.size == 256
        masks = [np.ascontiguousarray(o[2], dtype=np.uint8) for o in objects]
        arr = (InstanceMask * max(len(objects), 1))()
        for k, o in enumerate(objects):
            arr[k].track_id, arr[k].label_id, arr[k].mask = int(o[0]), int(o[1]), masks[k].ctypes.data
        pos = np.ascontiguousarray(cam_pos, dtype=np.float64)
        q = np.ascontiguousarray(cam_q, dtype=np.float64)
        mv = np.ascontiguousarray(moves if moves is not None else np.zeros(0, OBJECT_MOVE), dtype=OBJECT_MOVE)
        rm = np.ascontiguousarray(remove_tracks if remove_tracks is not None else [], dtype=np.int32)
        st = STAGES[stop_after] if isinstance(stop_after, str) else stop_after
        opt = RawOptions()
        opt.src_width, opt.src_height = src_size if src_size else (0, 0)
        opt.rescale = float(rescale)
        opt.sky_instance = int(sky_instance)
        bb = None if object_bbox is None else np.ascontiguousarray(object_bbox, dtype=np.float64)
        opt.object_bbox = bb.ctypes.data if bb is not None else None
        _check(self.L, self.L.sdm_update_raw_ex(self.h, _ptr(depth), _ptr(sm), _ptr(tab), C.cast(arr, C.c_void_p), len(objects),
                                                _ptr(pos), _ptr(q), _ptr(mv) if mv.size else None, mv.size,
                                                _ptr(rm) if rm.size else None, rm.size, flags, st, C.byref(opt)),
               "sdm_update_raw_ex")
        if sync:
            self.synchronize()

    def labeled_cloud(self):
        out = np.empty(self.W * self.H, LABELED_POINT)
        _check(self.L, self.L.sdm_get_labeled_cloud(self.h, _ptr(out)), "sdm_get_labeled_cloud")
        return out

    def update_begin(self, depth, cloud, cam_pos, cam_q, moves=None, remove_tracks=None, stop_after="all",
                     on_device=False, flags=0):
        keep, args = self._frame_args(depth, cloud, cam_pos, cam_q, moves, remove_tracks, on_device)
        fl = flags | (INPUT_ON_DEVICE if on_device else 0)
        st = STAGES[stop_after] if isinstance(stop_after, str) else stop_after
        ck = C.c_void_p()
        _check(self.L, self.L.sdm_update_begin(self.h, *args, fl, st, C.byref(ck)), "sdm_update_begin")
        return ck.value

    def frame_start(self, depth, cloud, cam_pos, cam_q, moves=None, remove_tracks=None, stop_after="all", on_device=False, flags=0):
        keep, args = self._frame_args(depth, cloud, cam_pos, cam_q, moves, remove_tracks, on_device)
        fl = flags | (INPUT_ON_DEVICE if on_device else 0)
        st = STAGES[stop_after] if isinstance(stop_after, str) else stop_after
        _check(self.L, self.L.sdm_frame_start(self.h, *args, fl, st), "sdm_frame_start")

    def frame_moves(self):
        _check(self.L, self.L.sdm_frame_moves(self.h), "sdm_frame_moves")

    def frame_moves_pending(self):
        """True: a further batch of a long object list has published its counts and waits for their exchange"""
        n = C.c_int32()
        _check(self.L, self.L.sdm_frame_moves_pending(self.h, C.byref(n)), "sdm_frame_moves_pending")
        return bool(n.value)

    def frame_predict(self):
        ck = C.c_void_p()
        _check(self.L, self.L.sdm_frame_predict(self.h, C.byref(ck)), "sdm_frame_predict")
        return ck.value

    def set_halo_buffers(self, counts_local, counts_all, send, recv_all, cap_records):
        _check(self.L, self.L.sdm_set_halo_buffers(self.h, _ptr(int(counts_local)), _ptr(int(counts_all)), _ptr(int(send)),
                                                    _ptr(int(recv_all)), cap_records), "sdm_set_halo_buffers")

    def comm_init(self, id_bytes, halo_cap=0):
        buf = np.frombuffer(bytes(id_bytes), np.uint8).copy()
        assert buf.size == 128
        _check(self.L, self.L.sdm_comm_init(self.h, _ptr(buf), halo_cap), "sdm_comm_init")

    def ipc_create(self, halo_cap=0):
        """This shard's receive arena for the exchanges without RCCL; returns its 64-byte hipIpc handle (sdm_ipc_create)."""
        buf = np.zeros(64, np.uint8)
        _check(self.L, self.L.sdm_ipc_create(self.h, halo_cap, _ptr(buf)), "sdm_ipc_create")
        return buf.tobytes()

    def ipc_connect(self, handles_all):
        """handles_all: the handles of all shards, in shard order (shard_count x 64 bytes)"""
        buf = np.frombuffer(bytes(handles_all), np.uint8).copy()
        _check(self.L, self.L.sdm_ipc_connect(self.h, _ptr(buf)), "sdm_ipc_connect")

    def comm_set_options(self, ck_exchange=-1, timeout_ms=0):
        """ck_exchange: 0 chunk-owner reduction, 1 one all-gather of the whole partial images, -1 unchanged"""
        _check(self.L, self.L.sdm_comm_set_options(self.h, ck_exchange, timeout_ms), "sdm_comm_set_options")

    def update_sharded(self, depth, cloud, cam_pos, cam_q, moves=None, remove_tracks=None, on_device=False, flags=0):
        keep, args = self._frame_args(depth, cloud, cam_pos, cam_q, moves, remove_tracks, on_device)
        fl = flags | (INPUT_ON_DEVICE if on_device else 0)
        _check(self.L, self.L.sdm_update_sharded(self.h, *args, fl), "sdm_update_sharded")

    def ck_chunk_elems(self):
        """pixels per shard of the chunk-owner ck exchange (sdm_ck_chunk_elems)"""
        n = C.c_int64()
        _check(self.L, self.L.sdm_ck_chunk_elems(self.h, C.byref(n)), "sdm_ck_chunk_elems")
        return int(n.value)

    def ck_reduce(self, stage_dev, full_dev):
        _check(self.L, self.L.sdm_ck_reduce(self.h, _ptr(int(stage_dev)), _ptr(int(full_dev))), "sdm_ck_reduce")

    def comm_timing(self, on=True):
        _check(self.L, self.L.sdm_comm_timing(self.h, 1 if on else 0), "sdm_comm_timing")

    def comm_times(self):
        """GPU microseconds of the last sharded frame's collectives: counts, halo, ck all-to-all, ck all-gather."""
        out = np.zeros(4, np.float64)
        _check(self.L, self.L.sdm_get_comm_times(self.h, _ptr(out)), "sdm_get_comm_times")
        return dict(zip(("counts_allgather", "halo_alltoall", "ck_alltoall", "ck_allgather"), [float(x) for x in out]))

    def device_alloc(self, nbytes):
        p = C.c_void_p()
        _check(self.L, self.L.sdm_device_alloc(self.h, nbytes, C.byref(p)), "sdm_device_alloc")
        return p.value

    def device_free(self, ptr):
        _check(self.L, self.L.sdm_device_free(self.h, _ptr(int(ptr))), "sdm_device_free")

    def device_upload(self, ptr, array):
        a = np.ascontiguousarray(array)
        _check(self.L, self.L.sdm_device_upload(self.h, _ptr(int(ptr)), _ptr(a), a.nbytes), "sdm_device_upload")

    def device_put(self, array):
        a = np.ascontiguousarray(array)
        p = self.device_alloc(a.nbytes)
        self.device_upload(p, a)
        return p

    def device_download(self, ptr, nbytes, dtype=np.uint8):
        out = np.empty(nbytes // np.dtype(dtype).itemsize, dtype)
        _check(self.L, self.L.sdm_device_download(self.h, _ptr(out), _ptr(int(ptr)), out.nbytes), "sdm_device_download")
        return out

    def device_synchronize(self):
        _check(self.L, self.L.sdm_device_synchronize(self.h), "sdm_device_synchronize")

    def update_finish(self, ck_parts_dev=None, n_parts=1, stop_after="all", flags=0):
        st = STAGES[stop_after] if isinstance(stop_after, str) else stop_after
        _check(self.L, self.L.sdm_update_finish(self.h, _ptr(ck_parts_dev), n_parts, flags, st), "sdm_update_finish")

    def stream(self):
        s = C.c_void_p()
        _check(self.L, self.L.sdm_stream(self.h, C.byref(s)), "sdm_stream")
        return s.value or 0

    def set_stream(self, hip_stream):
        _check(self.L, self.L.sdm_set_stream(self.h, _ptr(int(hip_stream)) if hip_stream else None), "sdm_set_stream")

    def set_ck_buffer(self, dev_ptr):
        _check(self.L, self.L.sdm_set_ck_buffer(self.h, _ptr(int(dev_ptr)) if dev_ptr else None), "sdm_set_ck_buffer")

    def set_issue_mode(self, mode):
        """0 launch by launch, 1 branched graph, 2 by the host's speed, 3 chain graph, 4 five chain graphs (sdm.h SDM_ISSUE_*)"""
        _check(self.L, self.L.sdm_set_issue_mode(self.h, int(mode)), "sdm_set_issue_mode")

    def synchronize(self):
        _check(self.L, self.L.sdm_synchronize(self.h), "sdm_synchronize")

    def voxels(self):
        out = np.empty(self.v_count, VOXEL_RESULT)
        _check(self.L, self.L.sdm_get_voxels(self.h, _ptr(out)), "sdm_get_voxels")
        return out

    def occupied(self, cap=None, zero_center=False, free=False, mark_fov=False):
        cap = cap or self.v_count
        out = np.empty(cap, POINT)
        n = C.c_size_t()
        fn = self.L.sdm_get_freespace if free else self.L.sdm_get_occupied
        _check(self.L, fn(self.h, _ptr(out), cap, C.byref(n), (1 if zero_center else 0) | (2 if mark_fov else 0)),
               "sdm_get_occupied")
        return out[:min(n.value, cap)], n.value

    def set_colours(self, label_bgr, perm, background_label=0, colour_by_label=False, jet_axis=0, evaluation_format=False):
        """label_bgr: (256, 3) uint8 BGR per label id; perm: 256 uint8 (color_map_int_256_)."""
        c = ColourConfig()
        lb = np.ascontiguousarray(label_bgr, np.uint8).reshape(256, 3)
        pm = np.ascontiguousarray(perm, np.uint8).reshape(256)
        C.memmove(c.label_bgr, lb.ctypes.data, 768)
        C.memmove(c.perm, pm.ctypes.data, 256)
        c.background_label, c.colour_by_label = int(background_label), 1 if colour_by_label else 0
        c.jet_axis, c.evaluation_format = int(jet_axis), 1 if evaluation_format else 0
        _check(self.L, self.L.sdm_set_colours(self.h, C.byref(c)), "sdm_set_colours")

    def occupied_rgb(self, cap=None, zero_center=False, free=False):
        """getOccupancyResult's cloud coloured and packed on the device: POINT_XYZRGB records."""
        cap = cap or self.v_count
        out = np.empty(cap, POINT_XYZRGB)
        n = C.c_size_t()
        fn = self.L.sdm_get_freespace_rgb if free else self.L.sdm_get_occupied_rgb
        _check(self.L, fn(self.h, _ptr(out), cap, C.byref(n), 1 if zero_center else 0), "sdm_get_occupied_rgb")
        return out[:min(n.value, cap)], n.value

    def object_particle_count(self, track):
        n = C.c_int64()
        _check(self.L, self.L.sdm_object_particle_count(self.h, track, C.byref(n)), "sdm_object_particle_count")
        return n.value

    def tracks_with_particles(self):
        """Track ids that own at least one slot, ascending (the non-empty keys of the reference's indices_map)."""
        out = np.zeros(65536, np.int32)
        n = C.c_int32(0)
        _check(self.L, self.L.sdm_tracks_with_particles(self.h, _ptr(out), out.size, C.byref(n)), "sdm_tracks_with_particles")
        return out[:n.value].copy()

    def stats_unchecked(self):
        """sdm_get_stats without raising on the status it returns (it reports the frame's capacity errors and fills the
        structure all the same)"""
        return self.stats(check=False)

    def stats(self, count_live=False, check=True):
        s = Stats()
        rc = self.L.sdm_get_stats(self.h, C.byref(s), 1 if count_live else 0)
        if check:
            _check(self.L, rc, "sdm_get_stats")
        d = {k: getattr(s, k) for k, _ in Stats._fields_ if k not in ("stage_ms", "restamped_slabs")}
        d["stage_ms"] = list(s.stage_ms)
        d["restamped_slabs"] = list(s.restamped_slabs)
        return d

    def set_profiling(self, on=True):
        _check(self.L, self.L.sdm_set_profiling(self.h, 1 if on else 0), "sdm_set_profiling")

    def force_generic_flood(self, on=True):
        _check(self.L, self.L.sdm_debug_force_generic_flood(self.h, 1 if on else 0), "sdm_debug_force_generic_flood")

    def ring_state(self):
        r = RingState()
        _check(self.L, self.L.sdm_get_ring_state(self.h, C.byref(r)), "sdm_get_ring_state")
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
        _check(self.L, self.L.sdm_set_ring_state(self.h, C.byref(r)), "sdm_set_ring_state")

    def stamps(self):
        c = self.cfg
        sx = np.empty(1 << c.x_n, np.uint32)
        sy = np.empty(1 << c.y_n, np.uint32)
        sz = np.empty(1 << c.z_n, np.uint32)
        _check(self.L, self.L.sdm_get_stamps(self.h, _ptr(sx), _ptr(sy), _ptr(sz)), "sdm_get_stamps")
        return sx, sy, sz

    def set_stamps(self, sx, sy, sz):
        sx, sy, sz = (np.ascontiguousarray(a, dtype=np.uint32) for a in (sx, sy, sz))
        _check(self.L, self.L.sdm_set_stamps(self.h, _ptr(sx), _ptr(sy), _ptr(sz)), "sdm_set_stamps")

    def dump_state(self):
        n = self.v_count * self.S
        st = {k: np.empty(n, dt) for k, dt in STATE_FIELDS}
        _check(self.L, self.L.sdm_dump_state(self.h, *[_ptr(st[k]) for k, _ in STATE_FIELDS]), "sdm_dump_state")
        return st

    def load_state(self, st):
        arrs = [np.ascontiguousarray(st[k], dtype=dt) for k, dt in STATE_FIELDS]
        _check(self.L, self.L.sdm_load_state(self.h, *[_ptr(a) for a in arrs]), "sdm_load_state")

    def ck_kappa(self):
        out = np.empty(self.W * self.H, np.float32)
        _check(self.L, self.L.sdm_get_ck_kappa(self.h, _ptr(out)), "sdm_get_ck_kappa")
        return out.reshape(self.H, self.W)

    def bin_counts(self):
        out = np.empty(self.W * self.H, np.uint32)
        _check(self.L, self.L.sdm_get_bin_counts(self.h, _ptr(out)), "sdm_get_bin_counts")
        return out.reshape(self.H, self.W)

    def bins(self):
        n = C.c_int64()
        _check(self.L, self.L.sdm_get_bins(self.h, None, 0, C.byref(n)), "sdm_get_bins")
        out = np.empty(max(n.value, 1), np.uint32)
        _check(self.L, self.L.sdm_get_bins(self.h, _ptr(out), n.value, C.byref(n)), "sdm_get_bins")
        return out[:n.value]

    def extrinsic(self):
        out = np.empty(16, np.float32)
        _check(self.L, self.L.sdm_get_extrinsic(self.h, _ptr(out)), "sdm_get_extrinsic")
        return out.reshape(4, 4)

    def fill_dense_ex(self, mode):
        _check(self.L, self.L.sdm_debug_fill_dense_ex(self.h, mode), "sdm_debug_fill_dense_ex")

    def hinted_groups(self):
        n = C.c_int64()
        _check(self.L, self.L.sdm_debug_hinted_groups(self.h, C.byref(n)), "sdm_debug_hinted_groups")
        return n.value

    def force_sweep_lists(self, mode):
        """Test hook: 1 / 0 = the non-incremental sweeps always / never hand their sparse voxels to per-tile lists, -1 = the
        library picks per sweep (sdm_debug_sweep_lists)."""
        _check(self.L, self.L.sdm_debug_sweep_lists(self.h, int(mode)), "sdm_debug_sweep_lists")

    def sweep_mode(self):
        """Test hook: bit 0 - the next non-incremental sweep uses per-tile lists; bit 1 - it is one launch (every group of
        512 voxels was dense in the last one) (sdm_debug_sweep_mode)."""
        v = C.c_int32(0)
        _check(self.L, self.L.sdm_debug_sweep_mode(self.h, C.byref(v)), "sdm_debug_sweep_mode")
        return v.value

    def set_alias_cap(self, cap):
        """Test hook: the table of older owner-set memberships reports its overflow at `cap` entries (before the first frame)."""
        _check(self.L, self.L.sdm_debug_alias_cap(self.h, int(cap)), "sdm_debug_alias_cap")

    def fill_dense(self):
        _check(self.L, self.L.sdm_debug_fill_dense(self.h), "sdm_debug_fill_dense")

    def time_occupancy_sweep(self, iters=20):
        ms = C.c_float()
        _check(self.L, self.L.sdm_time_occupancy_sweep(self.h, iters, C.byref(ms)), "sdm_time_occupancy_sweep")
        return ms.value


def comm_unique_id():
    """128-byte RCCL id drawn on this process (rank 0 broadcasts it to the other ranks)."""
    L = load_library()
    buf = np.zeros(128, np.uint8)
    _check(L, L.sdm_comm_unique_id(_ptr(buf)), "sdm_comm_unique_id")
    return buf.tobytes()


def test_scan(a):
    L = load_library()
    a = np.ascontiguousarray(a, dtype=np.uint32)
    out = np.empty_like(a)
    _check(L, L.sdm_test_scan(_ptr(a), _ptr(out), a.size), "sdm_test_scan")
    return out


def test_sort_pairs(keys, vals, nbits):
    L = load_library()
    keys = np.ascontiguousarray(keys, dtype=np.uint32)
    vals = np.ascontiguousarray(vals, dtype=np.uint32)
    ko, vo = np.empty_like(keys), np.empty_like(vals)
    _check(L, L.sdm_test_sort_pairs(_ptr(keys), _ptr(vals), _ptr(ko), _ptr(vo), keys.size, nbits), "sdm_test_sort_pairs")
    return ko, vo
