"""Synthetic depth + segmentation frames for tests and bench.py.

There is no dataset in the image (the reference's rosbags are external
downloads), so frames are ray-cast from a small analytic street scene:
ground plane, two side walls, static boxes and constant-velocity "car" boxes.
The output has exactly the layout the hot path consumes: a float32 depth
image (metres, camera-frame z) and a LabeledPoint[H][W] image
(reference include/utils/data_base.h:78-92), plus the camera pose and the list
of per-object 4x4 motions the object layer would hand to the particle update
(reference include/semantic_dsp_map.h:673-693).

World frame: x right, y down, z forward (camera frame at identity pose).
Pure numpy; deterministic for a given seed.
"""
import math

import numpy as np

LABELED_POINT = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("sigma", "<f4"),
                          ("track_id", "<u2"), ("label_id", "u1"), ("is_valid", "u1")])
OBJECT_MOVE = np.dtype([("track_id", "<i4"), ("T", "<f4", (16,))])

# label ids / static instance ids of the reference's cfg/object_info.csv
LABEL_ROAD, TRACK_ROAD = 8, 65528
LABEL_BUILDING, TRACK_BUILDING = 7, 65529
LABEL_POLE, TRACK_POLE = 12, 65524
LABEL_TREE, TRACK_TREE = 5, 65531
LABEL_VEGETATION, TRACK_VEGETATION = 6, 65530
LABEL_CAR = 15
MAX_MOVABLE_TRACK = 65522  # min static instance id (Misc = 65523) - 1, object_info_handler.h:49-69,84

# ---------------------------------------------------------------- presets
# grid / camera presets: BASELINE.json configs C1..C5
_KITTI360 = dict(fx=552.554261, fy=552.554261, cx=682.049453, cy=238.769549, width=1408, height=376,
                 depth_min=0.3, depth_max=30.0, voxel_size=0.15, window_half=5)        # settings.h:32-52
_ZED2_BOOST = dict(fx=0.5 * 527.8191528320312, fy=0.5 * 527.8191528320312, cx=0.5 * 633.9357299804688,
                   cy=0.5 * 366.3338623046875, width=640, height=360, depth_min=0.3, depth_max=15.0,
                   voxel_size=0.15, window_half=3)                                      # settings.h:101-141
_VKITTI2 = dict(fx=725.0087, fy=725.0087, cx=620.5, cy=187.0, width=1242, height=375,
                depth_min=0.3, depth_max=30.0, voxel_size=0.2, window_half=5)          # settings.h:79-98

_CODA = dict(fx=569.8286, fy=565.4818, cx=439.2660, cy=360.5810, width=960, height=540,
             depth_min=0.3, depth_max=10.0, voxel_size=0.15, window_half=5)                # settings.h:54-77
_ZED2_FULL = dict(fx=527.8191528320312, fy=527.8191528320312, cx=633.9357299804688, cy=366.3338623046875, width=1280,
                  height=720, depth_min=0.3, depth_max=15.0, voxel_size=0.15, window_half=5)  # settings.h:100-119, BOOST_MODE 0

CONFIGS = {
    # the grids the reference ships (settings/settings.h:32-143, one per SETTING value; BASELINE.json's cubes are below)
    "REF_KITTI360": dict(x_n=8, y_n=8, z_n=8, p_n=3, max_movable_track=MAX_MOVABLE_TRACK, **_KITTI360),     # SETTING 0
    "REF_CODA": dict(x_n=8, y_n=8, z_n=7, p_n=2, max_movable_track=MAX_MOVABLE_TRACK, **_CODA),             # SETTING 1
    "REF_VKITTI2": dict(x_n=8, y_n=7, z_n=8, p_n=3, max_movable_track=MAX_MOVABLE_TRACK, **_VKITTI2),       # SETTING 2
    "REF_ZED2_BOOST": dict(x_n=7, y_n=5, z_n=7, p_n=2, max_movable_track=MAX_MOVABLE_TRACK, **_ZED2_BOOST), # SETTING 3 (shipped)
    "REF_ZED2_SENSOR": dict(x_n=7, y_n=5, z_n=7, p_n=2, max_movable_track=MAX_MOVABLE_TRACK, **_ZED2_FULL), # its 1280x720 inputs
    "C1": dict(x_n=6, y_n=6, z_n=6, p_n=3, max_movable_track=MAX_MOVABLE_TRACK, **_KITTI360),
    "C2": dict(x_n=7, y_n=7, z_n=7, p_n=2, max_movable_track=MAX_MOVABLE_TRACK, **_ZED2_BOOST),
    "C3": dict(x_n=8, y_n=8, z_n=8, p_n=3, max_movable_track=MAX_MOVABLE_TRACK, **_VKITTI2),
    "C4": dict(x_n=8, y_n=8, z_n=8, p_n=3, max_movable_track=MAX_MOVABLE_TRACK, **_VKITTI2),
    "C5": dict(x_n=9, y_n=9, z_n=9, p_n=3, max_movable_track=MAX_MOVABLE_TRACK, **_VKITTI2),
    # small configurations for fast parity tests
    "T0": dict(x_n=5, y_n=5, z_n=5, p_n=3, max_movable_track=MAX_MOVABLE_TRACK,
               fx=80.0, fy=80.0, cx=64.0, cy=40.0, width=128, height=80, depth_min=0.3, depth_max=12.0,
               voxel_size=0.4, window_half=3),
    "T1": dict(x_n=6, y_n=5, z_n=6, p_n=2, max_movable_track=MAX_MOVABLE_TRACK,
               fx=120.0, fy=120.0, cx=96.0, cy=54.0, width=192, height=108, depth_min=0.3, depth_max=15.0,
               voxel_size=0.3, window_half=5),
}

# parameter presets: reference cfg/options_*.yaml
PARAMS = {
    "kitti360": dict(detection_probability=0.6, noise_number=0.6, nb_ptc_num_per_point=1, occupancy_threshold=0.1,
                     max_obersevation_lost_time=10, forgetting_rate=0.5, max_forget_count=3,
                     match_score_threshold=0.6, id_transition_probability=0.2, if_consider_depth_noise=1,
                     if_use_independent_filter=1, depth_noise_first_order=0.01, depth_noise_zero_order=0.1),
    "zed2": dict(detection_probability=0.8, noise_number=0.2, nb_ptc_num_per_point=1, occupancy_threshold=0.15,
                 max_obersevation_lost_time=20, forgetting_rate=1.0, max_forget_count=5,
                 match_score_threshold=0.6, id_transition_probability=0.5, if_consider_depth_noise=1,
                 if_use_independent_filter=0, depth_noise_first_order=0.02, depth_noise_zero_order=0.3),
    # cfg/options.yaml (the CODA run)
    "coda": dict(detection_probability=0.6, noise_number=0.6, nb_ptc_num_per_point=1, occupancy_threshold=0.1,
                 max_obersevation_lost_time=10, forgetting_rate=0.5, max_forget_count=3,
                 match_score_threshold=0.6, id_transition_probability=0.2, if_consider_depth_noise=1,
                 if_use_independent_filter=1, depth_noise_first_order=0.01, depth_noise_zero_order=0.2),
    "vkitti2": dict(detection_probability=0.98, noise_number=0.001, nb_ptc_num_per_point=1, occupancy_threshold=0.5,
                    max_obersevation_lost_time=5, forgetting_rate=1.0, max_forget_count=3,
                    match_score_threshold=0.6, id_transition_probability=0.2, if_consider_depth_noise=1,
                    if_use_independent_filter=0, depth_noise_first_order=0.01, depth_noise_zero_order=0.2),
    # the VKITTI2 preset with the constructor's default of three noisy births per point (semantic_dsp_map.h:29): thick
    # surfaces, several times the particles per frame - bench.py's busy scene
    "vkitti2_nb3": dict(detection_probability=0.98, noise_number=0.001, nb_ptc_num_per_point=3, occupancy_threshold=0.5,
                        max_obersevation_lost_time=5, forgetting_rate=1.0, max_forget_count=3,
                        match_score_threshold=0.6, id_transition_probability=0.2, if_consider_depth_noise=1,
                        if_use_independent_filter=0, depth_noise_first_order=0.01, depth_noise_zero_order=0.2),
    # exercises the Gaussian-noise birth path (nb > 1) and the no-noise birth path
    "noisy3": dict(detection_probability=0.95, noise_number=0.1, nb_ptc_num_per_point=3, occupancy_threshold=0.2,
                   max_obersevation_lost_time=5, forgetting_rate=1.0, max_forget_count=5,
                   match_score_threshold=0.3, id_transition_probability=0.1, if_consider_depth_noise=1,
                   if_use_independent_filter=0, depth_noise_first_order=0.01, depth_noise_zero_order=0.1),
    "nodepthnoise": dict(detection_probability=1.0, noise_number=0.001, nb_ptc_num_per_point=3,
                         occupancy_threshold=0.1, max_obersevation_lost_time=10, forgetting_rate=1.0,
                         max_forget_count=5, match_score_threshold=0.3, id_transition_probability=0.1,
                         if_consider_depth_noise=0, if_use_independent_filter=0, depth_noise_first_order=0.0,
                         depth_noise_zero_order=0.1),
}
CONFIG_PARAMS = {"REF_KITTI360": "kitti360", "REF_CODA": "coda", "REF_VKITTI2": "vkitti2", "REF_ZED2_BOOST": "zed2",
                 "REF_ZED2_SENSOR": "zed2", "C1": "kitti360", "C2": "zed2", "C3": "vkitti2", "C4": "vkitti2", "C5": "vkitti2",
                 "T0": "vkitti2", "T1": "zed2"}


# bench.py's `driven` leg and tests/test_driven_gpu.py: an EMPTY C3 map and a 66 m drive (220 frames of 0.3 m) down a 28 m
# wide street with 400 static and 12 moving boxes, three noisy births per point; the camera yaws 0.1 deg and drifts 6 mm
# sideways per frame (ring shifts on z every frame or two, on x every 33 frames), the moving boxes drive the camera's
# way at 0.22-0.4 m per frame.  Nothing is prefilled: every particle of the map is one the filter put there.
DRIVEN_SCENE = dict(n_static=400, n_dynamic=12, seed=17, yaw_rate_deg=0.1, street_half_width=14.0, street_length=130.0,
                    dyn_all_forward=True, dyn_speed=(0.22, 0.4), lane_half_width=3.5, lateral_rate=0.02)
DRIVEN_FRAMES = 220
DRIVEN_PARAMS = "vkitti2_nb3"


def noise_table(seed=20250217, n=1000000, stddev=0.05):
    """Host-side stand-in for the reference's gaussian_randoms table (basic_algorithms.h:394-402).
    On the GPU box the table is generated by rocRAND inside the library and read back; this numpy
    table is used for CPU-only tests."""
    rng = np.random.default_rng(seed)
    return (rng.standard_normal(n) * stddev).astype(np.float32)


def yaw_quat(theta):
    """(w, x, y, z) of a rotation by theta about the y (down) axis."""
    return np.array([math.cos(theta / 2), 0.0, math.sin(theta / 2), 0.0], np.float64)


def quat_to_R(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], np.float64)


class Scene:
    """Street scene scaled to the map extent of a configuration."""

    def __init__(self, cfg, n_static=24, n_dynamic=4, seed=7, speed=0.3, yaw_rate_deg=1.0,
                 invalid_fraction=0.0, dyn_speed=(0.2, 1.0), lateral_extra=None, street_half_width=None, street_length=None,
                 dyn_all_forward=False, lane_half_width=None, lateral_rate=0.05):
        """street_half_width / street_length / dyn_all_forward / lane_half_width / lateral_rate (all at their old values by
        default: the scenes of the committed fixtures do not change): a wider street, static boxes spread over
        `street_length` metres ahead instead of the map's extent and kept `lane_half_width` metres off the centre line,
        every moving box driving the camera's way, the camera's sideways drift per metre driven - what a drive longer
        than the map needs without running the camera into the clutter (bench.py `driven`)."""
        self.cfg = dict(cfg)
        self.rng = np.random.default_rng(seed)
        self.seed = seed
        half = 0.5 * (1 << min(cfg["x_n"], cfg["z_n"])) * cfg["voxel_size"]
        self.half = half
        self.ground_y = min(1.6, 0.35 * (1 << cfg["y_n"]) * cfg["voxel_size"])
        self.wall_x = min(8.0, 0.8 * half) if street_half_width is None else float(street_half_width)
        self.wall_top = -min(6.0, 0.45 * (1 << cfg["y_n"]) * cfg["voxel_size"])
        self.speed = speed
        self.yaw_rate = math.radians(yaw_rate_deg)
        self.lateral_extra = lateral_extra  # (t0, metres per frame): extra sideways (x) motion from frame t0 on
        self.lateral_rate = lateral_rate
        self.invalid_fraction = invalid_fraction
        zmax = max(2.0 * half, 6.0) if street_length is None else float(street_length)
        r = self.rng
        boxes, labels, tracks = [], [], []
        static_kinds = [(LABEL_POLE, TRACK_POLE, (0.3, 3.0, 0.3)), (LABEL_TREE, TRACK_TREE, (1.5, 4.0, 1.5)),
                        (LABEL_VEGETATION, TRACK_VEGETATION, (2.0, 1.0, 2.0)),
                        (LABEL_BUILDING, TRACK_BUILDING, (3.0, 5.0, 3.0))]
        for k in range(n_static):
            lab, trk, (sx, sy, sz) = static_kinds[k % len(static_kinds)]
            s = min(1.0, half / 12.0)
            sx, sy, sz = sx * s, min(sy * s, self.ground_y - self.wall_top), sz * s
            cx = r.uniform(-0.85 * self.wall_x, 0.85 * self.wall_x)
            cz = r.uniform(1.5, zmax)
            lane_free = 1.2 * s if lane_half_width is None else float(lane_half_width)
            if abs(cx) < lane_free + sx / 2:  # keep the driving lane free
                cx = math.copysign(lane_free + sx / 2 + 0.2, cx if cx != 0 else 1.0)
            boxes.append([cx - sx / 2, self.ground_y - sy, cz - sz / 2, cx + sx / 2, self.ground_y, cz + sz / 2])
            labels.append(lab)
            tracks.append(trk)
        self.static_boxes = np.array(boxes, np.float64).reshape(-1, 6)
        self.static_labels = np.array(labels, np.uint8)
        self.static_tracks = np.array(tracks, np.uint16)
        dyn, vel = [], []
        for k in range(n_dynamic):
            s = min(1.0, half / 12.0)
            sx, sy, sz = 1.8 * s, 1.5 * s, 4.0 * s
            lane = (-1) ** k * (1.2 * s + sx / 2 + 0.3)
            cz = r.uniform(3.0 * s + 1.0, max(0.9 * half, 4.0))
            dyn.append([lane - sx / 2, self.ground_y - sy, cz - sz / 2, lane + sx / 2, self.ground_y, cz + sz / 2])
            v = r.uniform(*dyn_speed) * s
            vel.append([0.0, 0.0, v if (k % 2 == 0 or dyn_all_forward) else -0.5 * v])
        self.dyn_boxes0 = np.array(dyn, np.float64).reshape(-1, 6)
        self.dyn_vel = np.array(vel, np.float64).reshape(-1, 3)
        self.dyn_tracks = np.arange(1, n_dynamic + 1, dtype=np.uint16)

    # ------------------------------------------------------------ trajectory
    def pose(self, t):
        theta = self.yaw_rate * t
        pos = np.array([self.lateral_rate * t * self.speed, 0.0, self.speed * t], np.float64)
        if self.lateral_extra is not None and t > self.lateral_extra[0]:
            pos[0] += (t - self.lateral_extra[0]) * self.lateral_extra[1]
        return pos, yaw_quat(theta)

    def dyn_boxes(self, t):
        b = self.dyn_boxes0.copy()
        if len(b):
            d = self.dyn_vel * t
            b[:, 0:3] += d
            b[:, 3:6] += d
        return b

    def moves(self, t):
        """Motions from frame t-1 to t of the dynamic objects (pure translations)."""
        out = np.zeros(len(self.dyn_tracks) if t > 0 else 0, OBJECT_MOVE)
        for k in range(len(out)):
            T = np.eye(4, dtype=np.float64)
            T[0:3, 3] = self.dyn_vel[k]
            out[k]["track_id"] = int(self.dyn_tracks[k])
            out[k]["T"] = T.astype(np.float32).reshape(16)
        return out

    # -------------------------------------------------------------- render
    def render(self, t, params, fast=True):
        """Returns depth (H,W) f32, cloud (H*W,) LABELED_POINT, cam_pos (3,) f32, cam_q (4,) f32.
        fast: a box is only tested against the rays of the image rectangle its eight corners project into (+ 2 pixels);
        the per-ray arithmetic is the same, so is the result (tests/test_synth.py holds the two paths equal)."""
        c = self.cfg
        W, H = c["width"], c["height"]
        pos, q = self.pose(t)
        R = quat_to_R(q)
        jj, ii = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
        dc = np.stack([(jj - c["cx"]) / c["fx"], (ii - c["cy"]) / c["fy"], np.ones_like(jj)], -1).reshape(-1, 3)
        # explicit products (no BLAS) so that frames are bit-reproducible on every host
        dw = np.stack([dc[:, 0] * R[a, 0] + dc[:, 1] * R[a, 1] + dc[:, 2] * R[a, 2] for a in range(3)], -1)
        n = dw.shape[0]
        best_t = np.full(n, 1000.0)
        best_label = np.zeros(n, np.uint8)
        best_track = np.full(n, 65535, np.uint16)

        def take(tt, ok, label, track):
            nonlocal best_t
            m = ok & (tt > 1e-6) & (tt < best_t)
            best_t = np.where(m, tt, best_t)
            best_label[m] = label
            best_track[m] = track

        def rect_of(box):
            """pixel rectangle (i0, i1, j0, j1) that contains every ray hitting the box, or None = the whole image"""
            cs = np.array([[box[x], box[y], box[z]] for x in (0, 3) for y in (1, 4) for z in (2, 5)], np.float64) - pos
            pc = cs @ R                                   # camera frame: R^T (corner - pos)
            if np.min(pc[:, 2]) <= 0.05:
                return None
            u = c["fx"] * pc[:, 0] / pc[:, 2] + c["cx"]
            v = c["fy"] * pc[:, 1] / pc[:, 2] + c["cy"]
            j0, j1 = int(np.floor(u.min())) - 2, int(np.ceil(u.max())) + 3
            i0, i1 = int(np.floor(v.min())) - 2, int(np.ceil(v.max())) + 3
            j0, j1, i0, i1 = max(j0, 0), min(j1, W), max(i0, 0), min(i1, H)
            return (i0, i1, j0, j1)

        with np.errstate(divide="ignore", invalid="ignore"):
            tg = (self.ground_y - pos[1]) / dw[:, 1]
            take(tg, dw[:, 1] > 1e-9, LABEL_ROAD, TRACK_ROAD)
            for sx in (-self.wall_x, self.wall_x):
                tw = (sx - pos[0]) / dw[:, 0]
                hy = pos[1] + tw * dw[:, 1]
                take(tw, (np.abs(dw[:, 0]) > 1e-9) & (hy >= self.wall_top) & (hy <= self.ground_y),
                     LABEL_BUILDING, TRACK_BUILDING)
            boxes = [(self.static_boxes, self.static_labels, self.static_tracks),
                     (self.dyn_boxes(t), np.full(len(self.dyn_tracks), LABEL_CAR, np.uint8), self.dyn_tracks)]
            inv = 1.0 / dw
            inv2 = inv.reshape(H, W, 3)
            bt2, bl2, bk2 = best_t.reshape(H, W), best_label.reshape(H, W), best_track.reshape(H, W)
            for bx, labs, trks in boxes:
                for k in range(len(bx)):
                    r = rect_of(bx[k]) if fast else None
                    if r is None:
                        t0 = (bx[k, 0:3] - pos) * inv
                        t1 = (bx[k, 3:6] - pos) * inv
                        tn = np.nanmax(np.minimum(t0, t1), axis=1)
                        tf = np.nanmin(np.maximum(t0, t1), axis=1)
                        best_t = bt2.reshape(-1)
                        take(tn, (tn <= tf) & (tf > 0), labs[k], trks[k])
                        bt2 = best_t.reshape(H, W)
                        continue
                    i0, i1, j0, j1 = r
                    if i0 >= i1 or j0 >= j1:
                        continue                          # out of the picture
                    iv = inv2[i0:i1, j0:j1]
                    t0 = (bx[k, 0:3] - pos) * iv
                    t1 = (bx[k, 3:6] - pos) * iv
                    tn = np.nanmax(np.minimum(t0, t1), axis=2)
                    tf = np.nanmin(np.maximum(t0, t1), axis=2)
                    cur = bt2[i0:i1, j0:j1]
                    m = (tn <= tf) & (tf > 0) & (tn > 1e-6) & (tn < cur)
                    bt2[i0:i1, j0:j1] = np.where(m, tn, cur)
                    bl2[i0:i1, j0:j1][m] = labs[k]
                    bk2[i0:i1, j0:j1][m] = trks[k]
            best_t = bt2.reshape(-1)

        depth = best_t.astype(np.float32)  # camera-frame z == ray parameter (dc.z == 1)
        if self.invalid_fraction > 0:
            r = np.random.default_rng(self.seed * 1000003 + t)
            bad = r.random(n) < self.invalid_fraction
            depth = np.where(bad, np.float32(np.nan), depth)
        valid = (~np.isnan(depth)) & (depth >= np.float32(c["depth_min"])) & (depth <= np.float32(c["depth_max"]))
        p = pos[None, :] + best_t[:, None] * dw
        cloud = np.zeros(n, LABELED_POINT)
        cloud["x"] = np.where(valid, p[:, 0], 0.0).astype(np.float32)
        cloud["y"] = np.where(valid, p[:, 1], 0.0).astype(np.float32)
        cloud["z"] = np.where(valid, p[:, 2], 0.0).astype(np.float32)
        zero = np.float32(params["depth_noise_zero_order"])
        first = np.float32(params["depth_noise_first_order"])
        if params["if_consider_depth_noise"]:
            sig = zero + first * np.where(valid, depth, np.float32(0))  # pointcloud_tools.h:284-286
        else:
            sig = np.full(n, np.float32(0.1))                          # pointcloud_tools.h:288
        # PINNED: sigma of an invalid pixel is uninitialised in the reference; zero-order term here
        cloud["sigma"] = np.where(valid, sig, zero if params["if_consider_depth_noise"] else np.float32(0.1))
        cloud["track_id"] = np.where(valid, best_track, 0)
        cloud["label_id"] = np.where(valid, best_label, 0)
        cloud["is_valid"] = valid.astype(np.uint8)
        return depth.reshape(H, W), cloud, pos.astype(np.float32), q.astype(np.float32)


# label id -> static instance id of the reference's cfg/object_info.csv (65535 where a label has none)
LABEL_TO_STATIC_INSTANCE = np.full(256, 65535, np.uint16)
for _l, _i in {0: 65535, 3: 65533, 4: 65532, 5: 65531, 6: 65530, 7: 65529, 8: 65528, 9: 65527, 10: 65526, 11: 65525,
               12: 65524, 13: 65523}.items():
    LABEL_TO_STATIC_INSTANCE[_l] = _i


def raw_inputs(cfg, cloud, scene):
    """The inputs the reference's update() receives for the same frame (mask_kpts_msgs semantics,
    docs/custom_files.md:16-45): a "static" MONO8 mask whose pixel value + 1 is the label id, and one MONO8 mask per
    movable object.  Derived from a rendered LabeledPoint image; pixels of movable objects show Road in the static
    mask (something static has to be there)."""
    H, W = cfg["height"], cfg["width"]
    lab = cloud["label_id"].astype(np.int32)
    trk = cloud["track_id"].astype(np.int32)
    movable = (trk <= cfg["max_movable_track"]) & (cloud["is_valid"] > 0)
    static_label = np.where(movable | (cloud["is_valid"] == 0), LABEL_ROAD, lab)
    static_mask = (static_label - 1).astype(np.uint8).reshape(H, W)
    objects = []
    for t in scene.dyn_tracks:
        m = (movable & (trk == int(t))).astype(np.uint8).reshape(H, W)
        objects.append((int(t), LABEL_CAR, m))
    return static_mask, objects


def make_frames(cfg_name, n_frames, params_name=None, seed=7, **scene_kw):
    """Convenience: list of (depth, cloud, cam_pos, cam_q, moves) for a named configuration."""
    cfg = CONFIGS[cfg_name]
    params = PARAMS[params_name or CONFIG_PARAMS[cfg_name]]
    sc = Scene(cfg, seed=seed, **scene_kw)
    frames = []
    for t in range(n_frames):
        depth, cloud, pos, q = sc.render(t, params)
        frames.append((depth, cloud, pos, q, sc.moves(t)))
    return cfg, params, frames


def _render_job(job):
    cfg, params, scene_kw, t = job
    return Scene(cfg, **scene_kw).render(t, params)


def render_frames(cfg, params, scene_kw, frames, workers=None):
    """[Scene(cfg, **scene_kw).render(t, params) for t in frames], rendered by a pool of worker processes (spawned: they
    import numpy and this module only, never the HIP library the parent may hold).  Frames are independent and a cluttered
    scene costs about a second each; with one worker the list is rendered in this process."""
    import multiprocessing as mp
    import os
    frames = list(frames)
    # (SDM_RENDER_WORKERS: fewer of them under a profiler that attaches to every child process)
    workers = min(workers or int(os.environ.get("SDM_RENDER_WORKERS", "0")) or min(64, os.cpu_count() or 1), len(frames))
    jobs = [(cfg, params, scene_kw, t) for t in frames]
    if workers <= 1:
        return [_render_job(j) for j in jobs]
    with mp.get_context("spawn").Pool(workers) as pool:
        return pool.map(_render_job, jobs, chunksize=1)


def render_frames_cached(cfg, params, scene_kw, frames, cache_path=None):
    """render_frames with the result kept in files <cache_path>.{key,depth,cloud,pose}.npy (development: A/B runs of the
    library on the same rendered frames).  The key file carries a digest of everything the frames depend on - configuration,
    parameters, scene, frame numbers - and the cache is ignored (and rewritten) when that does not match.  Plain arrays,
    written frame by frame into memory-mapped files and read back as views of them: no second copy of a few gigabytes of
    frames in RAM (the GPU boxes' memory limit is not generous), and nothing in the files is executed."""
    import hashlib
    import json
    import os
    frames = list(frames)
    if not cache_path:
        return render_frames(cfg, params, scene_kw, frames)
    key = hashlib.sha1(json.dumps([sorted(cfg.items()), sorted(params.items()), sorted(scene_kw.items()), frames],
                                  default=str).encode()).hexdigest()
    names = {k: "%s.%s.npy" % (cache_path, k) for k in ("key", "depth", "cloud", "pose")}
    n, hw = len(frames), cfg["width"] * cfg["height"]
    try:
        if str(np.load(names["key"], allow_pickle=False)) == key:
            depth = np.load(names["depth"], mmap_mode="r", allow_pickle=False)
            cloud = np.load(names["cloud"], mmap_mode="r", allow_pickle=False)
            pose = np.load(names["pose"], allow_pickle=False)
            if depth.shape[0] == n and cloud.shape == (n, hw * LABELED_POINT.itemsize):
                return [(depth[i], cloud[i].view(LABELED_POINT), pose[i, :3].copy(), pose[i, 3:].copy()) for i in range(n)]
    except (OSError, ValueError):
        pass
    out = render_frames(cfg, params, scene_kw, frames)
    depth = np.lib.format.open_memmap(names["depth"], mode="w+", dtype=np.float32, shape=(n, cfg["height"], cfg["width"]))
    cloud = np.lib.format.open_memmap(names["cloud"], mode="w+", dtype=np.uint8, shape=(n, hw * LABELED_POINT.itemsize))
    pose = np.zeros((n, 7), np.float32)
    for i, f in enumerate(out):
        depth[i] = f[0]
        cloud[i] = np.ascontiguousarray(f[1]).view(np.uint8).reshape(-1)
        pose[i, :3], pose[i, 3:] = f[2], f[3]
    depth.flush()
    cloud.flush()
    del depth, cloud
    np.save(names["pose"], pose)
    np.save(names["key"], np.array(key))
    return out


def remove_frame_cache(cache_path):
    import os
    for k in ("key", "depth", "cloud", "pose"):
        try:
            os.remove("%s.%s.npy" % (cache_path, k))
        except OSError:
            pass


STATE_FIELDS = [("px", np.float32), ("py", np.float32), ("pz", np.float32), ("w", np.float32),
                ("ts", np.uint16), ("track", np.uint16), ("label", np.uint8), ("status", np.uint8),
                ("forget", np.uint8), ("owner", np.uint16)]


def prefill_state(cfg, scene, n_particles, seed=11, shard_rank=0, shard_count=1):
    """A map state (SoA dump format of sdm_load_state) holding ~n_particles live particles.

    The street scene only ever shows a few 10^4 surface voxels to the camera, far fewer than the
    "2M particles" BASELINE.json quotes its metric on, so the benchmark map is topped up with particles
    in the space the camera cannot see into: below the ground plane and behind the side walls.  They
    are ordinary UPDATED particles (time stamp 1, the map then starts at global_time_stamp 1): the
    visibility stage has to load, project and occlusion-test those inside the frustum box every frame,
    the occupancy sweep fuses them, and the ring shift expires them as the ego moves — exactly the work a
    long-running map carries.  Ring state must be the initial one (no shift yet).
    With shard_count > 1 only the Z-slab of shard_rank is generated (arrays of V/shard_count * S slots)."""
    rng = np.random.default_rng(seed + 7919 * shard_rank)
    x_n, y_n, z_n, p_n = cfg["x_n"], cfg["y_n"], cfg["z_n"], cfg["p_n"]
    NX, NY, NZ, S = 1 << x_n, 1 << y_n, 1 << z_n, 1 << p_n
    size = np.float32(cfg["voxel_size"])
    NZl = NZ // shard_count
    z0 = NZl * shard_rank
    Vl = NX * NY * NZl
    pmin = [-(N >> 1) * size for N in (NX, NY, NZ)]
    gy = int(np.ceil((scene.ground_y + 0.5 - float(pmin[1])) / float(size)))
    wx_lo = int(np.floor((-scene.wall_x - 0.5 - float(pmin[0])) / float(size)))
    wx_hi = int(np.ceil((scene.wall_x + 0.5 - float(pmin[0])) / float(size)))
    mx = np.arange(NX)
    my = np.arange(NY)
    hidden = np.zeros((NZl, NY, NX), bool)
    hidden[:, my >= gy, :] = True
    hidden[:, :, (mx < wx_lo) | (mx > wx_hi)] = True
    cand = np.flatnonzero(hidden.ravel())
    del hidden
    n_vox = min(len(cand), max(n_particles // (S - 1), 1))
    vox = np.sort(rng.choice(cand, n_vox, replace=False)).astype(np.int64)   # local voxel index
    del cand
    st = {k: np.zeros(Vl * S, dt) for k, dt in STATE_FIELDS}
    st["owner"][:] = 0xFFFF
    st["status"].reshape(Vl, S)[:, 0] = 5                     # TIMEPTC
    vx, vy, vz = vox & (NX - 1), (vox >> x_n) & (NY - 1), (vox >> (x_n + y_n)) + z0
    under = vy >= gy
    for s in range(1, S):
        idx = vox * S + s
        st["px"][idx] = (vx.astype(np.float32) + rng.random(n_vox, np.float32)) * size + pmin[0]
        st["py"][idx] = (vy.astype(np.float32) + rng.random(n_vox, np.float32)) * size + pmin[1]
        st["pz"][idx] = (vz.astype(np.float32) + rng.random(n_vox, np.float32)) * size + pmin[2]
        st["w"][idx] = (0.08 + 0.5 * rng.random(n_vox, np.float32)).astype(np.float32)
        st["ts"][idx] = 1
        st["track"][idx] = np.where(under, TRACK_ROAD, TRACK_BUILDING)
        st["label"][idx] = np.where(under, LABEL_ROAD, LABEL_BUILDING)
        st["status"][idx] = 1                                 # UPDATED
    # every voxel of the map has been observed at frame 1 (a long-running map has seen its whole volume): the
    # occupancy sweep then has to read every slot of every voxel instead of skipping never-seen voxels
    st["ts"][0::S] = 1
    ring = {"global_time_stamp": 1, "moved_steps": [0, 0, 0], "eq_steps": [0, 0, 0],
            "map_center": [0.0, 0.0, 0.0], "last_pos": [0.0, 0.0, 0.0], "birth_cursor": 0, "move_cursor": 0}
    return st, ring, n_vox * (S - 1)


def write_clip(path, cfg_name, n_frames, params_name=None, seed=7, noise=None, **scene_kw):
    """Dump a synthetic clip in the format tools/replay/replay.cpp reads (ROS-free replay): per frame
    the inputs the reference's update() receives - depth, "static" MONO8 mask, one MONO8 mask per movable object, the
    double pose - plus the object motions its object layer would have produced.  Returns what was written, so that a
    test can push the same frames through the Python binding."""
    import struct
    from . import binding
    cfg = CONFIGS[cfg_name]
    params = PARAMS[params_name or CONFIG_PARAMS[cfg_name]]
    noise = noise_table() if noise is None else np.ascontiguousarray(noise, np.float32)
    sc = Scene(cfg, seed=seed, **scene_kw)
    c = binding.Config()
    for k, _ in binding.Config._fields_:
        setattr(c, k, cfg.get(k, 0) if k not in ("shard_count",) else 1)
    c.max_movable_track = cfg["max_movable_track"]
    p = binding.Params()
    for k, _ in binding.Params._fields_:
        setattr(p, k, params[k])
    frames = []
    with open(path, "wb") as f:
        f.write(b"SDMCLIP1")
        f.write(bytes(c))
        f.write(bytes(p))
        f.write(struct.pack("<I", noise.size))
        f.write(noise.tobytes())
        f.write(LABEL_TO_STATIC_INSTANCE.astype("<u2").tobytes())
        f.write(struct.pack("<I", n_frames))
        for t in range(n_frames):
            depth, cloud, pos, q = sc.render(t, params)
            static_mask, objects = raw_inputs(cfg, cloud, sc)
            pos64, q64 = sc.pose(t)
            moves = np.ascontiguousarray(sc.moves(t), dtype=binding.OBJECT_MOVE)
            f.write(np.asarray(pos64, "<f8").tobytes())
            f.write(np.asarray(q64, "<f8").tobytes())
            f.write(struct.pack("<III", 1, len(objects), moves.size))
            f.write(np.ascontiguousarray(depth, "<f4").tobytes())
            f.write(np.ascontiguousarray(static_mask, np.uint8).tobytes())
            for trk, lab, mask in objects:
                f.write(struct.pack("<ii", trk, lab))
                f.write(np.ascontiguousarray(mask, np.uint8).tobytes())
            f.write(moves.tobytes())
            frames.append((depth, static_mask, objects, pos64, q64, moves))
    return cfg, params, noise, frames
