"""CPU restatement of the reference's object layer - TEST INFRASTRUCTURE, never shipped and
never timed: only tests/ may import it.

Follows the reference class by class with numpy standing in for Eigen (np.linalg.svd for JacobiSVD):
  estimate_transformation          include/utils/basic_algorithms.h:54-92
  estimate_transformation_ransac   include/utils/basic_algorithms.h:104-195
  MotionEstimation                 include/object_layer.h:57-199   (translation part; the angular velocity feeds nothing)
  ObjectTransformations            include/object_layer.h:203-297
  ObjectSet                        include/object_layer.h:345-586
  ObjectLayer.update               include/semantic_dsp_map.h:304-566  (objectLevelUpdate)
  ObjectLayer.collect              include/semantic_dsp_map.h:588-736  (object loop of the prediction step)

Parity unpinned: the reference cannot be built here (Eigen/OpenCV/PCL are not in the image) and ships no tests or golden
vectors for this code; its RANSAC draws from an unseeded std::mt19937, so it has no reproducible output to pin against
either.  The sampler below (splitmix64) is this project's and is the one the product (csrc/objects.cpp) uses.
"""
import math

import numpy as np

M64 = (1 << 64) - 1
MODE_KITTI360, MODE_CODA, MODE_VKITTI2, MODE_ZED2 = 0, 1, 2, 3


class Sampler:
    def __init__(self, seed):
        self.s = seed & M64

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        return z ^ (z >> 31)

    def index(self, n):
        return ((self.next() >> 32) * n) >> 32


def mix64(v):
    return Sampler(v).next()


def call_seed(seed, global_time_stamp, track_id):
    return mix64((seed & M64) ^ mix64(((global_time_stamp & 0xFFFFFFFF) << 32) | (track_id & 0xFFFFFFFF)))


def estimate_transformation(P, Q):
    """basic_algorithms.h:54-92.  P, Q: 3 x N."""
    cp = P.mean(axis=1)
    cq = Q.mean(axis=1)
    H = (P - cp[:, None]) @ (Q - cq[:, None]).T
    U, _, Vt = np.linalg.svd(H)
    V = Vt.T
    R = V @ U.T
    if np.linalg.det(R) < 0:
        V = V.copy()
        V[:, 2] *= -1
        R = V @ U.T
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = cq - R @ cp
    return T


def estimate_transformation_ransac(P, Q, max_iterations=100, threshold=0.5, recompute_with_inliers=False, seed=0):
    """basic_algorithms.h:104-195 -> (T, inlier indices, mse over the inliers)."""
    N = P.shape[1]
    max_inliers = -1
    best_T = np.eye(4)
    inlier_indices = []
    rng = Sampler(seed)
    for _ in range(max_iterations):
        indices = []
        while len(indices) < 3:
            r = rng.index(N)
            if r not in indices:
                indices.append(r)
        T = estimate_transformation(P[:, indices], Q[:, indices])
        Pt = T[:3, :3] @ P + T[:3, 3:4]
        err = np.linalg.norm(Pt - Q, axis=0)
        temp = [j for j in range(N) if err[j] < threshold]
        if len(temp) > max_inliers:
            max_inliers = len(temp)
            best_T = T
            inlier_indices = temp
        if max_inliers > 0.9 * N:
            break
    if recompute_with_inliers and len(inlier_indices) >= 3:
        result = estimate_transformation(P[:, inlier_indices], Q[:, inlier_indices])
    else:
        result = best_T
    Pt = result[:3, :3] @ P + result[:3, 3:4]
    sq = ((Pt - Q) ** 2).sum(axis=0)
    total_in = float(sum(sq[j] for j in inlier_indices))
    mse = total_in / len(inlier_indices) if inlier_indices else float("nan")
    return result, inlier_indices, mse


class ObjectTransformations:
    """object_layer.h:203-297 with MotionEstimation (:57-199) folded in."""

    def __init__(self):
        self.t_matrix_vec, self.stamp_vec, self.delta_t_vec, self.reference_vec = [], [], [], []
        self.translation_velocity = np.zeros(3)
        self.updated = False

    def _erase_first(self):
        for v in (self.t_matrix_vec, self.stamp_vec, self.delta_t_vec, self.reference_vec):
            v.pop(0)

    def update(self, T, delta_t, reference_point, gts):
        self.t_matrix_vec.append(T.copy())
        self.delta_t_vec.append(delta_t)
        self.reference_vec.append(np.array(reference_point, dtype=np.float64))
        self.stamp_vec.append(gts)
        while self.reference_vec:
            if ((gts - self.stamp_vec[0]) & 0xFFFFFFFF) > 10:
                self._erase_first()
            else:
                break
        if len(self.t_matrix_vec) > 5:
            self._erase_first()
        if len(self.t_matrix_vec) < 2:
            self.updated = False
            return
        # estimateByTransformations (:92-132) + estimate (:139-172)
        total = np.zeros(3)
        with np.errstate(divide="ignore", invalid="ignore"):
            for T_i, dt, ref in zip(self.t_matrix_vec, self.delta_t_vec, self.reference_vec):
                pts = [ref, ref + np.array([1.0, 0, 0]), ref + np.array([0, 1.0, 0])]
                moved = [(T_i @ np.append(p, 1.0))[:3] for p in pts]
                prev = (pts[0] + pts[1] + pts[2]) / 3.0
                curr = (moved[0] + moved[1] + moved[2]) / 3.0
                total = total + (curr - prev) / dt
            self.translation_velocity = total / (len(self.t_matrix_vec) - 1)
        self.updated = True

    def predict(self, delta_t):
        if not self.updated:
            return None
        T = np.eye(4)
        T[:3, 3] = self.translation_velocity * delta_t
        return T


class Tracked:
    def __init__(self, label, gts):
        self.label = label
        self.observation_time_step = gts
        self.observation_count = 1
        self.to_match_with_templates = True
        self.to_match_with_previous = False
        self.t_matrix_vec = []
        self.moved_vec = []
        self.moved_probability = 0.5
        self.transformations = ObjectTransformations()


def point_out_of_fov(cfg, cam_pos, q, p, margin):
    """semantic_dsp_map.h:1421-1442; q = (w, x, y, z); Eigen's q.inverse() * v."""
    n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]
    w = q[0] / n2
    u = -np.array(q[1:4], dtype=np.float64) / n2
    v = np.asarray(p, dtype=np.float64) - np.asarray(cam_pos, dtype=np.float64)
    uv = 2.0 * np.cross(u, v)
    pc = v + w * uv + np.cross(u, uv)
    if pc[2] <= 0:
        return True
    px = cfg["fx"] * (pc[0] / pc[2]) + cfg["cx"]
    py = cfg["fy"] * (pc[1] / pc[2]) + cfg["cy"]
    return bool(px < margin or px >= cfg["image_width"] - margin or py < margin or py >= cfg["image_height"] - margin)


class ObjectLayer:
    """cfg keys = fields of sdm_objects_config.  Observations: dicts with track_id, label_id, is_static, kpts_current
    (n x 3), kpts_previous (n x 3 or None)."""

    def __init__(self, cfg):
        self.cfg = dict(cfg)
        self.tracked = {}
        self.last_kpts, self.last_stamp = {}, {}
        self.key_kpts, self.key_stamp = {}, {}
        self.time_stamp_last = 0.0

    def clear(self):
        self.tracked.clear()
        self.last_kpts.clear()
        self.last_stamp.clear()
        self.key_kpts.clear()
        self.key_stamp.clear()

    # ObjectSet::updateObject (object_layer.h:467-540)
    def _update_object(self, oid, T, reference_point, label, time_interval, moved_observation, gts):
        c = self.cfg
        o = self.tracked[oid]
        transition = (T @ np.append(reference_point, 1.0))[:3] - reference_point
        if c["mode"] == MODE_KITTI360:
            moving = False
        elif c["mode"] == MODE_CODA:
            moving = True
        else:
            if moved_observation == -1:
                moved = np.linalg.norm(transition) > c["movement_distance_threshold"]
            else:
                moved = moved_observation == 1
            if moved:
                o.moved_probability += c["movement_increment"]
            else:
                o.moved_probability -= c["movement_decrement"]
            moving = o.moved_probability > c["movement_probability_threshold"]
        o.moved_probability = min(1.0, max(0.0, o.moved_probability))
        o.label = label
        o.t_matrix_vec = [T.copy()]
        o.observation_time_step = gts
        o.observation_count += 1
        o.to_match_with_previous = False
        o.moved_vec = [bool(moving)]
        if moving:
            o.transformations.update(T, time_interval, reference_point, gts)

    @staticmethod
    def _predict_and_set(o, time_interval):
        T = o.transformations.predict(time_interval)
        if T is not None:
            o.t_matrix_vec = [T]
        o.to_match_with_previous = False

    def update(self, observations, cam_pos, cam_q, ts, gts):
        c = self.cfg
        matched = c["mode"] in (MODE_CODA, MODE_VKITTI2)
        observed = set()
        for ob in observations:
            if ob["track_id"] > c["max_movable_instance_id"] or ob["is_static"]:
                continue
            oid = ob["track_id"]
            observed.add(oid)
            if ob["label_id"] < 0:
                continue
            min_kpts = 5 if matched else 4
            cur = np.asarray(ob["kpts_current"], dtype=np.float64).reshape(-1, 3)
            success = False
            if oid not in self.tracked:
                closest = float("inf")
                for p in cur:
                    closest = min(closest, max(abs(p[0] - cam_pos[0]), abs(p[1] - cam_pos[1]), abs(p[2] - cam_pos[2])))
                if len(cur) == 0:
                    closest = np.finfo(np.float64).max
                if closest > c["map_half_size_scaled"]:
                    continue
                self.tracked[oid] = Tracked(ob["label_id"], gts)
                success = True
                if c["mode"] == MODE_ZED2:
                    self.last_kpts[oid], self.last_stamp[oid] = cur.copy(), ts
                    self.key_kpts[oid], self.key_stamp[oid] = cur.copy(), ts
            elif len(cur) >= min_kpts:
                T = np.eye(4)
                time_interval, moved_observation = 0.15, -1
                if matched:
                    prev = np.asarray(ob["kpts_previous"], dtype=np.float64).reshape(-1, 3)
                    T, inl, mse = estimate_transformation_ransac(prev.T, cur.T, 100, 0.5, True, call_seed(c["seed"], gts, oid))
                    f02, f05 = float(np.float32(0.2)), float(np.float32(0.5))
                    success = not (mse > f02 or len(inl) < 5 or len(inl) / float(len(cur)) < f05)
                    reference_point = prev[inl[0]] if inl else prev[0]
                else:
                    out_of_fov = False
                    for p in cur:
                        out_of_fov = point_out_of_fov(c, cam_pos, cam_q, p, 5)
                    time_diff = ts - self.last_stamp.setdefault(oid, 0.0)
                    moved_observation = 0
                    if out_of_fov:
                        success = False
                    elif oid not in self.last_kpts or len(self.last_kpts[oid]) < 4:  # fewer than 4 stored: UB in the reference
                        self.last_kpts[oid], self.last_stamp[oid] = cur.copy(), ts
                        self.key_kpts[oid], self.key_stamp[oid] = cur.copy(), ts
                        success = False
                    else:
                        last4, cur4 = self.last_kpts[oid][:4], cur[:4]
                        T, _, _ = estimate_transformation_ransac(last4.T, cur4.T, 2, 0.5, False, call_seed(c["seed"], gts, oid))
                        thr = max(c["movement_distance_threshold"], float(np.linalg.norm(cur4[1] - cur4[0])))
                        key0 = self.key_kpts[oid][0] if oid in self.key_kpts else np.zeros(3)
                        if np.linalg.norm(cur4[0] - key0) > thr:
                            moved_observation = 1
                        if ts - self.key_stamp.get(oid, 0.0) > 2.0:
                            self.key_kpts[oid], self.key_stamp[oid] = cur.copy(), ts
                        reference_point = last4[0].copy()
                        self.last_kpts[oid], self.last_stamp[oid] = cur.copy(), ts
                        time_interval = time_diff
                        success = True
                if success:
                    self._update_object(oid, T, reference_point, ob["label_id"], time_interval, moved_observation, gts)
            if matched and not success:
                o = self.tracked.get(oid)
                if o is not None and o.moved_vec and o.moved_vec[0]:
                    if o.transformations.updated:
                        self._predict_and_set(o, 0.2)
                    else:
                        o.observation_time_step = gts
                        o.to_match_with_previous = True
                        o.to_match_with_templates = False
        for oid, o in self.tracked.items():
            if oid in observed:
                continue
            if not o.moved_vec or not o.moved_vec[0]:
                continue
            dt = ts - self.time_stamp_last
            if math.fabs(dt) > 1.0:
                dt = 1.0
            self._predict_and_set(o, dt)
        self.time_stamp_last = ts

    def collect(self, gts, max_lost, present_tracks=()):
        moves, lost = [], []
        for oid in sorted(self.tracked):
            o = self.tracked[oid]
            if not o.moved_vec or not o.moved_vec[0]:
                continue
            if ((gts - o.observation_time_step) & 0xFFFFFFFF) >= (max_lost & 0xFFFFFFFF):
                lost.append(oid)
            elif o.t_matrix_vec:
                moves.append((oid, o.t_matrix_vec[0].astype(np.float32)))
        floating = sorted(set(t for t in present_tracks if t not in self.tracked))
        for oid in lost:
            del self.tracked[oid]
            self.last_kpts.pop(oid, None)
            self.last_stamp.pop(oid, None)
        for oid in floating:
            self.last_kpts.pop(oid, None)
            self.last_stamp.pop(oid, None)
            self.key_kpts.pop(oid, None)
            self.key_stamp.pop(oid, None)
        return moves, sorted(lost + floating)
