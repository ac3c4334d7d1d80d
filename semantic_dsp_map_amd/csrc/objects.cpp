// Object layer: which tracked objects move by which rigid transform this frame, which are
// wiped.  Host code, O(objects) per frame; produces the `moves` / `remove_tracks` arguments of sdm_update.  Plain
// doubles, no Eigen: the reference's Eigen calls are restated where their result is defined (SVD-based rigid fit,
// quaternion rotation) and left out where it is not used (angular velocity).  See include/sdm_objects.h for the list of
// reference functions and the deliberate differences.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <limits>
#include <map>
#include <new>
#include <set>
#include <vector>

#include "../../include/sdm_objects.h"

namespace {

struct V3 {
  double x = 0, y = 0, z = 0;
};
inline V3 operator+(const V3 &a, const V3 &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(const V3 &a, const V3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(const V3 &a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(const V3 &a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline double dot(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(const V3 &a, const V3 &b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double norm(const V3 &a) { return std::sqrt(dot(a, a)); }

// 4x4 row major; only rigid transforms are ever stored
struct M4 {
  double m[16];
  static M4 identity() {
    M4 t;
    for (int i = 0; i < 16; ++i) t.m[i] = (i % 5 == 0) ? 1.0 : 0.0;
    return t;
  }
  V3 apply(const V3 &p) const {
    return {m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
            m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]};
  }
};

// ---- sampler ------------------------------------------------------------------------------------------------------
// splitmix64; index = high 32 bits scaled to [0, n)  (the stream is part of the interface: sdm_objects.h, seed)
struct Sampler {
  uint64_t s;
  explicit Sampler(uint64_t seed) : s(seed) {}
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  int index(int n) { return (int)(((next() >> 32) * (uint64_t)n) >> 32); }
};
inline uint64_t mix64(uint64_t v) { return Sampler(v).next(); }
inline uint64_t call_seed(uint64_t seed, uint32_t global_time_stamp, int32_t track_id) {
  return mix64(seed ^ mix64(((uint64_t)global_time_stamp << 32) | (uint32_t)track_id));
}

// ---- 3x3 SVD (one-sided Jacobi), H = U diag(s) V^T, s descending -------------------------------------------------------
void svd3(const double H[9], double U[9], double s[3], double V[9]) {
  double A[9];  // columns are rotated until mutually orthogonal: A = H V
  std::memcpy(A, H, sizeof(A));
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  auto col_dot = [&](const double *M, int p, int q) { return M[p] * M[q] + M[3 + p] * M[3 + q] + M[6 + p] * M[6 + q]; };
  for (int sweep = 0; sweep < 60; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        const double alpha = col_dot(A, p, p), beta = col_dot(A, q, q), gamma = col_dot(A, p, q);
        if (gamma == 0.0 || std::fabs(gamma) <= 1e-300 + 1e-17 * std::sqrt(alpha * beta)) continue;
        rotated = true;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
        for (int r = 0; r < 3; ++r) {
          const double ap = A[3 * r + p], aq = A[3 * r + q];
          A[3 * r + p] = c * ap - sn * aq;
          A[3 * r + q] = sn * ap + c * aq;
          const double vp = V[3 * r + p], vq = V[3 * r + q];
          V[3 * r + p] = c * vp - sn * vq;
          V[3 * r + q] = sn * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  int order[3] = {0, 1, 2};
  double len[3];
  for (int k = 0; k < 3; ++k) len[k] = std::sqrt(col_dot(A, k, k));
  std::sort(order, order + 3, [&](int a, int b) { return len[a] > len[b]; });
  double Vs[9];
  V3 u[3];
  for (int k = 0; k < 3; ++k) {
    const int c = order[k];
    s[k] = len[c];
    for (int r = 0; r < 3; ++r) Vs[3 * r + k] = V[3 * r + c];
    u[k] = {A[c], A[3 + c], A[6 + c]};
  }
  std::memcpy(V, Vs, sizeof(Vs));
  // left vectors: normalised columns where the singular value is not negligible, an orthonormal completion otherwise
  // (the rotation V U^T does not depend on the completion once the sign fix of fit_rigid is applied, as long as the
  // rank is >= 2 - three non-collinear points)
  const double tiny = s[0] * 1e-13;
  int rank = 0;
  for (int k = 0; k < 3; ++k)
    if (s[k] > tiny && s[k] > 0) {
      u[k] = u[k] / s[k];
      rank = k + 1;
    } else {
      break;
    }
  if (rank == 0) {
    u[0] = {1, 0, 0};
    u[1] = {0, 1, 0};
    u[2] = {0, 0, 1};
  } else if (rank == 1) {
    V3 e = std::fabs(u[0].x) < 0.9 ? V3{1, 0, 0} : V3{0, 1, 0};
    u[1] = cross(u[0], e);
    u[1] = u[1] / norm(u[1]);
    u[2] = cross(u[0], u[1]);
  } else if (rank == 2) {
    u[2] = cross(u[0], u[1]);
    u[2] = u[2] / norm(u[2]);
  }
  for (int k = 0; k < 3; ++k) {
    U[k] = u[k].x;
    U[3 + k] = u[k].y;
    U[6 + k] = u[k].z;
  }
}

inline double det3(const double R[9]) {
  return R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
}

// estimateTransformation (basic_algorithms.h:54-92): centroids, H = Pc Qc^T, R = V U^T with the reflection fix on the
// column of the smallest singular value, t = cQ - R cP
M4 fit_rigid(const std::vector<V3> &P, const std::vector<V3> &Q) {
  const size_t n = P.size();
  V3 cp, cq;
  for (size_t i = 0; i < n; ++i) {
    cp = cp + P[i];
    cq = cq + Q[i];
  }
  cp = cp / (double)n;
  cq = cq / (double)n;
  double H[9] = {0};
  for (size_t i = 0; i < n; ++i) {
    const V3 a = P[i] - cp, b = Q[i] - cq;
    const double av[3] = {a.x, a.y, a.z}, bv[3] = {b.x, b.y, b.z};
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) H[3 * r + c] += av[r] * bv[c];
  }
  double U[9], s[3], V[9], R[9];
  svd3(H, U, s, V);
  auto vut = [&]() {
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) R[3 * r + c] = V[3 * r] * U[3 * c] + V[3 * r + 1] * U[3 * c + 1] + V[3 * r + 2] * U[3 * c + 2];
  };
  vut();
  if (det3(R) < 0) {
    for (int r = 0; r < 3; ++r) V[3 * r + 2] = -V[3 * r + 2];
    vut();
  }
  M4 T = M4::identity();
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) T.m[4 * r + c] = R[3 * r + c];
  T.m[3] = cq.x - (R[0] * cp.x + R[1] * cp.y + R[2] * cp.z);
  T.m[7] = cq.y - (R[3] * cp.x + R[4] * cp.y + R[5] * cp.z);
  T.m[11] = cq.z - (R[6] * cp.x + R[7] * cp.y + R[8] * cp.z);
  return T;
}

// estimateTransformationRANSAC (basic_algorithms.h:104-195).  Returns the mean squared error over the inliers
// (NaN for none, as 0/0 in the reference).
double fit_rigid_ransac(const std::vector<V3> &P, const std::vector<V3> &Q, M4 &result, std::vector<int> &inliers,
                        int max_iterations, double threshold, bool recompute_with_inliers, uint64_t seed) {
  const int n = (int)P.size();
  int max_inliers = -1;
  M4 best = M4::identity();
  Sampler rng(seed);
  inliers.clear();
  for (int it = 0; it < max_iterations; ++it) {
    int idx[3], have = 0;
    while (have < 3) {  // three distinct indices, drawn with rejection (:117-124)
      const int r = rng.index(n);
      bool dup = false;
      for (int k = 0; k < have; ++k) dup = dup || idx[k] == r;
      if (!dup) idx[have++] = r;
    }
    std::vector<V3> ps(3), qs(3);
    for (int j = 0; j < 3; ++j) {
      ps[j] = P[idx[j]];
      qs[j] = Q[idx[j]];
    }
    const M4 T = fit_rigid(ps, qs);
    std::vector<int> in;
    for (int j = 0; j < n; ++j)
      if (norm(T.apply(P[j]) - Q[j]) < threshold) in.push_back(j);
    if ((int)in.size() > max_inliers) {
      max_inliers = (int)in.size();
      best = T;
      inliers = in;
    }
    if (max_inliers > 0.9 * n) break;
  }
  if (recompute_with_inliers && inliers.size() >= 3) {
    std::vector<V3> pi(inliers.size()), qi(inliers.size());
    for (size_t j = 0; j < inliers.size(); ++j) {
      pi[j] = P[inliers[j]];
      qi[j] = Q[inliers[j]];
    }
    result = fit_rigid(pi, qi);
  } else {
    result = best;
  }
  double err_in = 0;
  for (int j : inliers) {
    const V3 e = result.apply(P[j]) - Q[j];
    err_in += dot(e, e);
  }
  return err_in / (double)inliers.size();
}

// ---- ObjectTransformations + MotionEstimation (object_layer.h:57-297), translation part ----------------------------------
struct Transformations {
  std::vector<M4> t_matrix;
  std::vector<uint32_t> stamp;
  std::vector<double> delta_t;
  std::vector<V3> reference;
  V3 translation_velocity;
  bool updated = false;

  void erase_first() {
    t_matrix.erase(t_matrix.begin());
    stamp.erase(stamp.begin());
    delta_t.erase(delta_t.begin());
    reference.erase(reference.begin());
  }
  // ObjectTransformations::update (:217-256)
  void update(const M4 &T, double dt, const V3 &ref, uint32_t gts) {
    t_matrix.push_back(T);
    delta_t.push_back(dt);
    reference.push_back(ref);
    stamp.push_back(gts);
    while (!reference.empty()) {
      if (gts - stamp[0] > 10) erase_first();  // unsigned difference, as in the reference
      else break;
    }
    if (t_matrix.size() > 5) erase_first();  // max_window_size
    if (t_matrix.size() < 2) {
      updated = false;
      return;
    }
    // estimateByTransformations + estimate (:92-172): per stored transform the centroid of (ref, ref + ex, ref + ey)
    // before and after; velocities summed and divided by n - 1 (PINNED: not n)
    V3 sum;
    for (size_t i = 0; i < t_matrix.size(); ++i) {
      const V3 p0 = reference[i], p1 = p0 + V3{1, 0, 0}, p2 = p0 + V3{0, 1, 0};
      const V3 prev = (p0 + p1 + p2) / 3.0;
      const V3 curr = (t_matrix[i].apply(p0) + t_matrix[i].apply(p1) + t_matrix[i].apply(p2)) / 3.0;
      sum = sum + (curr - prev) / delta_t[i];
    }
    translation_velocity = sum / (double)(t_matrix.size() - 1);
    updated = true;
  }
  // predictTMatrix -> predictTransformationMatrix (:187-198, 262-270): identity rotation, v * dt
  bool predict(double dt, M4 &T) const {
    if (!updated) return false;
    T = M4::identity();
    T.m[3] = translation_velocity.x * dt;
    T.m[7] = translation_velocity.y * dt;
    T.m[11] = translation_velocity.z * dt;
    return true;
  }
};

// ObjectSet::ObjectInTracking + MJObject (object_layer.h:302-366), the fields that are read somewhere
struct Tracked {
  int label = 0;
  uint32_t observation_time_step = 0;
  int observation_count = 0;
  bool to_match_with_templates = true, to_match_with_previous = false;
  std::vector<M4> t_matrix_vec;   // rigidbody_tmatrix_vec: empty or one entry
  std::vector<bool> moved_vec;    // rigidbody_moved_vec: empty or one entry
  double moved_probability = 0.5;
  Transformations transformations;
};

struct Keypoints {
  std::vector<V3> pts;
  double stamp = 0;
  bool present = false;
};

}  // namespace

struct sdm_objects {
  sdm_objects_config cfg;
  std::map<int, Tracked> tracked;  // object_tracking_hash_map
  // SETTING == 3 state of SemanticDSPMap: last / key 3-D keypoints per object with their time stamps
  std::map<int, Keypoints> last_kpts, key_kpts;
  double time_stamp_last = 0.0;  // `static double time_stamp_double_last` (semantic_dsp_map.h:308)
};

namespace {

// isPointOutOfFOV (semantic_dsp_map.h:1421-1442); q = (w, x, y, z), rotated by q.inverse() the way Eigen does it
// (conjugate / squared norm, then v + w * 2(u x v) + u x 2(u x v))
bool point_out_of_fov(const sdm_objects_config &c, const double cam_pos[3], const double q[4], const V3 &p, int margin) {
  const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  const double w = q[0] / n2;
  const V3 u{-q[1] / n2, -q[2] / n2, -q[3] / n2};
  const V3 v = p - V3{cam_pos[0], cam_pos[1], cam_pos[2]};
  const V3 uv = cross(u, v) * 2.0;
  const V3 pc = v + uv * w + cross(u, uv);
  if (pc.z <= 0) return true;
  const double px = c.fx * (pc.x / pc.z) + c.cx;
  const double py = c.fy * (pc.y / pc.z) + c.cy;
  return px < margin || px >= c.image_width - margin || py < margin || py >= c.image_height - margin;
}

// ObjectSet::updateObject (object_layer.h:467-540)
void update_object(sdm_objects *h, int id, const M4 &T, const V3 &reference_point, int label, double time_interval,
                   int moved_observation, uint32_t gts) {
  Tracked &o = h->tracked[id];
  const sdm_objects_config &c = h->cfg;
  const V3 transition = T.apply(reference_point) - reference_point;
  bool moving;
  if (c.mode == SDM_OBJECTS_MODE_KITTI360) {
    moving = false;
  } else if (c.mode == SDM_OBJECTS_MODE_CODA) {
    moving = true;
  } else {
    bool moved = moved_observation == -1 ? norm(transition) > c.movement_distance_threshold : moved_observation == 1;
    if (moved) o.moved_probability += c.movement_increment;
    else o.moved_probability -= c.movement_decrement;
    moving = o.moved_probability > c.movement_probability_threshold;  // PINNED: decided before the clamp
  }
  o.moved_probability = std::min(1.0, std::max(0.0, o.moved_probability));
  o.label = label;
  o.t_matrix_vec.assign(1, T);
  o.observation_time_step = gts;
  o.observation_count++;
  o.to_match_with_previous = false;
  o.moved_vec.assign(1, moving);
  if (moving) o.transformations.update(T, time_interval, reference_point, gts);
}

// ObjectSet::predictAndSetTransformation (object_layer.h:558-586)
void predict_and_set(Tracked &o, double time_interval) {
  M4 T;
  if (o.transformations.predict(time_interval, T)) o.t_matrix_vec.assign(1, T);
  o.to_match_with_previous = false;
}

std::vector<V3> points_of(const double *p, int n) {
  std::vector<V3> v((size_t)n);
  for (int i = 0; i < n; ++i) v[i] = {p[3 * i], p[3 * i + 1], p[3 * i + 2]};
  return v;
}

sdm_status update_impl(sdm_objects *h, const sdm_object_observation *obs, int32_t n_obs, const double cam_pos[3],
                       const double cam_q[4], double ts, uint32_t gts) {
  const sdm_objects_config &c = h->cfg;
  const bool matched_kpts = c.mode == SDM_OBJECTS_MODE_CODA || c.mode == SDM_OBJECTS_MODE_VKITTI2;
  std::set<int> observed;
  for (int i = 0; i < n_obs; ++i) {
    const sdm_object_observation &ob = obs[i];
    if (ob.track_id > c.max_movable_instance_id || ob.is_static) continue;  // :317
    const int id = ob.track_id;
    observed.insert(id);
    if (ob.label_id < 0) continue;  // label not in g_label_id_map_default (:331-334)
    const int min_kpts = matched_kpts ? 5 : 4;  // :337-341
    const std::vector<V3> cur = points_of(ob.kpts_current, ob.n_kpts);
    bool success = false;
    auto found = h->tracked.find(id);
    if (found == h->tracked.end()) {
      // Case 1 (:345-377): new object; ignored when every keypoint is further than 1.2 x half the map (Chebyshev)
      double closest = std::numeric_limits<double>::max();
      for (const V3 &p : cur) {
        const double dist = std::max(std::max(std::fabs(p.x - cam_pos[0]), std::fabs(p.y - cam_pos[1])), std::fabs(p.z - cam_pos[2]));
        if (dist < closest) closest = dist;
      }
      if (closest > c.map_half_size_scaled) continue;
      Tracked t;  // addNewObject (object_layer.h:388-410)
      t.label = ob.label_id;
      t.observation_time_step = gts;
      t.observation_count = 1;
      h->tracked[id] = t;
      success = true;
      if (c.mode == SDM_OBJECTS_MODE_ZED2) {  // :369-375
        h->last_kpts[id] = {cur, ts, true};
        h->key_kpts[id] = {cur, ts, true};
      }
    } else if (ob.n_kpts >= min_kpts) {
      // Case 2 (:379-513)
      M4 T = M4::identity();
      V3 reference_point;
      double time_interval = 0.15;  // default argument of updateObject
      int moved_observation = -1;
      if (matched_kpts) {  // :383-407
        if (!ob.kpts_previous) {
          // a tracker entry without usable previous key points (the adapter passes NULL when the two lists differ in
          // length; the reference would read past the shorter one): this object's key-point method fails, the frame's
          // other objects are processed as usual - one malformed entry must not drop the whole update
          success = false;
          reference_point = cur[0];
        } else {
          const std::vector<V3> prev = points_of(ob.kpts_previous, ob.n_kpts);
          std::vector<int> inl;
          const double mse = fit_rigid_ransac(prev, cur, T, inl, 100, 0.5, true, call_seed(c.seed, gts, id));
          // `mse > 0.2f`, `ratio < 0.5f`: float literals widened to double; NaN (no inliers) fails no comparison
          success = !(mse > (double)0.2f || inl.size() < 5 || (double)inl.size() / (double)ob.n_kpts < (double)0.5f);
          reference_point = inl.empty() ? prev[0] : prev[inl[0]];  // :492-500
        }
      } else {  // :408-483, box keypoints; the reference's matrices have exactly 4 columns
        bool out_of_fov = false;
        for (const V3 &p : cur) out_of_fov = point_out_of_fov(c, cam_pos, cam_q, p, 5);  // PINNED: the last one decides (:420-422)
        Keypoints &last = h->last_kpts[id];  // operator[] of the reference: a missing stamp reads 0 (:424)
        const double time_diff = ts - last.stamp;
        moved_observation = 0;
        if (out_of_fov) {
          success = false;
        } else if (!last.present || last.pts.size() < 4) {  // :431-438; fewer than four stored keypoints (the reference
          // would read past them): treated like none
          last = {cur, ts, true};
          h->key_kpts[id] = {cur, ts, true};
          success = false;
        } else {
          std::vector<V3> p4(last.pts.begin(), last.pts.begin() + 4);
          std::vector<V3> q4(cur.begin(), cur.begin() + 4);
          std::vector<int> inl;
          fit_rigid_ransac(p4, q4, T, inl, 2, 0.5, false, call_seed(c.seed, gts, id));
          double thr = c.movement_distance_threshold;  // :451-457
          const double width = norm(q4[1] - q4[0]);
          if (thr < width) thr = width;
          Keypoints &key = h->key_kpts[id];
          const V3 key0 = key.pts.empty() ? V3{} : key.pts[0];
          if (norm(q4[0] - key0) > thr) moved_observation = 1;
          if (ts - key.stamp > 2.0) key = {cur, ts, true};  // :469-472
          reference_point = p4[0];                            // :506
          last = {cur, ts, true};                             // :475-476
          time_interval = time_diff;
          success = true;
        }
      }
      if (success) update_object(h, id, T, reference_point, ob.label_id, time_interval, moved_observation, gts);
    }
    if (matched_kpts && !success) {  // Case 3 (:517-539): tracked, moving, no usable keypoints this frame
      auto it = h->tracked.find(id);
      if (it != h->tracked.end() && !it->second.moved_vec.empty() && it->second.moved_vec[0]) {
        if (it->second.transformations.updated) {
          predict_and_set(it->second, 0.2);  // default argument of predictAndSetTransformation
        } else {  // setFlagsUpdateByMatching (object_layer.h:544-553)
          it->second.observation_time_step = gts;
          it->second.to_match_with_previous = true;
          it->second.to_match_with_templates = false;
        }
      }
    }
  }
  // tracked, moving, not observed this frame: constant-velocity prediction (:543-561)
  for (auto &kv : h->tracked) {
    if (observed.count(kv.first)) continue;
    if (kv.second.moved_vec.empty() || !kv.second.moved_vec[0]) continue;
    double dt = ts - h->time_stamp_last;
    if (std::fabs(dt) > 1.0) dt = 1.0;  // PINNED: `abs(time_diff)` read as the floating-point overload
    predict_and_set(kv.second, dt);
  }
  h->time_stamp_last = ts;
  return SDM_OK;
}

void forget(sdm_objects *h, int id, bool floating) {
  h->tracked.erase(id);
  h->last_kpts.erase(id);                // :706-709
  if (floating) h->key_kpts.erase(id);   // only the floating-object path clears the key keypoints too (:727-732)
}

}  // namespace

extern "C" {

sdm_status sdm_objects_create(const sdm_objects_config *cfg, sdm_objects **out) {
  if (!cfg || !out || cfg->mode < 0 || cfg->mode > 3) return SDM_ERR_INVALID_ARGUMENT;
  sdm_objects *h = new (std::nothrow) sdm_objects();
  if (!h) return SDM_ERR_CAPACITY;
  h->cfg = *cfg;
  *out = h;
  return SDM_OK;
}

void sdm_objects_destroy(sdm_objects *h) { delete h; }

sdm_status sdm_objects_clear(sdm_objects *h) {
  if (!h) return SDM_ERR_INVALID_ARGUMENT;
  h->tracked.clear();
  h->last_kpts.clear();
  h->key_kpts.clear();
  return SDM_OK;
}

sdm_status sdm_objects_set_bayes(sdm_objects *h, double distance_threshold, double probability_threshold, double increment,
                                 double decrement) {
  if (!h) return SDM_ERR_INVALID_ARGUMENT;
  h->cfg.movement_distance_threshold = distance_threshold;
  h->cfg.movement_probability_threshold = probability_threshold;
  h->cfg.movement_increment = increment;
  h->cfg.movement_decrement = decrement;
  return SDM_OK;
}

sdm_status sdm_objects_update(sdm_objects *h, const sdm_object_observation *obs, int32_t n_obs, const double cam_pos[3],
                              const double cam_q[4], double time_stamp, uint32_t global_time_stamp) {
  if (!h || n_obs < 0 || (n_obs && !obs) || !cam_pos || !cam_q) return SDM_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < n_obs; ++i)
    if (obs[i].n_kpts < 0 || (obs[i].n_kpts && !obs[i].kpts_current)) return SDM_ERR_INVALID_ARGUMENT;
  try {
    return update_impl(h, obs, n_obs, cam_pos, cam_q, time_stamp, global_time_stamp);
  } catch (...) {
    return SDM_ERR_CAPACITY;
  }
}

sdm_status sdm_objects_collect(sdm_objects *h, uint32_t global_time_stamp, int32_t max_obersevation_lost_time,
                               const int32_t *present_tracks, int32_t n_present, sdm_object_move *moves, int32_t moves_cap,
                               int32_t *n_moves, int32_t *remove_tracks, int32_t remove_cap, int32_t *n_remove) {
  if (!h || !n_moves || !n_remove || moves_cap < 0 || remove_cap < 0 || (moves_cap && !moves) || (remove_cap && !remove_tracks) ||
      n_present < 0 || (n_present && !present_tracks))
    return SDM_ERR_INVALID_ARGUMENT;
  try {
    std::vector<int> lost;
    std::vector<sdm_object_move> mv;
    for (auto &kv : h->tracked) {  // semantic_dsp_map.h:593-693
      const Tracked &o = kv.second;
      if (o.moved_vec.empty() || !o.moved_vec[0]) continue;
      if (global_time_stamp - o.observation_time_step >= (uint32_t)max_obersevation_lost_time) {
        lost.push_back(kv.first);
      } else if (!o.t_matrix_vec.empty()) {
        sdm_object_move m;
        m.track_id = kv.first;
        for (int k = 0; k < 16; ++k) m.T[k] = (float)o.t_matrix_vec[0].m[k];  // Matrix4d -> Matrix4f (:675-676)
        mv.push_back(m);
      }
    }
    std::vector<int> floating;  // :713-733
    for (int k = 0; k < n_present; ++k)
      if (!h->tracked.count(present_tracks[k])) floating.push_back(present_tracks[k]);
    std::sort(floating.begin(), floating.end());
    floating.erase(std::unique(floating.begin(), floating.end()), floating.end());
    if ((int)mv.size() > moves_cap || (int)(lost.size() + floating.size()) > remove_cap) return SDM_ERR_CAPACITY;
    for (size_t k = 0; k < mv.size(); ++k) moves[k] = mv[k];
    *n_moves = (int32_t)mv.size();
    std::vector<int> all(lost);
    all.insert(all.end(), floating.begin(), floating.end());
    std::sort(all.begin(), all.end());
    for (size_t k = 0; k < all.size(); ++k) remove_tracks[k] = all[k];
    *n_remove = (int32_t)all.size();
    for (int id : lost) forget(h, id, false);
    for (int id : floating) forget(h, id, true);
    return SDM_OK;
  } catch (...) {
    return SDM_ERR_CAPACITY;
  }
}

sdm_status sdm_objects_query(sdm_objects *h, int32_t track_id, sdm_object_info *out) {
  if (!h || !out) return SDM_ERR_INVALID_ARGUMENT;
  std::memset(out, 0, sizeof(*out));
  auto it = h->tracked.find(track_id);
  if (it == h->tracked.end()) return SDM_OK;
  const Tracked &o = it->second;
  out->exists = 1;
  out->label_id = o.label;
  out->observation_time_step = (int32_t)o.observation_time_step;
  out->observation_count = o.observation_count;
  out->has_moved_flag = !o.moved_vec.empty();
  out->moving = !o.moved_vec.empty() && o.moved_vec[0];
  out->to_match_with_previous = o.to_match_with_previous;
  out->prediction_available = o.transformations.updated;
  out->n_transformations = (int32_t)o.transformations.t_matrix.size();
  out->has_t_matrix = !o.t_matrix_vec.empty();
  out->moved_probability = o.moved_probability;
  out->translation_velocity[0] = o.transformations.translation_velocity.x;
  out->translation_velocity[1] = o.transformations.translation_velocity.y;
  out->translation_velocity[2] = o.transformations.translation_velocity.z;
  if (!o.t_matrix_vec.empty()) std::memcpy(out->t_matrix, o.t_matrix_vec[0].m, sizeof(out->t_matrix));
  return SDM_OK;
}

sdm_status sdm_objects_count(sdm_objects *h, int32_t *n_tracked) {
  if (!h || !n_tracked) return SDM_ERR_INVALID_ARGUMENT;
  *n_tracked = (int32_t)h->tracked.size();
  return SDM_OK;
}

sdm_status sdm_objects_fit_rigid(const double *P, const double *Q, int32_t n, double T[16]) {
  if (!P || !Q || !T || n < 1) return SDM_ERR_INVALID_ARGUMENT;
  try {
    const M4 r = fit_rigid(points_of(P, n), points_of(Q, n));
    std::memcpy(T, r.m, sizeof(r.m));
    return SDM_OK;
  } catch (...) {
    return SDM_ERR_CAPACITY;
  }
}

sdm_status sdm_objects_fit_rigid_ransac(const double *P, const double *Q, int32_t n, int32_t max_iterations, double threshold,
                                        int32_t recompute_with_inliers, uint64_t seed, double T[16], int32_t *inliers,
                                        int32_t *n_inliers, double *mse_inliers) {
  // fewer than three points would loop forever in the reference's sampler (:117-124)
  if (!P || !Q || !T || n < 3 || max_iterations < 1) return SDM_ERR_INVALID_ARGUMENT;
  try {
    M4 r;
    std::vector<int> inl;
    const double mse = fit_rigid_ransac(points_of(P, n), points_of(Q, n), r, inl, max_iterations, threshold,
                                        recompute_with_inliers != 0, seed);
    std::memcpy(T, r.m, sizeof(r.m));
    if (n_inliers) *n_inliers = (int32_t)inl.size();
    if (inliers)
      for (size_t k = 0; k < inl.size(); ++k) inliers[k] = inl[k];
    if (mse_inliers) *mse_inliers = mse;
    return SDM_OK;
  } catch (...) {
    return SDM_ERR_CAPACITY;
  }
}

}  // extern "C"
