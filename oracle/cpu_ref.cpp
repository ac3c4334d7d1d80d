/*
 * oracle/cpu_ref.cpp — CPU oracle for the particle-grid update hot path of
 * tud-amr/semantic_dsp_map.
 *
 * TEST INFRASTRUCTURE ONLY (see cpu_ref.h).  PARITY UNPINNED: the reference
 * has no tests/golden vectors and cannot be built here; this file restates
 * its loops literally, single-threaded, with run-time grid dimensions and
 * without Eigen/OpenCV/PCL.  Every function cites the reference lines it
 * follows (paths relative to /root/reference/include).
 *
 * Where the reference relies on implementation-defined / undefined behaviour
 * or on un-vendored third-party arithmetic (Eigen, libstdc++), the choice made
 * here is marked "PINNED:" and is listed in DESIGN.md.
 *
 * Build: g++ -std=c++17 -O3 -ftree-vectorize -march=native -ffp-contract=off
 * (the reference's flags, CMakeLists.txt:5-8, plus contraction off so that
 * float results are reproducible on the GPU).
 */
#include "cpu_ref.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <queue>
#include <set>
#include <unordered_set>
#include <vector>

namespace {

constexpr uint32_t INVALID_PARTICLE_INDEX = 0xffffffffu;  // mc_ring/operations.h:35
constexpr float C_PARTICLE_OCC_INIT_WEIGHT = 0.05f;        // settings/settings.h:147
constexpr float c_min_rightly_updated_pdf = 0.1f;          // settings/settings.h:149
constexpr float g_depth_error_stddev_at_one_meter = 0.1f;  // settings/settings.h:150
constexpr int GAUSSIAN_PDF_NUM = 20000;                    // utils/basic_algorithms.h:378
constexpr uint16_t OWNER_NONE = 0xFFFF;

// mc_ring/buffer.h:43-50
enum Status : uint8_t { INVALID = 0, UPDATED = 1, REGULAR_BORN = 2, GUESSED_BORN = 3, COPIED = 4, TIMEPTC = 5 };

// mc_ring/buffer.h:57-79 (24 bytes)
struct Particle {
  float x, y, z, weight;
  uint16_t time_stamp;
  uint16_t track_id;
  uint16_t label_id;
  uint8_t status;
  uint8_t forget_count;
};

using Clock = std::chrono::steady_clock;
inline double ms_since(Clock::time_point t0) {
  return std::chrono::duration<double, std::milli>(Clock::now() - t0).count();
}

}  // namespace

struct oracle_map {
  oracle_config cfg{};
  oracle_params prm{};

  // derived grid constants (mc_ring/buffer.h:26-41, operations.h:730-766)
  uint32_t NX = 0, NY = 0, NZ = 0, S = 0, V = 0;
  float map_p_min[3]{}, map_p_max[3]{}, voxel_size_recip = 0.f;
  float tan_half_fovx = 0.f, tan_half_fovy = 0.f;

  // mc_ring/buffer.h:86-120
  std::vector<Particle> P;
  std::vector<uint32_t> stamps_x, stamps_y, stamps_z;
  int moved_steps[3]{}, eq_steps[3]{};
  float map_center[3]{}, ego_center[3]{};
  float last_pos[3]{};              // function-static in operations.h:70
  uint32_t global_time_stamp = 0;  // utils/data_base.h:22

  // per-pixel particle bins (buffer.h:90-93)
  std::vector<std::vector<uint32_t>> bins;
  std::vector<uint32_t> bin_num;

  // tables (basic_algorithms.h:377-461)
  std::vector<float> noise;
  std::vector<float> pdf;
  int birth_cursor = 0;  // SemanticDSPMap::gaussian_random_ (semantic_dsp_map.h:265)
  int move_cursor = 0;   // RingBufferOperations::gaussian_random_calculator_ (operations.h:1460)
  float forgetting_function[5]{};
  bool forgetting_initialized = false;

  // ObjectParticleHashMap (object_layer.h:20-52). PINNED: ordered containers so
  // that iteration is ascending particle index (the reference iterates
  // std::unordered_set order, operations.h:334).
  std::map<int, std::set<uint32_t>> owner;
  std::vector<uint16_t> owner_shadow;  // last inserting track per index, alias detection only

  std::vector<float> ck_kappa;
  std::vector<oracle_voxel_result> result;
  float extrinsic[16]{};
  oracle_stats stats{};

  // ------------------------------------------------------------------ init
  explicit oracle_map(const oracle_config &c) : cfg(c) {
    NX = 1u << cfg.x_n;
    NY = 1u << cfg.y_n;
    NZ = 1u << cfg.z_n;
    S = 1u << cfg.p_n;
    V = NX * NY * NZ;
    P.resize(size_t(V) * S);
    memset(P.data(), 0, P.size() * sizeof(Particle));
    stamps_x.assign(NX, 0);
    stamps_y.assign(NY, 0);
    stamps_z.assign(NZ, 0);
    bins.resize(size_t(cfg.width) * cfg.height);
    bin_num.assign(size_t(cfg.width) * cfg.height, 0);
    ck_kappa.assign(size_t(cfg.width) * cfg.height, 0.f);
    result.resize(V);
    owner_shadow.assign(size_t(V) * S, OWNER_NONE);
    // default parameters: SemanticDSPMap ctor, semantic_dsp_map.h:25-42
    prm.detection_probability = 0.95f;
    prm.noise_number = 0.1f;
    prm.nb_ptc_num_per_point = 3;
    prm.occupancy_threshold = 0.2f;
    prm.max_obersevation_lost_time = 5;
    prm.forgetting_rate = 1.0f;
    prm.max_forget_count = 5;
    prm.match_score_threshold = 0.3f;
    prm.id_transition_probability = 0.1f;
    prm.if_consider_depth_noise = 0;
    prm.if_use_independent_filter = 0;
    prm.depth_noise_first_order = 0.0f;
    prm.depth_noise_zero_order = 0.1f;
    initialize();
    calculatePdfTable();
    // isPointInFrustum statics, operations.h:1249-1250
    tan_half_fovx = (float)tan(atan2(cfg.width / 2.0, (double)cfg.fx));
    tan_half_fovy = (float)tan(atan2(cfg.height / 2.0, (double)cfg.fy));
  }

  // operations.h:684-723
  void clear() {
    global_time_stamp = 0;
    std::fill(stamps_x.begin(), stamps_x.end(), 0u);
    std::fill(stamps_y.begin(), stamps_y.end(), 0u);
    std::fill(stamps_z.begin(), stamps_z.end(), 0u);
    for (uint32_t i = 0; i < V; ++i) {
      size_t start = size_t(i) << cfg.p_n;
      Particle &t = P[start];
      t.status = TIMEPTC;
      t.x = t.y = t.z = t.weight = 0.f;
      t.time_stamp = 0;
      for (uint32_t j = 1; j < S; ++j) {
        Particle &p = P[start + j];
        p.status = INVALID;
        p.x = p.y = p.z = p.weight = 0.f;
        p.time_stamp = 0;
      }
    }
  }

  // operations.h:726-767
  void initialize() {
    clear();
    for (int a = 0; a < 3; ++a) moved_steps[a] = eq_steps[a] = 0;
    map_p_max[0] = (NX >> 1) * cfg.voxel_size;
    map_p_max[1] = (NY >> 1) * cfg.voxel_size;
    map_p_max[2] = (NZ >> 1) * cfg.voxel_size;
    for (int a = 0; a < 3; ++a) map_p_min[a] = -map_p_max[a];
    voxel_size_recip = 1.f / cfg.voxel_size;
    for (int a = 0; a < 3; ++a) map_center[a] = ego_center[a] = 0.f;
  }

  // SemanticDSPMap::clear, semantic_dsp_map.h:74-81
  void clearAll() {
    clear();
    owner.clear();
    std::fill(owner_shadow.begin(), owner_shadow.end(), OWNER_NONE);
  }

  // basic_algorithms.h:405-407, 456-460
  void calculatePdfTable() {
    pdf.resize(GAUSSIAN_PDF_NUM);
    const float m_pi_2f32 = 1.5707964f;  // glibc M_PI_2f32 (float)
    for (int i = 0; i < GAUSSIAN_PDF_NUM; ++i) {
      float value = (float)(i - GAUSSIAN_PDF_NUM / 2) * 0.001f;
      pdf[i] = (1.f / (sqrtf(2.f * m_pi_2f32))) * expf(-powf(value, 2) / (2));
    }
  }

  // basic_algorithms.h:417-422. PINNED: NaN (0/0) returns 1e-9f (reference: UB cast).
  inline float queryNormalPDF(float x, float mu, float sigma) const {
    float corrected_x = (x - mu) / sigma;
    if (!(corrected_x <= 9.9f && corrected_x >= -9.9f)) return 1e-9f;
    return pdf[static_cast<int>(corrected_x * 1000 + 10000)];
  }

  // basic_algorithms.h:426-433 (pre-increment, wrap to 0)
  inline float queryNoise(int &cursor) const {
    cursor += 1;
    if (cursor >= (int)noise.size()) cursor = 0;
    return noise[cursor];
  }

  // basic_algorithms.h:32-48 (table frozen at first call).
  // PINNED: forget_count >= 5 with max_forget_count > 5 reads out of bounds in the reference; returns 0 here.
  inline float getForgettingFactor(int forget_count) {
    if (!forgetting_initialized) {
      for (int i = 0; i < 5; ++i) forgetting_function[i] = (float)pow(2.5, -i / prm.forgetting_rate);
      forgetting_initialized = true;
    }
    if (forget_count < prm.max_forget_count && forget_count < 5) return forgetting_function[forget_count];
    return 0.f;
  }

  // -------------------------------------------------------------- index math
  // PINNED: float -> index cast. operations.h:867-869 casts a float to uint32_t;
  // values in (-1,0) truncate to 0 and are accepted, values <= -1 are UB
  // (x86: huge -> rejected), values >= N rejected.
  static inline bool floatToIdx(float f, uint32_t n, uint32_t &out) {
    if (!(f > -1.0f && f < (float)n)) return false;
    out = (uint32_t)(int32_t)f;
    return out < n;
  }

  // operations.h:1037-1070
  static inline uint32_t axisCorrect(int idx, uint32_t n) {
    if (idx < 0) return (uint32_t)(idx + (int)n);
    if (idx >= (int)n) return (uint32_t)(idx - (int)n);
    return (uint32_t)idx;
  }

  // operations.h:994-1019
  inline void mapToRing(int mx, int my, int mz, uint32_t &rx, uint32_t &ry, uint32_t &rz) const {
    rx = axisCorrect(mx + eq_steps[0], NX);
    ry = axisCorrect(my + eq_steps[1], NY);
    rz = axisCorrect(mz + eq_steps[2], NZ);
  }
  // operations.h:1022-1033
  inline void ringToMap(uint32_t rx, uint32_t ry, uint32_t rz, uint32_t &mx, uint32_t &my, uint32_t &mz) const {
    mx = axisCorrect((int)rx - eq_steps[0], NX);
    my = axisCorrect((int)ry - eq_steps[1], NY);
    mz = axisCorrect((int)rz - eq_steps[2], NZ);
  }
  // operations.h:890-923 (row-major, STORAGE_TYPE 0)
  inline uint32_t ringToVoxel(uint32_t rx, uint32_t ry, uint32_t rz) const {
    return (((rz << cfg.y_n) | ry) << cfg.x_n) | rx;
  }
  // operations.h:947-967
  inline void voxelToRing(uint32_t v, uint32_t &rx, uint32_t &ry, uint32_t &rz) const {
    rx = v & (NX - 1);
    ry = (v >> cfg.x_n) & (NY - 1);
    rz = (v >> (cfg.x_n + cfg.y_n)) & (NZ - 1);
  }

  // operations.h:849-883 (global frame position -> voxel + ring indices)
  inline void globalPosToVoxel(float px, float py, float pz, uint32_t &voxel, uint32_t &rx, uint32_t &ry,
                               uint32_t &rz) const {
    float mx = px - map_center[0], my = py - map_center[1], mz = pz - map_center[2];
    uint32_t ix = 0, iy = 0, iz = 0;
    bool ok = floatToIdx((mx - map_p_min[0]) * voxel_size_recip, NX, ix);
    ok = floatToIdx((my - map_p_min[1]) * voxel_size_recip, NY, iy) && ok;
    ok = floatToIdx((mz - map_p_min[2]) * voxel_size_recip, NZ, iz) && ok;
    if (ok) {
      mapToRing((int)ix, (int)iy, (int)iz, rx, ry, rz);
      voxel = ringToVoxel(rx, ry, rz);
    } else {
      voxel = INVALID_PARTICLE_INDEX;
    }
  }

  // operations.h:970-983 + 940-944: voxel min corner in the global frame
  inline void voxelToGlobalPos(uint32_t v, float out[3]) const {
    uint32_t rx, ry, rz, mx, my, mz;
    voxelToRing(v, rx, ry, rz);
    ringToMap(rx, ry, rz, mx, my, mz);
    out[0] = mx * cfg.voxel_size + map_p_min[0];
    out[1] = my * cfg.voxel_size + map_p_min[1];
    out[2] = mz * cfg.voxel_size + map_p_min[2];
    out[0] += map_center[0];
    out[1] += map_center[1];
    out[2] += map_center[2];
  }

  // operations.h:810-816
  inline bool isParticleVacant(const Particle &p, uint32_t rx, uint32_t ry, uint32_t rz) const {
    return p.status == INVALID || p.time_stamp < stamps_x[rx] || p.time_stamp < stamps_y[ry] ||
           p.time_stamp < stamps_z[rz];
  }
  // operations.h:824-837
  inline bool isVoxelValid(uint32_t v, uint32_t rx, uint32_t ry, uint32_t rz) const {
    const Particle &t = P[size_t(v) << cfg.p_n];
    if (t.time_stamp == 0) return false;
    if (t.time_stamp < stamps_x[rx] || t.time_stamp < stamps_y[ry] || t.time_stamp < stamps_z[rz]) return false;
    return true;
  }

  // operations.h:782-803
  inline uint32_t addParticleByGlobalPos(const Particle &particle, uint32_t &voxel_index) {
    uint32_t rx, ry, rz;
    globalPosToVoxel(particle.x, particle.y, particle.z, voxel_index, rx, ry, rz);
    if (voxel_index != INVALID_PARTICLE_INDEX) {
      uint32_t start = voxel_index << cfg.p_n;
      for (uint32_t i = 1; i < S; ++i) {
        Particle *p_ori = &P[start + i];
        if (isParticleVacant(*p_ori, rx, ry, rz)) {
          *p_ori = particle;
          return start + i;
        }
      }
      return INVALID_PARTICLE_INDEX;
    }
    return INVALID_PARTICLE_INDEX;
  }

  // operations.h:171-184
  inline void addNewParticleWithSemantics(float x, float y, float z, uint8_t label, uint16_t track,
                                          uint32_t &voxel_index, uint32_t &particle_index) {
    Particle particle;
    particle.status = REGULAR_BORN;
    particle.time_stamp = (uint16_t)global_time_stamp;
    particle.forget_count = 0;
    particle.x = x;
    particle.y = y;
    particle.z = z;
    particle.weight = C_PARTICLE_OCC_INIT_WEIGHT;
    particle.label_id = label;
    particle.track_id = track;
    particle_index = addParticleByGlobalPos(particle, voxel_index);
  }

  // ------------------------------------------------------------ owner sets
  inline void ownerInsert(int track, uint32_t idx) {  // object_layer.h:31-33
    uint16_t prev = owner_shadow[idx];
    if (prev != OWNER_NONE && prev != (uint16_t)track) {
      auto it = owner.find(prev);
      if (it != owner.end() && it->second.count(idx)) stats.alias_events++;
    }
    owner[track].insert(idx);
    owner_shadow[idx] = (uint16_t)track;
  }
  inline void ownerErase(int track, uint32_t idx) {  // object_layer.h:35-37
    owner[track].erase(idx);
  }

  // ------------------------------------------------------------- A1 ego shift
  // operations.h:1111-1191, 1196-1230
  void updateRingbufferIndexParams() {
    int steps[3];
    for (int a = 0; a < 3; ++a) steps[a] = static_cast<int>(ego_center[a] * voxel_size_recip);
    for (int a = 0; a < 3; ++a) map_center[a] = static_cast<float>(steps[a]) * cfg.voxel_size;
    uint32_t N[3] = {NX, NY, NZ};
    std::vector<uint32_t> *st[3] = {&stamps_x, &stamps_y, &stamps_z};
    for (int a = 0; a < 3; ++a) {
      int new_moved = steps[a] - moved_steps[a];
      if (new_moved > 0) {
        for (int i = 0; i < new_moved; ++i) {
          int idx = i + eq_steps[a];
          idx = (int)axisCorrect(idx, N[a]);
          (*st[a])[idx] = global_time_stamp;
        }
      } else if (new_moved < 0) {
        for (int i = 0; i < -new_moved; ++i) {
          int idx = (int)N[a] - 1 - i + eq_steps[a];
          idx = (int)axisCorrect(idx, N[a]);
          (*st[a])[idx] = global_time_stamp;
        }
      }
    }
    for (int a = 0; a < 3; ++a) {
      moved_steps[a] = steps[a];
      int o = steps[a];
      if (o > 0) eq_steps[a] = o % (int)N[a];
      else if (o < 0) eq_steps[a] = -(-o % (int)N[a]);
      else eq_steps[a] = 0;
    }
  }

  // operations.h:68-96. PINNED: norm = sqrt((x*x + y*y) + z*z); normalized() = v / norm if norm > 0.
  void updateEgoCenterPos(const float pos[3]) {
    const float mx = (1 << (cfg.x_n - 2)) * cfg.voxel_size;
    const float my = (1 << (cfg.y_n - 2)) * cfg.voxel_size;
    const float mz = (1 << (cfg.z_n - 2)) * cfg.voxel_size;
    const float max_once = std::min(std::min(mx, my), mz);
    float mv[3] = {pos[0] - last_pos[0], pos[1] - last_pos[1], pos[2] - last_pos[2]};
    float sq = (mv[0] * mv[0] + mv[1] * mv[1]) + mv[2] * mv[2];
    float dist = sqrtf(sq);
    float unit[3] = {mv[0], mv[1], mv[2]};
    if (sq > 0.f) {
      unit[0] = mv[0] / dist;
      unit[1] = mv[1] / dist;
      unit[2] = mv[2] / dist;
    }
    float new_pos[3] = {last_pos[0], last_pos[1], last_pos[2]};
    while (dist > max_once) {
      for (int a = 0; a < 3; ++a) new_pos[a] = new_pos[a] + unit[a] * max_once;
      for (int a = 0; a < 3; ++a) ego_center[a] = new_pos[a];
      updateRingbufferIndexParams();
      for (int a = 0; a < 3; ++a) mv[a] = pos[a] - new_pos[a];
      dist = sqrtf((mv[0] * mv[0] + mv[1] * mv[1]) + mv[2] * mv[2]);
    }
    for (int a = 0; a < 3; ++a) ego_center[a] = pos[a];
    updateRingbufferIndexParams();
    for (int a = 0; a < 3; ++a) last_pos[a] = pos[a];
  }

  // PINNED: 4x4 (row-major) times [x y z 1]: ((m0*x + m1*y) + m2*z) + m3
  static inline float row4(const float *r, float x, float y, float z) {
    return ((r[0] * x + r[1] * y) + r[2] * z) + r[3];
  }

  // -------------------------------------------------------- A5 object moves
  // semantic_dsp_map.h:588-699 (collection) + operations.h:321-362
  void moveObjects(const oracle_object_move *moves, int n_moves) {
    std::vector<int> tracks;
    std::vector<std::vector<uint32_t>> idx_sets;
    std::vector<const float *> mats;
    for (int k = 0; k < n_moves; ++k) {
      int track = moves[k].track_id;
      auto it = owner.find(track);
      if (it == owner.end()) continue;  // checkIfObjectExists, semantic_dsp_map.h:611
      tracks.push_back(track);
      idx_sets.emplace_back(it->second.begin(), it->second.end());
      mats.push_back(moves[k].T);
    }
    if (idx_sets.empty()) return;  // operations.h:323-325
    std::vector<std::vector<Particle>> new_particles(idx_sets.size());
    for (size_t i = 0; i < idx_sets.size(); ++i) {
      new_particles[i].resize(idx_sets[i].size());
      const float *T = mats[i];
      size_t j = 0;
      for (uint32_t idx : idx_sets[i]) {
        new_particles[i][j] = P[idx];
        float ox = P[idx].x, oy = P[idx].y, oz = P[idx].z;
        float nx = row4(T + 0, ox, oy, oz);
        float ny = row4(T + 4, ox, oy, oz);
        float nz = row4(T + 8, ox, oy, oz);
        new_particles[i][j].x = nx + queryNoise(move_cursor);
        new_particles[i][j].y = ny + queryNoise(move_cursor);
        new_particles[i][j].z = nz + queryNoise(move_cursor);
        P[idx].status = INVALID;  // deleteParticleByIndex
        ++j;
        stats.n_moved++;
      }
    }
    for (size_t i = 0; i < idx_sets.size(); ++i) {
      std::set<uint32_t> new_set;
      for (size_t j = 0; j < new_particles[i].size(); ++j) {
        uint32_t voxel_index;
        uint32_t pi = addParticleByGlobalPos(new_particles[i][j], voxel_index);
        if (pi != INVALID_PARTICLE_INDEX) {
          new_set.insert(pi);
          stats.n_move_reinserted++;
        }
      }
      // updatePtcIndicesOfObj, semantic_dsp_map.h:697-699
      owner[tracks[i]] = std::move(new_set);
    }
    // alias bookkeeping (diagnostic only): an index that now sits in two objects' sets
    for (size_t i = 0; i < idx_sets.size(); ++i) {
      for (uint32_t idx : owner[tracks[i]]) {
        uint16_t prev = owner_shadow[idx];
        if (prev != OWNER_NONE && prev != (uint16_t)tracks[i]) {
          auto it = owner.find(prev);
          if (it != owner.end() && it->second.count(idx)) stats.alias_events++;
        }
        owner_shadow[idx] = (uint16_t)tracks[i];
      }
    }
  }

  // object_layer.h:414-425
  void removeObject(int track) {
    auto it = owner.find(track);
    if (it == owner.end()) return;
    for (uint32_t idx : it->second) P[idx].status = INVALID;
    owner.erase(it);
  }

  // ------------------------------------------------------- A6 visibility
  // operations.h:1240-1258
  inline bool isPointInFrustum(float px, float py, float pz) const {
    float cx_ = row4(extrinsic + 0, px, py, pz);
    float cy_ = row4(extrinsic + 4, px, py, pz);
    float cz_ = row4(extrinsic + 8, px, py, pz);
    if (cz_ < cfg.depth_min || cz_ > cfg.depth_max) return false;
    if (std::fabs(cx_) > cz_ * tan_half_fovx) return false;
    if (std::fabs(cy_) > cz_ * tan_half_fovy) return false;
    return true;
  }

  // operations.h:1267-1290. PINNED: K*p/z evaluated as (fx*x + cx*z)/z, (fy*y + cy*z)/z
  inline bool projectToImage(float px, float py, float pz, int &row, int &col, float &cam_z) const {
    float x = row4(extrinsic + 0, px, py, pz);
    float y = row4(extrinsic + 4, px, py, pz);
    float z = row4(extrinsic + 8, px, py, pz);
    if (z < cfg.depth_min || z > cfg.depth_max) return false;
    float u = (cfg.fx * x + cfg.cx * z) / z;
    float v = (cfg.fy * y + cfg.cy * z) / z;
    row = static_cast<int>(v);
    col = static_cast<int>(u);
    if (row < 0 || row >= cfg.height || col < 0 || col >= cfg.width) return false;
    cam_z = z;
    return true;
  }

  // operations.h:653-667, 1297-1457
  void updateVisibleParticlesWithBFS(const float *depth) {
    const int W = cfg.width, H = cfg.height;
    std::fill(bin_num.begin(), bin_num.end(), 0u);
    for (auto &b : bins) b.clear();

    const int VX = NX + 1, VY = NY + 1, VZ = NZ + 1;
    std::vector<uint8_t> visited(size_t(VX) * VY * VZ, 0);
    std::vector<uint8_t> added(size_t(NX) * NY * NZ, 0);
    auto vid = [&](int x, int y, int z) { return (size_t(x) * VY + y) * VZ + z; };
    auto aid = [&](int x, int y, int z) { return (size_t(x) * NY + y) * NZ + z; };
    struct V3 { int x, y, z; };
    std::queue<V3> q;

    float off[3] = {map_center[0] + map_p_min[0], map_center[1] + map_p_min[1], map_center[2] + map_p_min[2]};

    // start vertex: camera-frame (0,0,1) in the global frame (operations.h:1312-1321).
    // PINNED: inverse of the extrinsic is the camera pose itself: p + R*(0,0,1).
    float sg[3] = {cam_R[2] + cam_p[0], cam_R[5] + cam_p[1], cam_R[8] + cam_p[2]};
    float sm[3] = {sg[0] - map_center[0], sg[1] - map_center[1], sg[2] - map_center[2]};
    V3 s{static_cast<int>((sm[0] + map_p_max[0]) * voxel_size_recip),
         static_cast<int>((sm[1] + map_p_max[1]) * voxel_size_recip),
         static_cast<int>((sm[2] + map_p_max[2]) * voxel_size_recip)};
    // PINNED: a start vertex outside the vertex grid indexes out of bounds in the reference; nothing is updated here.
    if (s.x < 0 || s.x > (int)NX || s.y < 0 || s.y > (int)NY || s.z < 0 || s.z > (int)NZ) return;
    q.push(s);
    bool first = true;
    const float one_sigma_error_coeff = g_depth_error_stddev_at_one_meter + 1.f;

    while (!q.empty()) {
      V3 c = q.front();
      q.pop();
      if (visited[vid(c.x, c.y, c.z)]) continue;
      visited[vid(c.x, c.y, c.z)] = 1;
      float gx = (float)c.x * cfg.voxel_size + off[0];
      float gy = (float)c.y * cfg.voxel_size + off[1];
      float gz = (float)c.z * cfg.voxel_size + off[2];
      bool in = isPointInFrustum(gx, gy, gz);
      if (first) {
        stats.bfs_start_in_frustum = in ? 1 : 0;
        first = false;
      }
      if (!in) continue;
      for (int dx = -1; dx <= 0; dx++)
        for (int dy = -1; dy <= 0; dy++)
          for (int dz = -1; dz <= 0; dz++) {
            int ax = c.x + dx, ay = c.y + dy, az = c.z + dz;
            if (ax < 0 || ax >= (int)NX || ay < 0 || ay >= (int)NY || az < 0 || az >= (int)NZ) continue;
            if (added[aid(ax, ay, az)]) continue;
            uint32_t rx, ry, rz;
            mapToRing(ax, ay, az, rx, ry, rz);
            uint32_t voxel_idx = ringToVoxel(rx, ry, rz);
            uint32_t start = voxel_idx << cfg.p_n;
            bool voxel_observed = false;
            int valid_particle_num_in_voxel = 0;
            for (uint32_t i = 1; i < S; ++i) {
              Particle *ptc = &P[start + i];
              if (ptc->status != INVALID) {
                if (ptc->time_stamp < stamps_x[rx] || ptc->time_stamp < stamps_y[ry] || ptc->time_stamp < stamps_z[rz]) {
                  ptc->status = INVALID;
                  continue;
                }
                valid_particle_num_in_voxel++;
                int row, col;
                float cam_z;
                if (projectToImage(ptc->x, ptc->y, ptc->z, row, col, cam_z)) {
                  float d = depth[size_t(row) * W + col];
                  if (d > cfg.depth_max) {
                    ptc->weight = C_PARTICLE_OCC_INIT_WEIGHT;
                    voxel_observed = true;
                    continue;
                  }
                  if (cam_z > d * one_sigma_error_coeff) continue;
                  voxel_observed = true;
                  int id = row * W + col;
                  bins[id].push_back(start + i);
                  bin_num[id]++;
                  stats.n_visible++;
                }
              }
            }
            if (voxel_observed) {
              P[start].time_stamp = (uint16_t)global_time_stamp;
            } else if (valid_particle_num_in_voxel == 0) {
              // mapXYZIdxToGlobalPose, operations.h:986-991
              float ix = (uint32_t)ax * cfg.voxel_size + map_p_min[0] + map_center[0];
              float iy = (uint32_t)ay * cfg.voxel_size + map_p_min[1] + map_center[1];
              float iz = (uint32_t)az * cfg.voxel_size + map_p_min[2] + map_center[2];
              int row, col;
              float cam_z;
              if (projectToImage(ix, iy, iz, row, col, cam_z)) {
                if (cam_z <= depth[size_t(row) * W + col]) P[start].time_stamp = (uint16_t)global_time_stamp;
              }
            }
            added[aid(ax, ay, az)] = 1;
            stats.n_frustum_voxels++;
          }
      static const int dir[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
      for (int i = 0; i < 6; ++i) {
        int nx = c.x + dir[i][0], ny = c.y + dir[i][1], nz = c.z + dir[i][2];
        if (nx < 0 || nx > (int)NX || ny < 0 || ny > (int)NY || nz < 0 || nz > (int)NZ || visited[vid(nx, ny, nz)]) continue;
        q.push(V3{nx, ny, nz});
      }
    }
    if (cfg.bin_order == 1) {
      for (auto &b : bins)
        if (b.size() > 1) std::sort(b.begin(), b.end());
    }
  }

  // ---------------------------------------------------- A7 weight update
  // semantic_dsp_map.h:960-1121
  void updateParticles(const oracle_labeled_point *cloud) {
    const int W = cfg.width, H = cfg.height;
    const int h = cfg.window_half;
    const bool indep = prm.if_use_independent_filter != 0;
    const int slabs = cfg.ck_slabs > 1 ? cfg.ck_slabs : 1;
    const uint32_t slab_len = NZ / slabs;
    std::vector<float> partial(slabs), row_partial(slabs);
    const bool canonical = cfg.bin_order == 1 || slabs > 1;
    for (int i = 0; i < H; ++i) {
      for (int j = 0; j < W; ++j) {
        const oracle_labeled_point &o = cloud[size_t(i) * W + j];
        if (!o.is_valid) continue;
        float sigma_this_pixel = o.sigma;
        float ck_this_pixel = 0.f;
        if (slabs > 1) std::fill(partial.begin(), partial.end(), 0.f);
        for (int m = -h; m <= h; ++m) {
          // canonical mode: every window row is summed on its own and the row sums are added in row order
          // (the GPU evaluates the rows in parallel); literal mode: one running sum like the reference.
          float row_sum = 0.f;
          if (slabs > 1) std::fill(row_partial.begin(), row_partial.end(), 0.f);
          for (int n = -h; n <= h; ++n) {
            int ni = i + m, nj = j + n;
            if (ni < 0 || ni >= H || nj < 0 || nj >= W) continue;
            int num = (int)bin_num[size_t(ni) * W + nj];
            if (num > 0) {
              const std::vector<uint32_t> &bin = bins[size_t(ni) * W + nj];
              for (int l = 0; l < num; ++l) {
                const Particle *particle = &P[bin[l]];
                if (indep) {
                  if (particle->track_id != o.track_id) continue;
                }
                float gk = queryNormalPDF(particle->x, o.x, sigma_this_pixel) *
                           queryNormalPDF(particle->y, o.y, sigma_this_pixel) *
                           queryNormalPDF(particle->z, o.z, sigma_this_pixel);
                if (!indep) {
                  gk *= getForgettingFactor(particle->forget_count);
                  if (particle->track_id != o.track_id) gk *= prm.id_transition_probability;
                }
                if (slabs > 1) {
                  uint32_t rz = (bin[l] >> cfg.p_n) >> (cfg.x_n + cfg.y_n);
                  row_partial[rz / slab_len] += particle->weight * gk;
                } else if (canonical) {
                  row_sum += particle->weight * gk;
                } else {
                  ck_this_pixel += particle->weight * gk;
                }
              }
            }
          }
          if (slabs > 1) {
            for (int g = 0; g < slabs; ++g) partial[g] += row_partial[g];
          } else if (canonical) {
            ck_this_pixel += row_sum;
          }
        }
        if (slabs > 1) {
          for (int g = 0; g < slabs; ++g) ck_this_pixel += partial[g];
        }
        ck_kappa[size_t(i) * W + j] = ck_this_pixel * prm.detection_probability + prm.noise_number;
      }
    }

    for (int i = 0; i < H; ++i) {
      for (int j = 0; j < W; ++j) {
        float sigma_this_pixel = cloud[size_t(i) * W + j].sigma;
        int num = (int)bin_num[size_t(i) * W + j];
        if (num <= 0) continue;
        const std::vector<uint32_t> &bin = bins[size_t(i) * W + j];
        for (int l = 0; l < num; ++l) {
          Particle *particle = &P[bin[l]];
          float acc = 0.f;
          bool updated_with_right_id = false;
          for (int m = -h; m <= h; ++m) {
            float row_acc = 0.f;  // canonical mode: per-row partial sums, see pass 1
            for (int n = -h; n <= h; ++n) {
              int ni = i + m, nj = j + n;
              if (ni < 0 || ni >= H || nj < 0 || nj >= W) continue;
              const oracle_labeled_point &o = cloud[size_t(ni) * W + nj];
              if (!o.is_valid) continue;
              if (indep) {
                if (o.track_id != particle->track_id) continue;
              }
              float gk = queryNormalPDF(particle->x, o.x, sigma_this_pixel) *
                         queryNormalPDF(particle->y, o.y, sigma_this_pixel) *
                         queryNormalPDF(particle->z, o.z, sigma_this_pixel);
              if (!indep) {
                if (particle->track_id != o.track_id) {
                  gk *= prm.id_transition_probability;
                } else {
                  if (gk > c_min_rightly_updated_pdf) updated_with_right_id = true;
                }
                gk *= getForgettingFactor(particle->forget_count);
              }
              if (canonical) row_acc += gk / ck_kappa[size_t(ni) * W + nj];
              else acc += gk / ck_kappa[size_t(ni) * W + nj];
            }
            if (canonical) acc += row_acc;
          }
          particle->weight *= (acc * prm.detection_probability + 1.f - prm.detection_probability);
          particle->status = UPDATED;
          particle->time_stamp = (uint16_t)global_time_stamp;
          if (!indep) {
            if (updated_with_right_id) {
              particle->forget_count = 0;
            } else {
              if (particle->forget_count < 5) particle->forget_count += 1;
            }
          }
        }
      }
    }
  }

  // ------------------------------------------------------------ A9 resample
  // semantic_dsp_map.h:1448-1519
  bool resampleParticlesInVoxel(uint32_t voxel_index) {
    uint32_t start = voxel_index << cfg.p_n;
    float weight_sum = 0.f;
    uint32_t updated_particle_num = 0;
    for (uint32_t i = 1; i < S; ++i) {
      if (P[start + i].status == UPDATED) {
        weight_sum += P[start + i].weight;
        ++updated_particle_num;
      }
    }
    const uint32_t resample_triger_ptc_num = S >> 1;
    if (updated_particle_num > resample_triger_ptc_num) {
      if (weight_sum < 0.01f) {
        for (uint32_t i = 1; i < S; ++i) {
          uint32_t pi = start + i;
          if (P[pi].status == UPDATED) {
            int t = P[pi].track_id;
            P[pi].status = INVALID;
            ownerErase(t, pi);
          }
        }
        return true;
      }
      float weight_per_particle = weight_sum / resample_triger_ptc_num;
      if (weight_per_particle > 1.f) weight_per_particle = 1.f;
      float particle_weight_sum = 0.f;
      float threshold = weight_per_particle;
      for (uint32_t i = 1; i < S; ++i) {
        uint32_t pi = start + i;
        if (P[pi].status == UPDATED) {
          int t = P[pi].track_id;
          particle_weight_sum += P[pi].weight;
          if (particle_weight_sum < threshold) {
            P[pi].status = INVALID;
            ownerErase(t, pi);
          } else {
            P[pi].weight = weight_per_particle;
            threshold += weight_per_particle;
            while (particle_weight_sum > threshold) threshold += weight_per_particle;
          }
        }
      }
      return true;
    }
    return false;
  }

  // ------------------------------------------------------------- A8 births
  // semantic_dsp_map.h:1148-1171
  void addNewbornParticleAndResample(const oracle_labeled_point &pt, std::unordered_set<uint32_t> &resampled) {
    uint32_t voxel_idx, ptc_idx;
    addNewParticleWithSemantics(pt.x, pt.y, pt.z, pt.label_id, pt.track_id, voxel_idx, ptc_idx);
    stats.n_birth_attempts++;
    if (ptc_idx != INVALID_PARTICLE_INDEX) {
      stats.n_birth_success++;
      int track = pt.track_id;
      if (track <= cfg.max_movable_track) ownerInsert(track, ptc_idx);
    }
    if (voxel_idx != INVALID_PARTICLE_INDEX && resampled.count(voxel_idx) == 0) {
      if (resampleParticlesInVoxel(voxel_idx)) resampled.insert(voxel_idx);
    }
  }

  // semantic_dsp_map.h:1177-1230
  void addNewbornParticleWithNoiseAndResample(const oracle_labeled_point &pt, std::unordered_set<uint32_t> &resampled) {
    for (int n = 0; n < prm.nb_ptc_num_per_point; ++n) {
      float noise3[3] = {0.f, 0.f, 0.f};
      if (prm.nb_ptc_num_per_point != 1) {
        float sigma = pt.sigma;
        noise3[0] = sigma * queryNoise(birth_cursor);
        noise3[1] = sigma * queryNoise(birth_cursor);
        noise3[2] = sigma * queryNoise(birth_cursor);
      }
      float x = pt.x + noise3[0], y = pt.y + noise3[1], z = pt.z + noise3[2];
      uint32_t voxel_idx, ptc_idx;
      addNewParticleWithSemantics(x, y, z, pt.label_id, pt.track_id, voxel_idx, ptc_idx);
      stats.n_birth_attempts++;
      if (ptc_idx != INVALID_PARTICLE_INDEX) {
        stats.n_birth_success++;
        int track = pt.track_id;
        if (track <= cfg.max_movable_track) ownerInsert(track, ptc_idx);
      } else {
        if (voxel_idx != INVALID_PARTICLE_INDEX && resampled.count(voxel_idx) == 0) {
          if (resampleParticlesInVoxel(voxel_idx)) {
            resampled.insert(voxel_idx);
            addNewParticleWithSemantics(x, y, z, pt.label_id, pt.track_id, voxel_idx, ptc_idx);
            if (ptc_idx != INVALID_PARTICLE_INDEX) {
              stats.n_birth_success++;
              int track = pt.track_id;
              if (track <= cfg.max_movable_track) ownerInsert(track, ptc_idx);
            }
          }
        }
      }
    }
  }

  // semantic_dsp_map.h:768-800
  void birthLoop(const oracle_labeled_point *cloud) {
    const int W = cfg.width, H = cfg.height;
    std::unordered_set<uint32_t> resampled;
    const int selection_interval = 3;
    for (int row_start = 0; row_start < selection_interval; row_start++)
      for (int col_start = 0; col_start < selection_interval; col_start++)
        for (int i = row_start; i < H; i += selection_interval)
          for (int j = col_start; j < W; j += selection_interval) {
            const oracle_labeled_point &pt = cloud[size_t(i) * W + j];
            if (!pt.is_valid) continue;
            if (!prm.if_consider_depth_noise) addNewbornParticleAndResample(pt, resampled);
            else addNewbornParticleWithNoiseAndResample(pt, resampled);
          }
    stats.n_resampled_voxels = (int64_t)resampled.size();
  }

  // -------------------------------------------------------- A10 occupancy
  // operations.h:390-448, 623-639; semantic_dsp_map.h:1239-1257
  void occupancySweep() {
    for (uint32_t v = 0; v < V; ++v) {
      uint32_t start = v << cfg.p_n;
      uint32_t rx, ry, rz;
      voxelToRing(v, rx, ry, rz);
      // PINNED: label/track are uninitialised locals in the reference when no contributor wins; (0,0) here.
      uint8_t label_id = 0;
      uint16_t track_id = 0;
      float weight_sum, guessed_weight;
      if (!isVoxelValid(v, rx, ry, rz)) {
        weight_sum = -1.f;
        guessed_weight = 0.f;
      } else {
        std::map<uint16_t, float> track_id_weight_map;
        std::map<uint16_t, uint8_t> track_id_label_map;
        weight_sum = 0.f;
        guessed_weight = 0.f;
        for (uint32_t i = 1; i < S; ++i) {
          Particle &p = P[start + i];
          if (!isParticleVacant(p, rx, ry, rz)) {
            weight_sum += p.weight;
            if (p.weight > 1.f) p.weight = 1.f;
            if (p.status == GUESSED_BORN) {
              guessed_weight += p.weight;
            } else if (p.status == UPDATED && p.weight < C_PARTICLE_OCC_INIT_WEIGHT) {
              p.status = INVALID;
              continue;
            }
            if (track_id_weight_map.find(p.track_id) == track_id_weight_map.end()) track_id_weight_map[p.track_id] = 0.f;
            track_id_weight_map[p.track_id] += p.weight;
            track_id_label_map[p.track_id] = (uint8_t)p.label_id;
          }
        }
        float max_weight = 0.f;
        for (auto it = track_id_weight_map.begin(); it != track_id_weight_map.end(); ++it) {
          if (it->second > max_weight) {
            max_weight = it->second;
            track_id = it->first;
            label_id = track_id_label_map[it->first];
          }
        }
      }
      int occ;
      if (weight_sum > prm.occupancy_threshold) occ = 1;
      else if (weight_sum < 0) occ = -1;
      else if (guessed_weight >= C_PARTICLE_OCC_INIT_WEIGHT) occ = 2;
      else occ = 0;
      result[v].wsum = weight_sum;
      result[v].track = track_id;
      result[v].label = label_id;
      result[v].occ = (int8_t)occ;
      if (occ > 0) stats.n_occupied++;
    }
  }

  // ------------------------------------------------------------- frame
  float cam_R[9]{}, cam_p[3]{};

  // semantic_dsp_map.h:744-747. PINNED: Eigen's Quaternion::toRotationMatrix formula in float, and the
  // rigid inverse [R^T | -(R^T p)] instead of Eigen's general 4x4 inverse (version unpinned).
  void computeExtrinsic(const float pos[3], const float q[4]) {
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    const float tx = 2.f * x, ty = 2.f * y, tz = 2.f * z;
    const float twx = tx * w, twy = ty * w, twz = tz * w;
    const float txx = tx * x, txy = ty * x, txz = tz * x;
    const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
    float R[9];
    R[0] = 1.f - (tyy + tzz);
    R[1] = txy - twz;
    R[2] = txz + twy;
    R[3] = txy + twz;
    R[4] = 1.f - (txx + tzz);
    R[5] = tyz - twx;
    R[6] = txz - twy;
    R[7] = tyz + twx;
    R[8] = 1.f - (txx + tyy);
    memcpy(cam_R, R, sizeof(R));
    for (int a = 0; a < 3; ++a) cam_p[a] = pos[a];
    for (int r = 0; r < 3; ++r) {
      // row r of R^T = column r of R
      float a = R[0 * 3 + r], b = R[1 * 3 + r], c = R[2 * 3 + r];
      extrinsic[r * 4 + 0] = a;
      extrinsic[r * 4 + 1] = b;
      extrinsic[r * 4 + 2] = c;
      extrinsic[r * 4 + 3] = -((a * pos[0] + b * pos[1]) + c * pos[2]);
    }
    extrinsic[12] = extrinsic[13] = extrinsic[14] = 0.f;
    extrinsic[15] = 1.f;
  }

  int64_t countLive() const {
    int64_t n = 0;
    for (uint32_t v = 0; v < V; ++v) {
      uint32_t rx, ry, rz;
      voxelToRing(v, rx, ry, rz);
      for (uint32_t i = 1; i < S; ++i)
        if (!isParticleVacant(P[(size_t(v) << cfg.p_n) + i], rx, ry, rz)) n++;
    }
    return n;
  }

  int update(const float *depth, const oracle_labeled_point *cloud, const float cam_pos[3], const float cam_q[4],
             const oracle_object_move *moves, int n_moves, const int32_t *remove_tracks, int n_remove,
             int stop_after) {
    int64_t alias_keep = stats.alias_events;
    stats = oracle_stats{};
    stats.alias_events = alias_keep;
    if (noise.empty()) noise.assign(1000000, 0.f);
    global_time_stamp += 1;  // semantic_dsp_map.h:173
    auto done = [&](int s) { return stop_after != 0 && stop_after <= s; };

    auto t0 = Clock::now();
    updateEgoCenterPos(cam_pos);  // semantic_dsp_map.h:584-585
    stats.stage_ms[1] = ms_since(t0);
    if (done(1)) return 0;

    t0 = Clock::now();
    moveObjects(moves, n_moves);
    stats.stage_ms[2] = ms_since(t0);
    if (done(2)) return 0;

    t0 = Clock::now();
    for (int i = 0; i < n_remove; ++i) removeObject(remove_tracks[i]);  // semantic_dsp_map.h:702-736
    stats.stage_ms[3] = ms_since(t0);
    if (done(3)) return 0;

    t0 = Clock::now();
    computeExtrinsic(cam_pos, cam_q);
    updateVisibleParticlesWithBFS(depth);
    stats.stage_ms[4] = ms_since(t0);
    if (done(4)) return 0;

    t0 = Clock::now();
    updateParticles(cloud);
    stats.stage_ms[5] = ms_since(t0);
    if (done(5)) return 0;

    t0 = Clock::now();
    birthLoop(cloud);
    stats.stage_ms[6] = ms_since(t0);
    if (done(6)) return 0;

    t0 = Clock::now();
    occupancySweep();
    stats.stage_ms[7] = ms_since(t0);
    return 0;
  }
};

// ------------------------------------------------------------------ C ABI
// utils/pointcloud_tools.h:1104-1133
template <typename T>
static void manual_resize(const T *src, int src_w, int src_h, T *dst, int dst_w, int dst_h, float scale) {
  const float scale_inv = 1.f / scale;
  for (int i = 0; i < dst_h; ++i)
    for (int j = 0; j < dst_w; ++j) {
      int si = static_cast<int>(i * scale_inv), sj = static_cast<int>(j * scale_inv);
      si = std::min(si, src_h - 1);
      sj = std::min(sj, src_w - 1);
      dst[(size_t)i * dst_w + j] = src[(size_t)si * src_w + sj];
    }
}

extern "C" {

oracle_map *oracle_create(const oracle_config *cfg) {
  if (!cfg) return nullptr;
  if (cfg->x_n + cfg->y_n + cfg->z_n + cfg->p_n > 31) return nullptr;  // operations.h:54-58
  if (cfg->x_n < 2 || cfg->y_n < 2 || cfg->z_n < 2 || cfg->p_n < 1) return nullptr;
  return new oracle_map(*cfg);
}
void oracle_destroy(oracle_map *m) { delete m; }
// A deep copy of the whole map - particles, ring state, table cursors and the owner SETS as they are (an index can sit in
// two of them, object_layer.h:20-52: dump_state / load_state carry one owner per index and lose the second) - with another
// summation order for what follows: the way to part a literal-order run from a canonical-order one in mid-drive.
oracle_map *oracle_clone(const oracle_map *m, int32_t bin_order) {
  if (!m) return nullptr;
  oracle_map *c = new oracle_map(*m);
  c->cfg.bin_order = bin_order;
  return c;
}
void oracle_clear(oracle_map *m) { m->clearAll(); }
void oracle_set_params(oracle_map *m, const oracle_params *p) { m->prm = *p; }
void oracle_set_noise_table(oracle_map *m, const float *table, int32_t n) { m->noise.assign(table, table + n); }

int oracle_update(oracle_map *m, const float *depth, const oracle_labeled_point *cloud, const float cam_pos[3],
                  const float cam_q[4], const oracle_object_move *moves, int32_t n_moves,
                  const int32_t *remove_tracks, int32_t n_remove, int32_t stop_after) {
  return m->update(depth, cloud, cam_pos, cam_q, moves, n_moves, remove_tracks, n_remove, stop_after);
}

void oracle_get_voxels(oracle_map *m, oracle_voxel_result *out) {
  memcpy(out, m->result.data(), m->result.size() * sizeof(oracle_voxel_result));
}
void oracle_get_stats(oracle_map *m, oracle_stats *out) {
  m->stats.live_particles = m->countLive();
  *out = m->stats;
}
void oracle_get_ring_state(oracle_map *m, oracle_ring_state *o) {
  o->global_time_stamp = m->global_time_stamp;
  for (int a = 0; a < 3; ++a) {
    o->moved_steps[a] = m->moved_steps[a];
    o->eq_steps[a] = m->eq_steps[a];
    o->map_center[a] = m->map_center[a];
    o->last_pos[a] = m->last_pos[a];
  }
  o->birth_cursor = m->birth_cursor;
  o->move_cursor = m->move_cursor;
}
void oracle_set_ring_state(oracle_map *m, const oracle_ring_state *o) {
  m->global_time_stamp = o->global_time_stamp;
  for (int a = 0; a < 3; ++a) {
    m->moved_steps[a] = o->moved_steps[a];
    m->eq_steps[a] = o->eq_steps[a];
    m->map_center[a] = o->map_center[a];
    m->last_pos[a] = o->last_pos[a];
  }
  m->birth_cursor = o->birth_cursor;
  m->move_cursor = o->move_cursor;
}
void oracle_get_stamps(oracle_map *m, uint32_t *sx, uint32_t *sy, uint32_t *sz) {
  memcpy(sx, m->stamps_x.data(), m->NX * 4);
  memcpy(sy, m->stamps_y.data(), m->NY * 4);
  memcpy(sz, m->stamps_z.data(), m->NZ * 4);
}
void oracle_set_stamps(oracle_map *m, const uint32_t *sx, const uint32_t *sy, const uint32_t *sz) {
  memcpy(m->stamps_x.data(), sx, m->NX * 4);
  memcpy(m->stamps_y.data(), sy, m->NY * 4);
  memcpy(m->stamps_z.data(), sz, m->NZ * 4);
}

void oracle_dump_state(oracle_map *m, float *px, float *py, float *pz, float *w, uint16_t *ts, uint16_t *track,
                       uint8_t *label, uint8_t *status, uint8_t *forget, uint16_t *owner) {
  size_t n = m->P.size();
  for (size_t i = 0; i < n; ++i) {
    const Particle &p = m->P[i];
    if (px) px[i] = p.x;
    if (py) py[i] = p.y;
    if (pz) pz[i] = p.z;
    if (w) w[i] = p.weight;
    if (ts) ts[i] = p.time_stamp;
    if (track) track[i] = p.track_id;
    if (label) label[i] = (uint8_t)p.label_id;
    if (status) status[i] = p.status;
    if (forget) forget[i] = p.forget_count;
  }
  if (owner) {
    for (size_t i = 0; i < n; ++i) owner[i] = OWNER_NONE;
    for (auto &kv : m->owner)
      for (uint32_t idx : kv.second) owner[idx] = (uint16_t)kv.first;
  }
}
void oracle_load_state(oracle_map *m, const float *px, const float *py, const float *pz, const float *w,
                       const uint16_t *ts, const uint16_t *track, const uint8_t *label, const uint8_t *status,
                       const uint8_t *forget, const uint16_t *owner) {
  size_t n = m->P.size();
  for (size_t i = 0; i < n; ++i) {
    Particle &p = m->P[i];
    p.x = px[i];
    p.y = py[i];
    p.z = pz[i];
    p.weight = w[i];
    p.time_stamp = ts[i];
    p.track_id = track[i];
    p.label_id = label[i];
    p.status = status[i];
    p.forget_count = forget[i];
  }
  m->owner.clear();
  std::fill(m->owner_shadow.begin(), m->owner_shadow.end(), OWNER_NONE);
  if (owner) {
    for (size_t i = 0; i < n; ++i)
      if (owner[i] != OWNER_NONE) {
        m->owner[owner[i]].insert((uint32_t)i);
        m->owner_shadow[i] = owner[i];
      }
  }
}

void oracle_get_ck_kappa(oracle_map *m, float *out) { memcpy(out, m->ck_kappa.data(), m->ck_kappa.size() * 4); }
void oracle_get_bin_counts(oracle_map *m, uint32_t *out) { memcpy(out, m->bin_num.data(), m->bin_num.size() * 4); }
int64_t oracle_get_bins(oracle_map *m, uint32_t *out, int64_t cap) {
  int64_t n = 0;
  for (auto &b : m->bins)
    for (uint32_t idx : b) {
      if (n < cap) out[n] = idx;
      n++;
    }
  return n;
}
void oracle_get_extrinsic(oracle_map *m, float *out16) { memcpy(out16, m->extrinsic, 64); }
void oracle_get_pdf_table(oracle_map *m, float *out) { memcpy(out, m->pdf.data(), m->pdf.size() * 4); }

// utils/pointcloud_tools.h:88-310.  PINNED: K^-1 = (1/fx, -cx/fx, 1/fy, -cy/fy) in double (the reference inverts K
// with Eigen), K^-1*(j,i,1) = (ifx*j + icx, ify*i + icy, 1), R from Eigen's toRotationMatrix formula in double,
// camera-to-global = ((r0*x + r1*y) + r2*z) + t; fields of invalid points (uninitialised in the reference) are zero
// with sigma = zero-order term; sigma of a point the ZED2 box filter turns into Background (unset in the reference) is
// the noise model's.
void oracle_generate_cloud_ex(oracle_map *m, const float *depth_in, const uint8_t *static_in, const uint16_t *label_to_inst,
                              const int32_t *obj_track, const int32_t *obj_label, const uint8_t *obj_in, int32_t n_objects,
                              const double cam_pos[3], const double cam_q[4], int32_t consider_instance,
                              int32_t src_width, int32_t src_height, float rescale, int32_t sky_instance,
                              const double *object_bbox, oracle_labeled_point *out, float *depth_out) {
  const oracle_config &c = m->cfg;
  const int W = c.width, H = c.height;
  const size_t hw = (size_t)W * H;
  std::vector<float> depth_r;
  std::vector<uint8_t> static_r, obj_r;
  const float *depth = depth_in;
  const uint8_t *static_mask = static_in, *obj_masks = obj_in;
  if (src_width > 0) {  // BOOST mode (:98-102, 126-130, 172-176)
    const size_t shw = (size_t)src_width * src_height;
    depth_r.resize(hw);
    manual_resize(depth_in, src_width, src_height, depth_r.data(), W, H, rescale);
    depth = depth_r.data();
    if (static_in) {
      static_r.resize(hw);
      manual_resize(static_in, src_width, src_height, static_r.data(), W, H, rescale);
      static_mask = static_r.data();
    }
    obj_r.resize(hw * (size_t)std::max(n_objects, 1));
    for (int k = 0; k < n_objects; ++k) manual_resize(obj_in + shw * k, src_width, src_height, obj_r.data() + hw * k, W, H, rescale);
    obj_masks = obj_r.data();
  }
  if (depth_out) memcpy(depth_out, depth, hw * sizeof(float));
  const double w = cam_q[0], x = cam_q[1], y = cam_q[2], z = cam_q[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  const double R[9] = {1.0 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1.0 - (txx + tzz), tyz - twx,
                       txz - twy,         tyz + twx, 1.0 - (txx + tyy)};
  const double ifx = 1.0 / (double)c.fx, icx = -(double)c.cx / (double)c.fx;
  const double ify = 1.0 / (double)c.fy, icy = -(double)c.cy / (double)c.fy;
  const double dmin = (double)c.depth_min, dmax = (double)c.depth_max;
  const float sigma_invalid = m->prm.if_consider_depth_noise ? m->prm.depth_noise_zero_order : 0.1f;
  for (int i = 0; i < H; ++i)
    for (int j = 0; j < W; ++j) {
      const size_t p = (size_t)i * W + j;
      const float dv = depth[p];
      oracle_labeled_point o;
      const oracle_labeled_point invalid = {0.f, 0.f, 0.f, sigma_invalid, 0, 0, 0};
      if (std::isnan(dv) || (double)dv < dmin || (double)dv > dmax) {
        out[p] = invalid;
        continue;
      }
      uint32_t inst = 65535u;
      int label = 0;
      bool from_object = false;
      if (static_mask) {
        uint32_t pixel_label = (uint32_t)static_mask[p] + 1u;  // :137-138
        inst = label_to_inst[pixel_label > 255u ? 255u : pixel_label];
      }
      if (consider_instance)
        for (int k = 0; k < n_objects; ++k)
          if (obj_masks[(size_t)k * hw + p] > 0) {  // :196-207
            inst = (uint32_t)obj_track[k];
            label = obj_label[k];
            from_object = true;
          }
      if (sky_instance >= 0 && inst == (uint32_t)sky_instance) {  // :236-242
        out[p] = invalid;
        continue;
      }
      const double px = (ifx * (double)j + icx) * (double)dv;  // :243
      const double py = (ify * (double)i + icy) * (double)dv;
      const double pz = (double)dv;
      const double gx = ((R[0] * px + R[1] * py) + R[2] * pz) + cam_pos[0];  // :247
      const double gy = ((R[3] * px + R[4] * py) + R[5] * pz) + cam_pos[1];
      const double gz = ((R[6] * px + R[7] * py) + R[8] * pz) + cam_pos[2];
      const float sigma = m->prm.if_consider_depth_noise ? m->prm.depth_noise_zero_order + m->prm.depth_noise_first_order * dv : 0.1f;
      if (object_bbox && consider_instance && (int)inst < c.max_movable_track) {  // :254-272
        double b[6] = {0, 0, 0, 0, 0, 0};  // std::map default for a track id without an object
        for (int k = 0; k < n_objects; ++k)
          if ((uint32_t)obj_track[k] == inst)
            for (int q = 0; q < 6; ++q) b[q] = object_bbox[k * 6 + q];
        if (gx < b[0] || gx > b[1] || gy < b[2] || gy > b[3] || gz < b[4] || gz > b[5]) {
          o.x = (float)gx;
          o.y = (float)gy;
          o.z = (float)gz;
          o.sigma = sigma;
          o.track_id = 65535;
          o.label_id = 0;
          o.is_valid = 1;
          out[p] = o;
          continue;
        }
      }
      if ((int)inst > c.max_movable_track) {  // :277-283
        label = 0;
        if (static_mask)
          for (int l = 0; l < 256; ++l)
            if (label_to_inst[l] == inst) {
              label = l;
              break;
            }
      } else if (!from_object) {
        label = 0;
      }
      o.x = (float)gx;  // :298-300
      o.y = (float)gy;
      o.z = (float)gz;
      o.sigma = sigma;
      o.track_id = (uint16_t)inst;
      o.label_id = (uint8_t)label;
      o.is_valid = 1;
      out[p] = o;
    }
}

void oracle_generate_cloud(oracle_map *m, const float *depth, const uint8_t *static_mask, const uint16_t *label_to_inst,
                           const int32_t *obj_track, const int32_t *obj_label, const uint8_t *obj_masks, int32_t n_objects,
                           const double cam_pos[3], const double cam_q[4], int32_t consider_instance,
                           oracle_labeled_point *out) {
  oracle_generate_cloud_ex(m, depth, static_mask, label_to_inst, obj_track, obj_label, obj_masks, n_objects, cam_pos, cam_q,
                           consider_instance, 0, 0, 1.f, -1, nullptr, out, nullptr);
}

int32_t oracle_point_in_frustum(oracle_map *m, float x, float y, float z) { return m->isPointInFrustum(x, y, z) ? 1 : 0; }
uint32_t oracle_pos_to_voxel(oracle_map *m, float x, float y, float z) {
  uint32_t v, rx, ry, rz;
  m->globalPosToVoxel(x, y, z, v, rx, ry, rz);
  return v;
}
void oracle_voxel_to_pos(oracle_map *m, uint32_t voxel, float out[3]) { m->voxelToGlobalPos(voxel, out); }
float oracle_query_pdf(oracle_map *m, float x, float mu, float sigma) { return m->queryNormalPDF(x, mu, sigma); }
float oracle_forgetting_factor(oracle_map *m, int32_t c) { return m->getForgettingFactor(c); }
uint32_t oracle_add_particle(oracle_map *m, float x, float y, float z, uint8_t label, uint16_t track) {
  uint32_t v, p;
  m->addNewParticleWithSemantics(x, y, z, label, track, v, p);
  return p;
}
int32_t oracle_resample_voxel(oracle_map *m, uint32_t voxel) { return m->resampleParticlesInVoxel(voxel) ? 1 : 0; }
void oracle_set_global_time_stamp(oracle_map *m, uint32_t t) { m->global_time_stamp = t; }

}  // extern "C"
