/**
 * semantic_dsp_map.h — the reference's `class SemanticDSPMap` (public API verbatim, reference
 * include/semantic_dsp_map.h:21-251) on top of libsdm_hip (include/sdm.h).
 *
 * Drop-in for the particle-update path only: everything the reference does inside
 * subObjectLevelUpdate (semantic_dsp_map.h:576-955) runs on the MI355X behind sdm_update(); what stays on the
 * host is what the reference also does outside that function:
 *   - track-id reallocation (semantic_dsp_map.h:179-186),
 *   - the object layer (objectLevelUpdate, :306-566): the library's own restatement (sdm_objects.h, SdmBuiltinObjectLayer)
 *     is switched on at the first update() of a preset with consider_instance, like the reference's built-in one;
 *     another implementation (e.g. the reference's ObjectSet) plugs in through SdmObjectLayer / setObjectLayer(),
 *   - packing of the depth image and the MONO8 masks for sdm_update_raw_ex(), which runs generateLabeledPointCloud
 *     (utils/pointcloud_tools.h:88-310) on the device, including the BOOST-mode manualResize and
 *     the ZED2 sky / bounding-box filters (SdmGridPreset::Zed2Boost); the per-object boxes are computed here from the
 *     key points,
 *   - nothing of the output side: occupied / free voxels are compacted, coloured (semantic_dsp_map.h:1274-1351,
 *     including OpenCV's 8-bit RGB <-> HSV round trip and the "V x 0.7" dimming outside the view) and packed as
 *     pcl::PointXYZRGB on the device (sdm_get_occupied_rgb, row N2); update() copies the array into the cloud.
 *
 * Label tables: with SDM_HAVE_REFERENCE_TRACKING_TYPES defined (the node includes the reference's utils/data_base.h
 * first) the class reads g_label_id_map_default, g_label_color_map_default, g_instance_id_to_label_map_default and
 * g_max_movable_object_instance_id at the first update(), i.e. after ObjectInfoHandler::readObjectInfo has filled them
 * from the CSV (src/mapping.cpp:89-93) - an unchanged node needs no extra call.
 *
 * Compile-time grid/camera constants of settings/settings.h become SdmGridPreset.  Like the reference, the class picks
 * its preset from the macros SETTING (0 KITTI_360, 1 CODA, 2 VIRTUAL_KITTI2, 3 ZED2; settings.h:22) and BOOST_MODE
 * (:25-29) when the translation unit defines them - the node still includes the reference's settings/settings.h, or is
 * built with -DSETTING=n [-DBOOST_MODE=b] - and from what the reference ships (SETTING 3, hence BOOST_MODE 1) when it
 * does not: an unchanged node gets the grid it was built for.  setGridPreset() before the first update() overrides it.
 *
 * Needs Eigen3, OpenCV and PCL headers like the reference does.  They are not available in the build image of this
 * repository, where this file is only compile-checked against minimal stand-in headers (tests/mock_includes).
 */
#pragma once

#include <Eigen/Dense>
#include <opencv2/opencv.hpp>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <limits>
#include <map>
#include <set>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "sdm.h"
#include "sdm_objects.h"

// settings/settings.h:22-29: `#define SETTING 3` as shipped, BOOST_MODE 1 for SETTING 3 and 0 otherwise
#ifdef SETTING
#define SDM_SETTING (SETTING)
#else
#define SDM_SETTING 3
#endif
#ifdef BOOST_MODE
#define SDM_BOOST_MODE (BOOST_MODE)
#else
#define SDM_BOOST_MODE ((SDM_SETTING) == 3 ? 1 : 0)
#endif

#ifndef SDM_HAVE_REFERENCE_TRACKING_TYPES
/// utils/data_base.h:25-31
struct BBox2D {
  int x1, y1, x2, y2;
};
/// utils/tracking_result_handler.h:15-26 (the mask_kpts_msgs input format, docs/custom_files.md:16-45)
struct MaskKpts {
  int track_id;
  std::string label;
  std::vector<Eigen::Vector3d> kpts_current;
  std::vector<Eigen::Vector3d> kpts_previous;
  cv::Mat mask;
  BBox2D bbox;
};
#endif

/// settings/settings.h:32-141 as data
struct SdmGridPreset {
  int x_n, y_n, z_n, p_n;
  float voxel_size, fx, fy, cx, cy;
  int width, height;
  float depth_min, depth_max;
  int window_half;
  bool consider_instance;
  /// BOOST_MODE (settings.h:26, 137-143): inputs arrive at src_width x src_height and are reduced by `rescale`
  /// (manualResize); width/height and the intrinsics above are the reduced ones.  src_width = 0: no BOOST mode.
  int src_width = 0, src_height = 0;
  float rescale = 1.f;
  /// SETTING 3 (ZED2): sky pixels are dropped and every movable object's points are clipped to the box of its
  /// current key points +- 1 m (pointcloud_tools.h:174-196, 236-242, 254-272)
  bool zed2_filters = false;
  /// the reference's SETTING for the object layer (settings.h:22): 1 CODA, 2 VIRTUAL_KITTI2, 3 ZED2
  int object_mode = 2;
  static SdmGridPreset Kitti360() { return {8, 8, 8, 3, 0.15f, 552.554261f, 552.554261f, 682.049453f, 238.769549f, 1408, 376, 0.3f, 30.f, 5, false}; }
  static SdmGridPreset Coda() { return {8, 8, 7, 2, 0.15f, 569.8286f, 565.4818f, 439.2660f, 360.5810f, 960, 540, 0.3f, 10.f, 5, true}; }
  static SdmGridPreset VirtualKitti2() { return {8, 7, 8, 3, 0.2f, 725.0087f, 725.0087f, 620.5f, 187.f, 1242, 375, 0.3f, 30.f, 5, true}; }
  /// SETTING 3 with BOOST_MODE 1, the reference's ZED2 configuration (settings.h:24-27, 100-119, 137-143, semantic_dsp_map.h:964-970)
  static SdmGridPreset Zed2Boost() {
    SdmGridPreset p{7, 5, 7, 2, 0.15f, 0.5f * 527.8191528320312f, 0.5f * 527.8191528320312f, 0.5f * 633.9357299804688f,
                    0.5f * 366.3338623046875f, 640, 360, 0.3f, 15.f, 3, true};
    p.src_width = 1280;
    p.src_height = 720;
    p.rescale = 0.5f;
    p.zed2_filters = true;
    p.object_mode = 3;
    return p;
  }
  /// SETTING 3 without BOOST_MODE: the ZED2 grid at the sensor's 1280 x 720 (settings.h:100-119, 128-134), window 5
  static SdmGridPreset Zed2() {
    SdmGridPreset p{7, 5, 7, 2, 0.15f, 527.8191528320312f, 527.8191528320312f, 633.9357299804688f, 366.3338623046875f, 1280, 720, 0.3f, 15.f, 5, true};
    p.zed2_filters = true;
    p.object_mode = 3;
    return p;
  }
  /// BOOST_MODE 1 on top of a full-size preset (settings.h:135-143: image and intrinsics scaled by g_image_rescale = 0.5;
  /// semantic_dsp_map.h:964-970: neighbour half-size 3 instead of 5; pointcloud_tools.h:101, 129, 175: inputs resized)
  static SdmGridPreset Boosted(SdmGridPreset p) {
    const float r = 0.5f;
    p.src_width = p.width;
    p.src_height = p.height;
    p.rescale = r;
    p.fx = r * p.fx;
    p.fy = r * p.fy;
    p.cx = r * p.cx;
    p.cy = r * p.cy;
    p.width = (int)(r * (float)p.width);
    p.height = (int)(r * (float)p.height);
    p.window_half = 3;
    return p;
  }
  /// the preset the reference compiles in for `#define SETTING setting` / `#define BOOST_MODE boost` (settings.h:22-143);
  /// throws std::invalid_argument where the reference has `#error`
  static SdmGridPreset FromSetting(int setting, int boost) {
    SdmGridPreset p;
    switch (setting) {
      case 0: p = Kitti360(); p.object_mode = 2; break;
      case 1: p = Coda(); p.object_mode = 1; break;
      case 2: p = VirtualKitti2(); p.object_mode = 2; break;
      case 3: p = Zed2(); break;
      default: throw std::invalid_argument("SETTING must be 0 (KITTI_360), 1 (CODA), 2 (VIRTUAL_KITTI2) or 3 (ZED2)");
    }
    return boost ? Boosted(p) : p;
  }
};

/// What the particle layer needs from the object layer each frame (semantic_dsp_map.h:588-736).
struct SdmObjectLayer {
  virtual ~SdmObjectLayer() {}
  /// objectLevelUpdate (semantic_dsp_map.h:306-566)
  virtual void update(const std::vector<MaskKpts> &ins_seg_result, const Eigen::Vector3d &camera_position,
                      const Eigen::Quaterniond &camera_orientation, double time_stamp) = 0;
  /// objects to move this frame, in a deterministic order (ascending track id), and objects to wipe
  virtual void collect(uint32_t global_time_stamp, int max_obersevation_lost_time, std::vector<sdm_object_move> &moves,
                       std::vector<int32_t> &remove_tracks) = 0;
  virtual void clear() = 0;
  /// global_time_stamp after the increment at the head of update() (semantic_dsp_map.h:173), told before update()
  virtual void setGlobalTimeStamp(uint32_t) {}
  /// the track ids that own particles in the map right now (sdm_tracks_with_particles: the non-empty keys of the
  /// reference's obj_ptc_hash_map.indices_map, which its floating-object check walks, semantic_dsp_map.h:712-736), told
  /// before collect()
  virtual void setTracksWithParticles(const int32_t *, int32_t) {}
};

/// The library's own object layer (sdm_objects.h): objectLevelUpdate and the object loop of the
/// prediction step restated on plain doubles with a seeded RANSAC sampler.  "Floating" objects (tracks that own
/// particles but are not tracked, semantic_dsp_map.h:713-733): the map says which track ids own particles
/// (setTracksWithParticles; SemanticDSPMap::update asks sdm_tracks_with_particles every frame) and those go to
/// sdm_objects_collect as "present".  Driven without a map (nobody tells), it falls back on its own bookkeeping: every
/// movable track id that came with a mask may have particles born under it and counts as present until wiped.
class SdmBuiltinObjectLayer : public SdmObjectLayer {
 public:
  SdmBuiltinObjectLayer(const sdm_objects_config &cfg, const std::unordered_map<std::string, int> &label_ids)
      : h_(nullptr), label_ids_(label_ids), global_time_stamp_(0), max_movable_(cfg.max_movable_instance_id) {
    if (sdm_objects_create(&cfg, &h_) != SDM_OK) throw std::invalid_argument("sdm_objects_create: bad configuration");
  }
  ~SdmBuiltinObjectLayer() override { sdm_objects_destroy(h_); }
  SdmBuiltinObjectLayer(const SdmBuiltinObjectLayer &) = delete;
  SdmBuiltinObjectLayer &operator=(const SdmBuiltinObjectLayer &) = delete;

  void setGlobalTimeStamp(uint32_t t) override { global_time_stamp_ = t; }
  void update(const std::vector<MaskKpts> &ins_seg_result, const Eigen::Vector3d &camera_position,
              const Eigen::Quaterniond &camera_orientation, double time_stamp) override {
    std::vector<sdm_object_observation> obs(ins_seg_result.size());
    std::vector<std::vector<double>> cur(ins_seg_result.size()), prev(ins_seg_result.size());
    seen_this_frame_.clear();
    for (size_t i = 0; i < ins_seg_result.size(); ++i) {
      const MaskKpts &s = ins_seg_result[i];
      for (const Eigen::Vector3d &p : s.kpts_current) cur[i].insert(cur[i].end(), {p.x(), p.y(), p.z()});
      for (const Eigen::Vector3d &p : s.kpts_previous) prev[i].insert(prev[i].end(), {p.x(), p.y(), p.z()});
      auto it = label_ids_.find(s.label);
      obs[i].track_id = s.track_id;
      obs[i].label_id = it == label_ids_.end() ? -1 : it->second;
      obs[i].is_static = s.label == "static";
      obs[i].n_kpts = (int32_t)s.kpts_current.size();
      obs[i].kpts_current = cur[i].empty() ? nullptr : cur[i].data();
      obs[i].kpts_previous = prev[i].size() == cur[i].size() && !prev[i].empty() ? prev[i].data() : nullptr;
      if (!obs[i].is_static && s.track_id <= max_movable_) {
        maybe_present_.insert(s.track_id);
        seen_this_frame_.insert(s.track_id);
      }
    }
    const double pos[3] = {camera_position.x(), camera_position.y(), camera_position.z()};
    const double q[4] = {camera_orientation.w(), camera_orientation.x(), camera_orientation.y(), camera_orientation.z()};
    if (sdm_objects_update(h_, obs.empty() ? nullptr : obs.data(), (int32_t)obs.size(), pos, q, time_stamp, global_time_stamp_) != SDM_OK)
      std::cerr << "sdm_objects_update failed" << std::endl;
  }
  void collect(uint32_t global_time_stamp, int max_obersevation_lost_time, std::vector<sdm_object_move> &moves,
               std::vector<int32_t> &remove_tracks) override {
    int32_t n_tracked = 0;
    sdm_objects_count(h_, &n_tracked);
    const std::vector<int32_t> present = have_owner_tracks_ ? owner_tracks_ : std::vector<int32_t>(maybe_present_.begin(), maybe_present_.end());
    have_owner_tracks_ = false;  // (told anew before every collect)
    const int32_t cap = n_tracked + (int32_t)present.size();
    moves.resize((size_t)cap);
    remove_tracks.resize((size_t)cap);
    int32_t n_moves = 0, n_remove = 0;
    if (sdm_objects_collect(h_, global_time_stamp, max_obersevation_lost_time, present.empty() ? nullptr : present.data(),
                            (int32_t)present.size(), moves.data(), cap, &n_moves, remove_tracks.data(), cap, &n_remove) != SDM_OK) {
      std::cerr << "sdm_objects_collect failed" << std::endl;
      n_moves = n_remove = 0;
    }
    moves.resize((size_t)n_moves);
    remove_tracks.resize((size_t)n_remove);
    // wiped: nothing of it is left in the map - unless it also came with a mask THIS frame: sdm_update applies the
    // removals before this frame's births, so particles are born under the id again right away and it has to be
    // offered as "present" next frame like the reference's obj_ptc_hash_map would (semantic_dsp_map.h:713-736)
    for (int32_t id : remove_tracks)
      if (!seen_this_frame_.count(id)) maybe_present_.erase(id);
  }
  void setTracksWithParticles(const int32_t *ids, int32_t n) override {
    owner_tracks_.assign(ids, ids + (n > 0 ? n : 0));
    have_owner_tracks_ = true;
  }
  /// the frame these removals belonged to was not applied: offer them again
  void requeue(const std::vector<int32_t> &remove_tracks) {
    for (int32_t id : remove_tracks) maybe_present_.insert(id);
  }
  void setBayes(double distance_threshold, double probability_threshold, double increment, double decrement) {
    sdm_objects_set_bayes(h_, distance_threshold, probability_threshold, increment, decrement);
  }
  void clear() override {
    sdm_objects_clear(h_);
    maybe_present_.clear();
  }
  sdm_objects *handle() { return h_; }

 private:
  sdm_objects *h_;
  std::unordered_map<std::string, int> label_ids_;
  uint32_t global_time_stamp_;
  int max_movable_;
  std::set<int32_t> maybe_present_, seen_this_frame_;
  std::vector<int32_t> owner_tracks_;
  bool have_owner_tracks_ = false;
};

/// Where the last update() call spent its wall-clock time (milliseconds); see SemanticDSPMap::lastUpdateTimes().
struct SdmUpdateTimes {
  double objects = 0;  ///< object layer: update, owner-track query (waits for the previous frame), collect
  double pack = 0;     ///< depth image and masks into the staging buffers
  double frame = 0;    ///< sdm_update_raw_ex: uploads, labelled cloud, the frame's launches (returns when the uploads are done)
  double emit = 0;     ///< waiting for the frame + occupied / free-space clouds back to the host
  double total = 0;
};

/// Page-locked host staging for the images handed to sdm_update_raw_ex (sdm_host_alloc): an upload from it is one DMA
/// transfer at PCIe speed; from a std::vector the runtime first copies through bounce buffers of its own.
template <typename T>
class SdmPinnedBuffer {
 public:
  SdmPinnedBuffer() = default;
  SdmPinnedBuffer(const SdmPinnedBuffer &) = delete;
  SdmPinnedBuffer &operator=(const SdmPinnedBuffer &) = delete;
  ~SdmPinnedBuffer() { sdm_host_free(p_); }
  /// n elements, contents undefined after a growth
  void resize(size_t n) {
    if (n > cap_) {
      sdm_host_free(p_);
      p_ = nullptr;
      cap_ = 0;
      void *q = nullptr;
      if (sdm_host_alloc(n * sizeof(T), &q) != SDM_OK) throw std::bad_alloc();
      p_ = static_cast<T *>(q);
      cap_ = n;
    }
    n_ = n;
  }
  size_t size() const { return n_; }
  T *data() { return p_; }
  T &operator[](size_t i) { return p_[i]; }

 private:
  T *p_ = nullptr;
  size_t n_ = 0, cap_ = 0;
};

class SemanticDSPMap {
 public:
  /// semantic_dsp_map.h:25-67
  SemanticDSPMap()
      : map_(nullptr), object_layer_(nullptr), preset_(SdmGridPreset::FromSetting(SDM_SETTING, SDM_BOOST_MODE)), device_(0),
        global_time_stamp_(0), visualize_with_zero_center_(false), if_out_evaluation_format_(false) {
    params_.detection_probability = 0.95f;
    params_.noise_number = 0.1f;
    params_.nb_ptc_num_per_point = 3;
    params_.occupancy_threshold = 0.2f;
    params_.max_obersevation_lost_time = 5;
    params_.forgetting_rate = 1.0f;
    params_.max_forget_count = 5;
    params_.match_score_threshold = 0.3f;
    params_.id_transition_probability = 0.1f;
    params_.if_consider_depth_noise = 0;
    params_.if_use_independent_filter = 0;
    params_.depth_noise_first_order = 0.0f;
    params_.depth_noise_zero_order = 0.1f;
    for (int i = 0; i < 256; ++i) color_map_int_256_.push_back(i);
    // semantic_dsp_map.h:45-48 shuffles with an unseeded generator; a fixed LCG permutation keeps runs reproducible
    uint32_t s = 12345u;
    for (int i = 255; i > 0; --i) {
      s = s * 1664525u + 1013904223u;
      std::swap(color_map_int_256_[i], color_map_int_256_[(s >> 8) % (i + 1)]);
    }
    for (int i = 0; i < 256; ++i) {  // jet map, semantic_dsp_map.h:50-63
      Eigen::Vector3i c;
      if (i < 64) c << 0, 0, i * 4;
      else if (i < 128) c << 0, (i - 64) * 4, 255;
      else if (i < 192) c << (i - 128) * 4, 255, 255 - (i - 128) * 4;
      else c << 255, 255 - (i - 192) * 4, 0;
      color_map_jet_256_.push_back(c);
    }
    defaultLabelTables();
  }
  ~SemanticDSPMap() {
    if (map_) sdm_destroy(map_);
  }
  SemanticDSPMap(const SemanticDSPMap &) = delete;
  SemanticDSPMap &operator=(const SemanticDSPMap &) = delete;

  // ---- additions (the reference fixes these at compile time / inside the class) ----
  void setGridPreset(const SdmGridPreset &p) { preset_ = p; }
  const SdmGridPreset &gridPreset() const { return preset_; }
  void setDevice(int hip_device) { device_ = hip_device; }
  /// The Gaussian table the prediction step and the noisy births draw from (basic_algorithms.h:394-402: 1,000,000 draws
  /// with standard deviation prediction_stddev_ = 0.05, seeded from std::random_device in the reference).  By default the
  /// map fills its own with rocRAND (fixed seed); a caller that needs a particular table - a reproducible run, a parity
  /// test - hands it over before the first update().
  void setNoiseTable(const float *table, size_t n) { noise_table_.assign(table, table + n); }
  void setObjectLayer(SdmObjectLayer *layer) { object_layer_ = layer; }
  /// Use the library's object layer (sdm_objects.h).  mode = the reference's SETTING (settings.h:22); call after
  /// setGridPreset / setLabelTables / setBeyesianMovementParameters.
  void useBuiltinObjectLayer(int mode, uint64_t seed = 20250217ull) {
    sdm_objects_config c;
    c.mode = mode;
    c.max_movable_instance_id = max_movable_track_;
    c.movement_distance_threshold = beyesian_[0];
    c.movement_probability_threshold = beyesian_[1];
    c.movement_increment = beyesian_[2];
    c.movement_decrement = beyesian_[3];
    const int n_biggest = std::max(std::max(preset_.x_n, preset_.y_n), preset_.z_n);
    c.map_half_size_scaled = (double)preset_.voxel_size * (double)(1 << (n_biggest - 1)) * 1.2;  // semantic_dsp_map.h:356
    c.fx = preset_.fx;
    c.fy = preset_.fy;
    c.cx = preset_.cx;
    c.cy = preset_.cy;
    c.image_width = preset_.width;
    c.image_height = preset_.height;
    c.seed = seed;
    builtin_layer_.reset(new SdmBuiltinObjectLayer(c, label_id_));
    object_layer_ = builtin_layer_.get();
  }
  /// label tables of utils/object_info_handler.h:28-91 (CSV): label name -> label id, static label -> instance id
  void setLabelTables(const std::map<std::string, int> &label_ids, const std::map<std::string, int> &static_instance_ids) {
    label_id_.clear();
    static_instance_to_label_.clear();
    for (auto &kv : label_ids) label_id_[kv.first] = kv.second;
    int min_static = 65536;
    for (auto &kv : static_instance_ids) {
      static_instance_to_label_[kv.second] = label_id_[kv.first];
      min_static = std::min(min_static, kv.second);
    }
    const int max_movable = min_static - 1;  // object_info_handler.h:84
    if (map_ && max_movable != max_movable_track_)
      std::cerr << "setLabelTables: g_max_movable_object_instance_id changed after the map was created (" << max_movable_track_
                << " -> " << max_movable << "): the device keeps the value of the first update()" << std::endl;
    max_movable_track_ = max_movable;
    rebuildLabelToInstance();
    pushColours();  // the emitted colours are baked on the device: tables set after the first update() take effect too
  }
  /// g_label_color_map_default (utils/data_base.h:216-232): BGR colour of a label id
  void setLabelColours(const std::map<int, cv::Vec3b> &label_colours) {
    label_color_.clear();
    for (const auto &kv : label_colours) label_color_[kv.first] = kv.second;
    pushColours();
  }

  // ---- the reference's public interface ----
  /// semantic_dsp_map.h:74-81
  void clear() {
    if (map_) check(sdm_clear(map_), "sdm_clear");
    if (object_layer_) object_layer_->clear();
    global_time_stamp_ = 0;
  }
  /// semantic_dsp_map.h:85-88 (template matching is dead code in the shipped node)
  void setTemplatePath(std::string) {}
  /// semantic_dsp_map.h:101-115
  void setMapParameters(float detection_probability, float noise_number, int nb_ptc_num_per_point, float occupancy_threshold,
                        int max_obersevation_lost_time, float forgetting_rate = 1.f, int max_forget_count = 5,
                        float match_score_threshold = 0.5, float id_transition_probability = 0.1) {
    params_.detection_probability = detection_probability;
    params_.noise_number = noise_number;
    params_.nb_ptc_num_per_point = nb_ptc_num_per_point;
    params_.occupancy_threshold = occupancy_threshold;
    params_.max_obersevation_lost_time = max_obersevation_lost_time;
    params_.forgetting_rate = forgetting_rate;
    params_.max_forget_count = max_forget_count;
    params_.match_score_threshold = match_score_threshold;
    params_.id_transition_probability = id_transition_probability;
    std::cout << "max_obersevation_lost_time_ = " << max_obersevation_lost_time << std::endl;
    std::cout << "id_transition_probability_ = " << id_transition_probability << std::endl;
    pushParams();
  }
  /// semantic_dsp_map.h:121-125
  void setMapOptions(bool if_consider_depth_noise, bool if_use_independent_filter) {
    params_.if_consider_depth_noise = if_consider_depth_noise ? 1 : 0;
    params_.if_use_independent_filter = if_use_independent_filter ? 1 : 0;
    pushParams();
  }
  /// semantic_dsp_map.h:130-134
  void setVisualizeOptions(bool visualize_with_zero_center, bool if_out_evaluation_format) {
    visualize_with_zero_center_ = visualize_with_zero_center;
    if_out_evaluation_format_ = if_out_evaluation_format;
    pushColours();
  }
  /// semantic_dsp_map.h:142-148 (consumed by the object layer)
  void setBeyesianMovementParameters(double distance_threshold, double probability_threshold, double increment, double decrement) {
    beyesian_[0] = distance_threshold;
    beyesian_[1] = probability_threshold;
    beyesian_[2] = increment;
    beyesian_[3] = decrement;
    if (builtin_layer_) builtin_layer_->setBayes(distance_threshold, probability_threshold, increment, decrement);
  }
  const double *beyesianMovementParameters() const { return beyesian_; }
  /// semantic_dsp_map.h:153-156
  void setOccupancyThreshold(float threshold) {
    params_.occupancy_threshold = threshold;
    pushParams();
  }
  /// semantic_dsp_map.h:161-166
  void setDepthNoiseModelParameters(float first_order, float zero_order) {
    params_.depth_noise_first_order = first_order;
    params_.depth_noise_zero_order = zero_order;
    std::cout << "Noise model is " << first_order << " * distance + " << zero_order << std::endl;
    pushParams();
  }

  /// semantic_dsp_map.h:170-251.  Like the reference: no return code, diagnostics on stderr, output clouds are
  /// appended to and neither cleared nor given width/height.
  void update(cv::Mat &depth_value_mat, std::vector<MaskKpts> &ins_seg_result, Eigen::Vector3d &camera_position,
              Eigen::Quaterniond &camera_orientation, pcl::PointCloud<pcl::PointXYZRGB>::Ptr &occupied_point_cloud,
              pcl::PointCloud<pcl::PointXYZRGB>::Ptr &freespace_point_cloud, bool if_get_freespace = false,
              double time_stamp_double = 0.0) {
    using clock = std::chrono::steady_clock;
    const auto ms_since = [](clock::time_point t) { return std::chrono::duration<double, std::milli>(clock::now() - t).count(); };
    const clock::time_point t_begin = clock::now();
    times_ = SdmUpdateTimes();
    ensureMap();
    if (preset_.consider_instance && !object_layer_) useBuiltinObjectLayer(preset_.object_mode);  // like the reference's built-in one
    global_time_stamp_ += 1;
    for (size_t i = 0; i < ins_seg_result.size(); ++i) {  // :179-186
      if (ins_seg_result[i].label != "static" && ins_seg_result[i].track_id > max_movable_track_) {
        std::cout << "Reach the maximum movable object instance id. ID reallocated." << std::endl;
        ins_seg_result[i].track_id = ins_seg_result[i].track_id % max_movable_track_;
      }
    }
    std::vector<sdm_object_move> moves;
    std::vector<int32_t> removals;
    if (preset_.consider_instance && object_layer_) {  // :189-191
      object_layer_->setGlobalTimeStamp(global_time_stamp_);
      object_layer_->update(ins_seg_result, camera_position, camera_orientation, time_stamp_double);
      {  // the keys of the owner sets as the previous frame left them (semantic_dsp_map.h:712-736 walks them here)
        int32_t n_own = 0;
        if (owner_tracks_.size() < 64) owner_tracks_.resize(64);
        if (check(sdm_tracks_with_particles(map_, owner_tracks_.data(), (int32_t)owner_tracks_.size(), &n_own), "sdm_tracks_with_particles")) {
          if ((size_t)n_own > owner_tracks_.size()) {
            owner_tracks_.resize((size_t)n_own);
            check(sdm_tracks_with_particles(map_, owner_tracks_.data(), (int32_t)owner_tracks_.size(), &n_own), "sdm_tracks_with_particles");
          }
          object_layer_->setTracksWithParticles(owner_tracks_.data(), n_own);
        }
      }
      object_layer_->collect(global_time_stamp_, params_.max_obersevation_lost_time, moves, removals);
    }
    // (the lists go to the library whole: it works lists longer than SDM_MAX_MOVES / SDM_MAX_REMOVALS off in batches inside
    // the frame, with the reference's order - semantic_dsp_map.h:588-736 loops over whatever the object layer hands it)
    removals.insert(removals.begin(), pending_removals_.begin(), pending_removals_.end());
    pending_removals_.clear();
    times_.objects = ms_since(t_begin);
    clock::time_point t_phase = clock::now();
    if (packRawInputs(depth_value_mat, ins_seg_result) != 0) {
      frameLost(removals);
      return;
    }
    times_.pack = ms_since(t_phase);
    t_phase = clock::now();

    // pose in double for the back-projection (pointcloud_tools.h:243-247); the library casts it to float for the
    // map update like the reference does (semantic_dsp_map.h:584, 745)
    const double cam_pos_d[3] = {camera_position.x(), camera_position.y(), camera_position.z()};
    const double cam_q_d[4] = {camera_orientation.w(), camera_orientation.x(), camera_orientation.y(), camera_orientation.z()};
    sdm_raw_options opt;
    opt.src_width = preset_.src_width;
    opt.src_height = preset_.src_height;
    opt.rescale = preset_.rescale;
    opt.sky_instance = -1;
    if (preset_.zed2_filters && label_id_.count("Sky")) {  // g_label_to_instance_id_map_default["Sky"]
      const int sky_label = label_id_["Sky"];
      if (sky_label >= 0 && sky_label < 256) opt.sky_instance = label_to_inst_[sky_label];
    }
    opt.object_bbox = preset_.zed2_filters && !boxes_.empty() ? boxes_.data() : nullptr;
    if (!check(sdm_update_raw_ex(map_, depth_.data(), have_static_ ? static_mask_.data() : nullptr, label_to_inst_,
                                 objects_.empty() ? nullptr : objects_.data(), (int32_t)objects_.size(), cam_pos_d, cam_q_d,
                                 moves.empty() ? nullptr : moves.data(), (int32_t)moves.size(),
                                 removals.empty() ? nullptr : removals.data(), (int32_t)removals.size(),
                                 preset_.consider_instance ? 0u : SDM_NO_INSTANCES, SDM_STAGE_ALL, &opt),
               "sdm_update_raw_ex")) {
      frameLost(removals);
      return;
    }
    times_.frame = ms_since(t_phase);
    t_phase = clock::now();
    emit(occupied_point_cloud, false);
    if (if_get_freespace) emit(freespace_point_cloud, true);
    // the getters above waited for the frame: surface what the device flagged (work-list overflows are recorded in
    // counters that the next frame resets)
    check(sdm_synchronize(map_), "sdm_update (device status)");
    times_.emit = ms_since(t_phase);
    times_.total = ms_since(t_begin);
  }

  sdm_map *handle() { return map_; }
  /// not in the reference: the phases of the last update() call
  const SdmUpdateTimes &lastUpdateTimes() const { return times_; }

 private:
  sdm_map *map_;
  SdmObjectLayer *object_layer_;
  std::unique_ptr<SdmBuiltinObjectLayer> builtin_layer_;
  SdmGridPreset preset_;
  sdm_params params_;
  int device_;
  uint32_t global_time_stamp_;
  bool visualize_with_zero_center_, if_out_evaluation_format_;
  double beyesian_[4] = {0.1, 0.69, 0.1, 0.15};
  int max_movable_track_ = 65523;  // utils/data_base.h:196
  std::vector<int> color_map_int_256_;
  std::vector<Eigen::Vector3i> color_map_jet_256_;
  std::unordered_map<std::string, int> label_id_;          // g_label_id_map_default
  std::unordered_map<int, int> static_instance_to_label_;   // g_instance_id_to_label_map_default -> label id
  std::unordered_map<int, cv::Vec3b> label_color_;          // g_label_color_map_default (BGR)
  SdmPinnedBuffer<float> depth_;
  SdmPinnedBuffer<uint8_t> static_mask_, object_masks_;
  std::vector<sdm_instance_mask> objects_;
  std::vector<double> boxes_;  // ZED2: per object min x, max x, min y, max y, min z, max z
  uint16_t label_to_inst_[256];  // g_label_to_instance_id_map_default as a table (65535 = Background's instance)
  bool have_static_ = false;
  std::vector<sdm_point_xyzrgb> points_;
  std::vector<int32_t> pending_removals_;
  std::vector<int32_t> owner_tracks_;
  std::vector<float> noise_table_;
  SdmUpdateTimes times_;
  bool tables_from_reference_ = false;

  /// a frame that did not reach the map: its removals are offered again
  void frameLost(const std::vector<int32_t> &removals) {
    if (builtin_layer_) builtin_layer_->requeue(removals);
    else pending_removals_.insert(pending_removals_.end(), removals.begin(), removals.end());
  }
  /// colour tables + output format -> device (row N2)
  void pushColours() {
    if (!map_) return;
    sdm_colour_config c;
    std::memset(&c, 0, sizeof(c));
    for (const auto &kv : label_color_)
      if (kv.first >= 0 && kv.first < 256)
        for (int k = 0; k < 3; ++k) c.label_bgr[kv.first][k] = kv.second[k];
    for (int i = 0; i < 256; ++i) c.perm[i] = (uint8_t)color_map_int_256_[i];
    c.background_label = label_id_.count("Background") ? label_id_["Background"] : 0;
    c.colour_by_label = preset_.consider_instance ? 0 : 1;  // SETTING == 0 (semantic_dsp_map.h:1296-1302)
    c.jet_axis = preset_.zed2_filters ? 1 : 0;              // SETTING == 3 colours by y (:1281-1286)
    c.evaluation_format = if_out_evaluation_format_ ? 1 : 0;
    check(sdm_set_colours(map_, &c), "sdm_set_colours");
  }

  bool check(sdm_status s, const char *what) {
    if (s == SDM_OK) return true;
    std::cerr << what << " failed: " << sdm_last_error() << std::endl;  // the reference reports on stdout/stderr and carries on
    return false;
  }
  void pushParams() {
    if (map_) check(sdm_set_params(map_, &params_), "sdm_set_params");
  }
  void ensureMap() {
    if (map_) return;
#ifdef SDM_HAVE_REFERENCE_TRACKING_TYPES
    // the reference's label globals (utils/data_base.h:108-232), as ObjectInfoHandler::readObjectInfo left them
    // (utils/object_info_handler.h:76-87, called at src/mapping.cpp:89-93 before the first frame)
    if (!tables_from_reference_) {
      label_id_.clear();
      static_instance_to_label_.clear();
      label_color_.clear();
      for (const auto &kv : g_label_id_map_default) label_id_[kv.first] = kv.second;
      for (const auto &kv : g_instance_id_to_label_map_default)
        if (label_id_.count(kv.second)) static_instance_to_label_[kv.first] = label_id_[kv.second];
      for (const auto &kv : g_label_color_map_default) label_color_[kv.first] = kv.second;
      max_movable_track_ = g_max_movable_object_instance_id;
      rebuildLabelToInstance();
      tables_from_reference_ = true;
    }
#endif
    sdm_config c{};
    c.x_n = preset_.x_n;
    c.y_n = preset_.y_n;
    c.z_n = preset_.z_n;
    c.p_n = preset_.p_n;
    c.voxel_size = preset_.voxel_size;
    c.fx = preset_.fx;
    c.fy = preset_.fy;
    c.cx = preset_.cx;
    c.cy = preset_.cy;
    c.width = preset_.width;
    c.height = preset_.height;
    c.depth_min = preset_.depth_min;
    c.depth_max = preset_.depth_max;
    c.window_half = preset_.window_half;
    c.max_movable_track = max_movable_track_;
    c.device = device_;
    c.shard_rank = 0;
    c.shard_count = 1;
    if (sdm_create(&c, &map_) != SDM_OK) throw std::runtime_error(std::string("sdm_create: ") + sdm_last_error());
    // prediction_stddev_ = 0.05 table of 1,000,000 draws (semantic_dsp_map.h:40,66; basic_algorithms.h:394-402)
    if (noise_table_.empty()) check(sdm_generate_noise_table(map_, 20250217ull, 1000000, 0.05f), "sdm_generate_noise_table");
    else check(sdm_upload_noise_table(map_, noise_table_.data(), (int32_t)noise_table_.size()), "sdm_upload_noise_table");
    pushParams();
    pushColours();
    depth_.resize((size_t)c.width * c.height);
    static_mask_.resize((size_t)c.width * c.height);
  }
  void rebuildLabelToInstance() {
    for (int l = 0; l < 256; ++l) label_to_inst_[l] = 65535;
    for (const auto &kv : static_instance_to_label_)
      if (kv.second >= 0 && kv.second < 256) label_to_inst_[kv.second] = (uint16_t)kv.first;
  }

  void defaultLabelTables() {  // utils/data_base.h:108-232
    const char *names[] = {"Background", "Terrain", "Sky", "Tree", "Vegetation", "Building", "Road", "GuardRail",
                           "TrafficSign", "TrafficLight", "Pole", "Misc", "Truck", "Car", "Person"};
    const int ids[] = {0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15};
    for (int i = 0; i < 15; ++i) label_id_[names[i]] = ids[i];
    for (int i = 0; i < 12; ++i) static_instance_to_label_[65535 - i] = ids[i];
    const int bgr[15][4] = {{0, 0, 0, 0},      {2, 200, 0, 210},  {3, 255, 200, 90}, {4, 0, 199, 0},    {5, 0, 240, 90},
                            {6, 140, 140, 140}, {7, 100, 60, 100}, {8, 255, 100, 250}, {9, 0, 255, 255},  {10, 0, 200, 200},
                            {11, 0, 130, 255},  {12, 80, 80, 80},  {13, 60, 60, 160},  {14, 80, 127, 255}, {15, 139, 139, 0}};
    for (auto &c : bgr) label_color_[c[0]] = cv::Vec3b((uchar)c[1], (uchar)c[2], (uchar)c[3]);
    max_movable_track_ = 65523;
    rebuildLabelToInstance();
  }

  /// Host half of PointCloudTools::generateLabeledPointCloud (utils/pointcloud_tools.h:88-213): validate the depth
  /// image and lay the MONO8 masks out as dense H x W buffers; the per-pixel work (:218-304) is the device kernel
  /// behind sdm_update_raw.
  /// one MONO8 mask into a W x H staging image; what the mask does not cover is set to `fill`
  static void copyMask(const cv::Mat &mask, uint8_t *dst, int W, int H, int fill) {
    const int rows = std::max(std::min(mask.rows, H), 0), cols = std::max(std::min(mask.cols, W), 0);
    for (int j = 0; j < rows; ++j) {
      std::memcpy(dst + (size_t)j * W, mask.ptr<uchar>(j), (size_t)cols);
      if (cols < W) std::memset(dst + (size_t)j * W + cols, fill, (size_t)(W - cols));
    }
    if (rows < H) std::memset(dst + (size_t)rows * W, fill, (size_t)(H - rows) * W);
  }
  int packRawInputs(const cv::Mat &depth, const std::vector<MaskKpts> &seg) {
    if (depth.empty()) {
      std::cerr << "Error: depth image is empty." << std::endl;
      return -1;
    }
    // BOOST mode: the images keep the sensor's size here, the library reduces them (manualResize) on the device
    const int W = preset_.src_width > 0 ? preset_.src_width : preset_.width;
    const int H = preset_.src_width > 0 ? preset_.src_height : preset_.height;
    if (depth.cols != W || depth.rows != H) {
      std::cerr << "Error: depth image size does not match the grid preset." << std::endl;
      return -1;
    }
    const size_t hw = (size_t)W * H;
    if (depth_.size() != hw) {
      depth_.resize(hw);
      static_mask_.resize(hw);
    }
    for (int i = 0; i < H; ++i) std::memcpy(&depth_[(size_t)i * W], depth.ptr<float>(i), (size_t)W * sizeof(float));  // (row by row: a cv::Mat need not be continuous)
    have_static_ = false;
    for (const auto &s : seg) {  // :121-143, the first "static" entry
      if (s.label != "static") continue;
      // uncovered pixels: no label -> Background's instance
      copyMask(s.mask, static_mask_.data(), W, H, 255);
      have_static_ = true;
      break;
    }
    objects_.clear();
    boxes_.clear();
    size_t n_obj = 0;
    if (preset_.consider_instance)
      for (const auto &s : seg) n_obj += s.label != "static";
    object_masks_.resize(n_obj * hw);
    if (preset_.consider_instance) {  // :163-213, later masks override earlier ones
      size_t k_obj = 0;
      for (const auto &s : seg) {
        if (s.label == "static") continue;
        uint8_t *dst = object_masks_.data() + k_obj * hw;
        copyMask(s.mask, dst, W, H, 0);
        auto it = label_id_.find(s.label);
        sdm_instance_mask o;
        o.track_id = s.track_id;
        o.label_id = it == label_id_.end() ? 0 : it->second;
        o.mask = dst;
        objects_.push_back(o);
        if (preset_.zed2_filters) {  // pointcloud_tools.h:174-196 (max starts at the smallest positive double there)
          double lo[3] = {std::numeric_limits<double>::max(), std::numeric_limits<double>::max(), std::numeric_limits<double>::max()};
          double hi[3] = {std::numeric_limits<double>::min(), std::numeric_limits<double>::min(), std::numeric_limits<double>::min()};
          for (const auto &kpt : s.kpts_current)
            for (int a = 0; a < 3; ++a) {
              if (kpt(a) < lo[a]) lo[a] = kpt(a);
              if (kpt(a) > hi[a]) hi[a] = kpt(a);
            }
          const double margin = 1.0;
          for (int a = 0; a < 3; ++a) {
            boxes_.push_back(lo[a] - margin);
            boxes_.push_back(hi[a] + margin);
          }
        }
        ++k_obj;
      }
    }
    return 0;
  }

  /// getOccupancyResult output side (semantic_dsp_map.h:1258-1376): compacted, coloured and packed as pcl::PointXYZRGB
  /// on the device; appended to the cloud as they come (the reference appends too).
  void emit(pcl::PointCloud<pcl::PointXYZRGB>::Ptr &out, bool free_space) {
    static_assert(sizeof(pcl::PointXYZRGB) == sizeof(sdm_point_xyzrgb), "sdm_point_xyzrgb is pcl::PointXYZRGB's layout");
    size_t n = 0;
    const size_t cap = (size_t)1 << (preset_.x_n + preset_.y_n + preset_.z_n);
    if (points_.size() < 1024) points_.resize(1024);
    auto get = free_space ? sdm_get_freespace_rgb : sdm_get_occupied_rgb;
    const int32_t flags = visualize_with_zero_center_ ? SDM_POINTS_ZERO_CENTER : 0;
    if (!check(get(map_, points_.data(), points_.size(), &n, flags), "sdm_get_occupied_rgb")) return;
    if (n > points_.size()) {  // (with headroom: a map that grows a little every frame would come here every frame)
      points_.resize(std::min(n + n / 2 + 1024, cap));
      if (!check(get(map_, points_.data(), points_.size(), &n, flags), "sdm_get_occupied_rgb")) return;
    }
    n = std::min(n, points_.size());
    const size_t first_new = out->points.size();
    out->points.resize(first_new + n);
    if (n) std::memcpy(static_cast<void *>(&out->points[first_new]), points_.data(), n * sizeof(sdm_point_xyzrgb));
  }
};
