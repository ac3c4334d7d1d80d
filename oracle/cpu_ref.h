/*
 * oracle/cpu_ref.h — C ABI of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference ships no tests and no golden vectors for
 * this path and cannot be compiled in this image (needs
 * Eigen, OpenCV, PCL, ROS).  The oracle is a literal single-threaded
 * restatement of the reference's loops; it is pinned only by the analytic
 * known-answer tests in tests/test_oracle_kat.py that are derived from the
 * reference code (file:line cited on each function in cpu_ref.cpp).
 */
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_map oracle_map;

/* Compile-time constants of the reference (include/settings/settings.h:18-150)
 * turned into run-time configuration. */
typedef struct {
  int32_t x_n, y_n, z_n, p_n; /* log2 voxels per axis, log2 slots per voxel   */
  float voxel_size;           /* C_VOXEL_SIZE                                 */
  float fx, fy, cx, cy;       /* g_camera_*                                   */
  int32_t width, height;      /* g_image_width/height                         */
  float depth_min, depth_max; /* g_depth_range_min/max                        */
  int32_t window_half;        /* 5, or 3 in BOOST mode (semantic_dsp_map.h:964-970) */
  int32_t max_movable_track;  /* g_max_movable_object_instance_id (data_base.h:196) */
  int32_t bin_order;          /* 0: BFS push order (literal reference); 1: ascending particle index (canonical) */
  int32_t ck_slabs;           /* 1; >1 emulates the Z-slab partial-sum order of the multi-GPU path */
} oracle_config;

/* setMapParameters / setMapOptions / setDepthNoiseModelParameters
 * (semantic_dsp_map.h:101-166) */
typedef struct {
  float detection_probability;
  float noise_number;
  int32_t nb_ptc_num_per_point;
  float occupancy_threshold;
  int32_t max_obersevation_lost_time;
  float forgetting_rate;
  int32_t max_forget_count;
  float match_score_threshold;
  float id_transition_probability;
  int32_t if_consider_depth_noise;
  int32_t if_use_independent_filter;
  float depth_noise_first_order;
  float depth_noise_zero_order;
} oracle_params;

/* LabeledPoint (utils/data_base.h:78-92), 20 bytes */
typedef struct {
  float x, y, z;
  float sigma;
  uint16_t track_id;
  uint8_t label_id;
  uint8_t is_valid;
} oracle_labeled_point;

typedef struct {
  int32_t track_id;
  float T[16]; /* row-major 4x4, already cast to float (semantic_dsp_map.h:673-674) */
} oracle_object_move;

typedef struct {
  float wsum;     /* weight_sum of calculateWeightAndSemanticsInVoxel (-1 = unknown) */
  uint16_t track;
  uint8_t label;
  int8_t occ;     /* -1 unknown, 0 free, 1 occupied, 2 guessed occupied */
} oracle_voxel_result;

/* stage ids for stop_after (0 = run everything) */
enum {
  ORACLE_STAGE_ALL = 0,
  ORACLE_STAGE_EGO = 1,
  ORACLE_STAGE_MOVE = 2,
  ORACLE_STAGE_REMOVE = 3,
  ORACLE_STAGE_VISIBILITY = 4,
  ORACLE_STAGE_WEIGHT = 5,
  ORACLE_STAGE_BIRTH = 6,
  ORACLE_STAGE_OCCUPANCY = 7
};

typedef struct {
  uint32_t global_time_stamp;
  int32_t moved_steps[3];
  int32_t eq_steps[3];
  float map_center[3];
  float last_pos[3];
  int32_t birth_cursor; /* SemanticDSPMap::gaussian_random_ cursor            */
  int32_t move_cursor;  /* RingBufferOperations::gaussian_random_calculator_  */
} oracle_ring_state;

typedef struct {
  int64_t live_particles;      /* non-vacant slots                             */
  int64_t n_visible;           /* particles binned to pixels this frame        */
  int64_t n_birth_attempts;
  int64_t n_birth_success;
  int64_t n_resampled_voxels;
  int64_t n_moved;             /* particles copied by object moves             */
  int64_t n_move_reinserted;
  int64_t n_frustum_voxels;    /* voxels handled by the visibility BFS         */
  int64_t n_occupied;          /* voxels with occ > 0                          */
  int64_t alias_events;        /* owner-set inserts of an index that is already in another object's set */
  int64_t bfs_start_in_frustum;
  double stage_ms[8];          /* [1..7] per-stage wall time of the last update */
} oracle_stats;

oracle_map *oracle_create(const oracle_config *cfg);
void oracle_destroy(oracle_map *m);
oracle_map *oracle_clone(const oracle_map *m, int32_t bin_order);  /* deep copy (owner sets included), other summation order */
void oracle_clear(oracle_map *m);
void oracle_set_params(oracle_map *m, const oracle_params *p);
/* Gaussian noise table, normally 1,000,000 floats ~ N(0, 0.05^2)
 * (basic_algorithms.h:377-402); passed as data so both sides index the same floats */
void oracle_set_noise_table(oracle_map *m, const float *table, int32_t n);

/* One call of subObjectLevelUpdate (semantic_dsp_map.h:576-955) preceded by
 * global_time_stamp += 1 (semantic_dsp_map.h:173).  cam_q is (w,x,y,z). */
int oracle_update(oracle_map *m, const float *depth, const oracle_labeled_point *cloud,
                  const float cam_pos[3], const float cam_q[4],
                  const oracle_object_move *moves, int32_t n_moves,
                  const int32_t *remove_tracks, int32_t n_remove, int32_t stop_after);

void oracle_get_voxels(oracle_map *m, oracle_voxel_result *out);
void oracle_get_stats(oracle_map *m, oracle_stats *out);
void oracle_get_ring_state(oracle_map *m, oracle_ring_state *out);
void oracle_set_ring_state(oracle_map *m, const oracle_ring_state *in);
void oracle_get_stamps(oracle_map *m, uint32_t *sx, uint32_t *sy, uint32_t *sz);
void oracle_set_stamps(oracle_map *m, const uint32_t *sx, const uint32_t *sy, const uint32_t *sz);

/* SoA dump / load of every slot (V*S entries per array).  owner = track id of
 * the owner set that holds the index, 0xFFFF if none. */
void oracle_dump_state(oracle_map *m, float *px, float *py, float *pz, float *w, uint16_t *ts,
                       uint16_t *track, uint8_t *label, uint8_t *status, uint8_t *forget,
                       uint16_t *owner);
void oracle_load_state(oracle_map *m, const float *px, const float *py, const float *pz,
                       const float *w, const uint16_t *ts, const uint16_t *track,
                       const uint8_t *label, const uint8_t *status, const uint8_t *forget,
                       const uint16_t *owner);

/* diagnostics of the last update */
void oracle_get_ck_kappa(oracle_map *m, float *out /* H*W */);
void oracle_get_bin_counts(oracle_map *m, uint32_t *out /* H*W */);
/* concatenated bins in pixel order, returns number written (<= cap) */
int64_t oracle_get_bins(oracle_map *m, uint32_t *out, int64_t cap);
/* extrinsic used by the last update (row-major 4x4) */
void oracle_get_extrinsic(oracle_map *m, float *out16);
/* standard_gaussian_pdf, 20000 floats (basic_algorithms.h:405-407) */
void oracle_get_pdf_table(oracle_map *m, float *out);

/* PointCloudTools::generateLabeledPointCloud (utils/pointcloud_tools.h:88-310), general (non-BOOST, non-ZED2) path.
 * static_mask may be NULL; obj_masks = n_objects consecutive H*W masks; label_to_inst[256]. */
void oracle_generate_cloud(oracle_map *m, const float *depth, const uint8_t *static_mask, const uint16_t *label_to_inst,
                           const int32_t *obj_track, const int32_t *obj_label, const uint8_t *obj_masks, int32_t n_objects,
                           const double cam_pos[3], const double cam_q[4], int32_t consider_instance,
                           oracle_labeled_point *out);

/* The preset-specific parts (BOOST-mode manualResize, pointcloud_tools.h:1104-1133; ZED2 sky exclusion and per-object
 * box filter, :174-196, 236-242, 254-272).  src_width = 0: no resize; sky_instance < 0 / object_bbox NULL: off.
 * Inputs are src_width x src_height when resizing.  depth_out (H*W, may be NULL) receives the resized depth image. */
void oracle_generate_cloud_ex(oracle_map *m, const float *depth, const uint8_t *static_mask, const uint16_t *label_to_inst,
                              const int32_t *obj_track, const int32_t *obj_label, const uint8_t *obj_masks, int32_t n_objects,
                              const double cam_pos[3], const double cam_q[4], int32_t consider_instance,
                              int32_t src_width, int32_t src_height, float rescale, int32_t sky_instance,
                              const double *object_bbox, oracle_labeled_point *out, float *depth_out);

/* helpers exposed for known-answer tests */
uint32_t oracle_pos_to_voxel(oracle_map *m, float x, float y, float z); /* 0xffffffff if outside */
void oracle_voxel_to_pos(oracle_map *m, uint32_t voxel, float out[3]);  /* global min corner */
int32_t oracle_point_in_frustum(oracle_map *m, float x, float y, float z); /* operations.h:1240-1258, last frame's extrinsic */
float oracle_query_pdf(oracle_map *m, float x, float mu, float sigma);
float oracle_forgetting_factor(oracle_map *m, int32_t forget_count);
uint32_t oracle_add_particle(oracle_map *m, float x, float y, float z, uint8_t label, uint16_t track);
int32_t oracle_resample_voxel(oracle_map *m, uint32_t voxel);
void oracle_set_global_time_stamp(oracle_map *m, uint32_t t);

#ifdef __cplusplus
}
#endif
