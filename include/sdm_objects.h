/* sdm_objects.h - C ABI of the object layer, host side, no GPU involved.
 *
 * What it replaces in the reference (all host code, O(objects) per frame):
 *   SemanticDSPMap::objectLevelUpdate            include/semantic_dsp_map.h:304-566
 *   the object loop at the head of the prediction include/semantic_dsp_map.h:588-736 (which objects move by which 4x4
 *                                                 matrix, which are wiped) - the part that PRODUCES the inputs of sdm_update
 *   ObjectSet / MJObject / ObjectTransformations /
 *   MotionEstimation                              include/object_layer.h:57-648
 *   estimateTransformation(+RANSAC)               include/utils/basic_algorithms.h:54-195
 * The owner sets (ObjectParticleHashMap, object_layer.h:20-52) are NOT here: they live in HBM next to the particles
 * (sdm.h: sdm_object_particle_count, removal list of sdm_update).
 *
 * Differences from the reference, all deliberate (DESIGN.md "Object layer"):
 *   - the RANSAC sampler is seeded (the reference seeds std::mt19937 from std::random_device, basic_algorithms.h:110-112):
 *     the stream of a call depends on (config seed, global time stamp, track id) only, so a clip replays identically and
 *     the result does not depend on the order the objects are listed in;
 *   - SETTING (settings.h:22, a compile-time switch of the reference) is the run-time field `mode`;
 *   - moves and removals come out in ascending track-id order (the reference iterates an unordered_map);
 *   - the template-matching branch (semantic_dsp_map.h:616-672) is dead in the reference (getFlagUseTemplateMatching()
 *     is false in every shipped configuration) and is not built; its flag is kept and can be read back;
 *   - the angular velocity of MotionEstimation::estimate (object_layer.h:166-169) feeds nothing - the predicted matrix
 *     is translation-only (:192-197) - and is not computed.
 * Everything else, including the reference's quirks (PINNED list in DESIGN.md), is kept: division by n-1 in the velocity
 * mean, the out-of-view test that only looks at the last keypoint, the moving test before the probability is clamped.
 *
 * All functions return sdm_status (sdm.h) and never throw; pointers are not retained past the call.
 */
#ifndef SDM_OBJECTS_H_
#define SDM_OBJECTS_H_

#include <stdint.h>
#include "sdm.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sdm_objects sdm_objects;

/* settings.h:22 */
enum { SDM_OBJECTS_MODE_KITTI360 = 0, SDM_OBJECTS_MODE_CODA = 1, SDM_OBJECTS_MODE_VKITTI2 = 2, SDM_OBJECTS_MODE_ZED2 = 3 };

typedef struct {
  int32_t mode;                           /* SETTING: 0 static scene, 1 always moving, 2 matched keypoints + Bayes, 3 box keypoints + Bayes */
  int32_t max_movable_instance_id;        /* g_max_movable_object_instance_id (data_base.h): larger track ids are static */
  double movement_distance_threshold;     /* beyesian_movement_distance_threshold_     (semantic_dsp_map.h:36, cfg/options*.yaml) */
  double movement_probability_threshold;  /* beyesian_movement_probability_threshold_  (:37) */
  double movement_increment;              /* beyesian_movement_increment_              (:38) */
  double movement_decrement;              /* beyesian_movement_decrement_              (:39) */
  double map_half_size_scaled;            /* C_VOXEL_SIZE * 2^(N_BIGGEST-1) * 1.2      (:356) */
  double fx, fy, cx, cy;                  /* isPointOutOfFOV (:1421-1442); mode 0/3 only */
  int32_t image_width, image_height;
  uint64_t seed;                          /* RANSAC sampler */
} sdm_objects_config;

/* one entry of ins_seg_result (MaskKpts, utils/tracking_result_handler.h:15-26) without the mask */
typedef struct {
  int32_t track_id;
  int32_t label_id;            /* g_label_id_map_default[label]; < 0 = label not in the table (object ignored, :331-334) */
  int32_t is_static;           /* label == "static" (:317) */
  int32_t n_kpts;              /* kpts_current.size(); kpts_previous must hold as many in modes 1/2 */
  const double *kpts_current;  /* 3 * n_kpts, global frame */
  const double *kpts_previous; /* 3 * n_kpts or NULL (modes 0/3 keep their own previous keypoints, :435-477) */
} sdm_object_observation;

typedef struct {
  int32_t exists;
  int32_t label_id;
  int32_t observation_time_step;
  int32_t observation_count;
  int32_t has_moved_flag;        /* rigidbody_moved_vec.size() > 0 */
  int32_t moving;                /* rigidbody_moved_vec[0] */
  int32_t to_match_with_previous;
  int32_t prediction_available;  /* ObjectTransformations::checkIfUpdated() */
  int32_t n_transformations;     /* window length of ObjectTransformations (<= 5) */
  int32_t has_t_matrix;          /* rigidbody_tmatrix_vec.size() > 0 */
  double moved_probability;
  double translation_velocity[3];
  double t_matrix[16];           /* rigidbody_tmatrix_vec[0], row major */
} sdm_object_info;

sdm_status sdm_objects_create(const sdm_objects_config *cfg, sdm_objects **out);
void sdm_objects_destroy(sdm_objects *h);
/* ObjectSet::clear (object_layer.h:379-384) + the keypoint maps of SemanticDSPMap */
sdm_status sdm_objects_clear(sdm_objects *h);
/* SemanticDSPMap::setBeyesianMovementParameters (semantic_dsp_map.h:142-148) */
sdm_status sdm_objects_set_bayes(sdm_objects *h, double distance_threshold, double probability_threshold, double increment,
                                 double decrement);

/* objectLevelUpdate (semantic_dsp_map.h:304-566).  cam_q = (w, x, y, z).  global_time_stamp is the value AFTER the
 * increment at the head of SemanticDSPMap::update (:173). */
sdm_status sdm_objects_update(sdm_objects *h, const sdm_object_observation *obs, int32_t n_obs, const double cam_pos[3],
                              const double cam_q[4], double time_stamp, uint32_t global_time_stamp);

/* The object loop of the prediction step (semantic_dsp_map.h:588-736): the objects to move (track id + float 4x4
 * matrix = sdm_object_move, what sdm_update takes), the objects lost for max_obersevation_lost_time frames and the
 * "floating" ones - tracks that own particles in the map (present_tracks, may be NULL) but are not tracked - to wipe.
 * Lost and floating objects leave the tracker here, as in the reference.  Ascending track ids. */
sdm_status sdm_objects_collect(sdm_objects *h, uint32_t global_time_stamp, int32_t max_obersevation_lost_time,
                               const int32_t *present_tracks, int32_t n_present, sdm_object_move *moves, int32_t moves_cap,
                               int32_t *n_moves, int32_t *remove_tracks, int32_t remove_cap, int32_t *n_remove);

sdm_status sdm_objects_query(sdm_objects *h, int32_t track_id, sdm_object_info *out);
sdm_status sdm_objects_count(sdm_objects *h, int32_t *n_tracked);

/* The two numerical routines on their own (tests, other callers).  P, Q: 3 x n column major = n points xyz.  T: 4x4 row
 * major.  estimateTransformation (basic_algorithms.h:54-92) and estimateTransformationRANSAC (:104-195; seeded). */
sdm_status sdm_objects_fit_rigid(const double *P, const double *Q, int32_t n, double T[16]);
sdm_status sdm_objects_fit_rigid_ransac(const double *P, const double *Q, int32_t n, int32_t max_iterations, double threshold,
                                        int32_t recompute_with_inliers, uint64_t seed, double T[16], int32_t *inliers,
                                        int32_t *n_inliers, double *mse_inliers);

#ifdef __cplusplus
}
#endif
#endif /* SDM_OBJECTS_H_ */
