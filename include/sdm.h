/*
 * sdm.h — C ABI of libsdm_hip: the MI355X-native particle-grid update of
 * tud-amr/semantic_dsp_map (hot path only).
 *
 * The library replaces the private sub-object level of the reference's
 * SemanticDSPMap class; each entry point cites the reference interface it stands
 * in for (paths relative to the reference's include/).  The header-only adapter
 * include/semantic_dsp_map.h puts the reference's own class and method
 * signatures back on top of these calls.
 *
 * Conventions: plain C, caller owns every pointer, nothing is retained after a
 * call returns, no exceptions cross the boundary, every function returns an
 * sdm_status (0 = ok).  One map per handle (the reference allows one per
 * process: its map is a set of header-defined globals, mc_ring/buffer.h:86-120).
 * Calls on one handle must come from one thread at a time (the reference is
 * single-threaded, src/mapping.cpp:156).
 */
#ifndef SDM_H_
#define SDM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sdm_map sdm_map;

typedef enum {
  SDM_OK = 0,
  SDM_ERR_INVALID_ARGUMENT = 1, /* bad config / null pointer / size mismatch          */
  SDM_ERR_NO_DEVICE = 2,        /* no HIP device, or the requested ordinal is missing */
  SDM_ERR_HIP = 3,              /* a HIP runtime call failed (sdm_last_error has text)  */
  SDM_ERR_CAPACITY = 4,         /* a per-frame work list overflowed its capacity      */
  SDM_ERR_NOT_CONVERGED = 5,    /* frustum flood fill needed more rounds than budgeted */
  SDM_ERR_COMM = 6              /* multi-GPU exchange failed                           */
} sdm_status;

/* Compile-time constants of the reference (settings/settings.h:18-150:
 * C_VOXEL_NUM_AXIS_*_N, C_MAX_PARTICLE_NUM_PER_VOXEL_N, C_VOXEL_SIZE, g_camera_*,
 * g_image_*, g_depth_range_*, BOOST_MODE window) as run-time configuration. */
typedef struct {
  int32_t x_n, y_n, z_n;      /* log2 voxels per axis (x_n+y_n+z_n+p_n <= 31, operations.h:54-58) */
  int32_t p_n;                /* log2 slots per voxel, 1..4; slot 0 is the time particle         */
  float voxel_size;           /* metres                                                           */
  float fx, fy, cx, cy;       /* pinhole intrinsics                                               */
  int32_t width, height;      /* image size                                                       */
  float depth_min, depth_max; /* metres                                                           */
  int32_t window_half;        /* SMC-PHD neighbour half-size: 5, or 3 in BOOST mode (semantic_dsp_map.h:964-970) */
  int32_t max_movable_track;  /* g_max_movable_object_instance_id (utils/data_base.h:196)         */
  int32_t device;             /* HIP device ordinal                                               */
  int32_t shard_rank;         /* Z-slab sharding: this process owns ring-z slab shard_rank of shard_count */
  int32_t shard_count;        /* 1 = whole map on one GPU                                         */
  int64_t max_visible;        /* capacity of the per-frame visible-particle list, 0 = default     */
} sdm_config;

/* SemanticDSPMap::setMapParameters / setMapOptions / setDepthNoiseModelParameters
 * (semantic_dsp_map.h:101-125, 161-166).  Field spelling follows the reference. */
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
} sdm_params;

/* LabeledPoint (utils/data_base.h:78-92): position in the global frame, sigma,
 * track id, label id, validity.  20 bytes, same field order. */
typedef struct {
  float x, y, z;
  float sigma;
  uint16_t track_id;
  uint8_t label_id;
  uint8_t is_valid;
} sdm_labeled_point;

/* One rigid-body motion handed down by the object layer: the track id and
 * rigidbody_tmatrix_vec[0] cast to float (semantic_dsp_map.h:673-679), row-major. */
typedef struct {
  int32_t track_id;
  float T[16];
} sdm_object_move;

/* Per-voxel result of determineIfVoxelOccupied (mc_ring/operations.h:623-639),
 * indexed by storage voxel index.  8 bytes. */
typedef struct {
  float wsum;     /* weight sum, -1 = unknown voxel                       */
  uint16_t track; /* winning track id (0 if none)                          */
  uint8_t label;  /* its label id                                          */
  int8_t occ;     /* -1 unknown, 0 free, 1 occupied, 2 guessed occupied    */
} sdm_voxel_result;

/* One emitted voxel of getOccupancyResult (semantic_dsp_map.h:1239-1383): min-corner
 * position (global frame, or camera-centred) and the raw semantics; colouring stays
 * in the adapter. 16 bytes. */
typedef struct {
  float x, y, z;
  uint16_t track;
  uint8_t label;
  int8_t occ;
} sdm_point;

/* What one block of kernel arguments holds of the object lists.  Every entry point takes lists of ANY length (whole maps:
 * round 5; Z-slab shards: round 6): they are worked off in batches of this size inside the frame - every object's
 * particles are taken out before any is re-inserted and the noise draws run on across the batches, like
 * moveParticlesInSetsByTransformations does it (mc_ring/operations.h:321-362); a frame with longer lists is issued launch
 * by launch, not from the captured graph.  On a Z-slab shard every batch's per-object member counts are exchanged before
 * the batch is applied: sdm_update_sharded does that itself, the split entry points through sdm_frame_moves_pending. */
#define SDM_MAX_MOVES 48
#define SDM_MAX_REMOVALS 128

/* sdm_update flags */
#define SDM_INPUT_ON_DEVICE 0x1u /* depth / cloud are device pointers already resident in HBM */
#define SDM_SKIP_OCCUPANCY 0x2u  /* do not run the occupancy sweep (debug)                     */
#define SDM_NO_INSTANCES 0x4u    /* sdm_update_raw: ignore the object masks (g_consider_instance == false, settings.h:49) */

/* stage ids (also indices of sdm_stats.stage_ms): the reference's own stage timers,
 * semantic_dsp_map.h:916-921 */
enum {
  SDM_STAGE_ALL = 0,
  SDM_STAGE_EGO = 1,
  SDM_STAGE_MOVE = 2,
  SDM_STAGE_REMOVE = 3,
  SDM_STAGE_VISIBILITY = 4,
  SDM_STAGE_WEIGHT = 5,
  SDM_STAGE_BIRTH = 6,
  SDM_STAGE_OCCUPANCY = 7
};

typedef struct {
  uint32_t global_time_stamp;
  int32_t moved_steps[3];
  int32_t eq_steps[3];
  float map_center[3];
  float last_pos[3];
  int32_t birth_cursor;
  int32_t move_cursor;
} sdm_ring_state;

typedef struct {
  int64_t live_particles;   /* filled by sdm_get_stats(…, count_live=1) only */
  int64_t n_visible;
  int64_t n_birth_attempts;
  int64_t n_birth_success;
  int64_t n_resampled_voxels;
  int64_t n_moved;
  int64_t n_move_reinserted;
  int64_t n_frustum_voxels;
  int64_t n_occupied;
  int64_t flood_rounds;
  int64_t bfs_start_in_frustum;
  int64_t live_voxels;      /* observed voxels holding a live slot (count_live=1) */
  int64_t sweep_live_voxels; /* voxels the last occupancy sweep evaluated in full: the ones written to since the sweep before */
  int64_t sweep_tiles;      /* 2048-voxel tiles the last occupancy sweep looked into (something in them was written or stamped) */
  double stage_ms[8];       /* GPU time per stage of the last update when profiling is on */
  int64_t restamped_slabs[3]; /* slabs the last update's ring shift re-stamped, per axis (x, y, z) */
  int64_t graph_frames;     /* frames replayed from the captured hipGraph so far ...                       */
  int64_t direct_frames;    /* ... and frames issued launch by launch                                      */
  double host_enqueue_us;   /* host time to issue 50 empty kernel launches (a frame's worth), measured at sdm_create */
  int64_t halo_dropped;     /* sharded maps: slab-crossing copies of the last update beyond the export capacity towards their
                               destination shard (dropped; SDM_ERR_CAPACITY at the next sdm_synchronize) */
  int64_t alias_entries;    /* indices that sit in a second owner set besides their latest one (the reference's sets are real
                               sets, object_layer.h:20-52), as of now - deleted entries included until the next frame's
                               garbage collection */
  int64_t alias_overflowed; /* 1 = that table has overflowed (65536 entries) since the last sdm_clear / sdm_load_state: the
                               sets are incomplete, SDM_ERR_CAPACITY is reported at every synchronisation from then on */
} sdm_stats;

/* ---- host placement.  A frame is a chain of ~50 dependent launches; the command processor fetches every packet and
 * signals every completion through host memory the HIP runtime allocates where the calling thread runs.  From the
 * socket the GPU does not hang off, every gap between two dependent kernels is 2-4 us longer (C3 frame: 0.292 against
 * 0.265 ms).  sdm_bind_host_thread moves the CALLING thread (and the threads it starts afterwards) onto the NUMA node
 * of `device`, within the CPUs the process may use, and returns that node (-1: nothing to do - one node, no sysfs entry,
 * no allowed CPU there).  Best called first thing in the process, before any other HIP call: it finds the node in sysfs
 * (KFD topology) without touching the runtime, so that the runtime's first allocations land on the right node too; only
 * when sysfs cannot tell does it ask HIP for the device's PCI address.
 * sdm_create calls it for its device unless SDM_NUMA_BIND=0 is set in the environment. */
int32_t sdm_bind_host_thread(int32_t device);
/* The NUMA node of HIP device `device` found WITHOUT the HIP runtime (KFD topology in sysfs + the *_VISIBLE_DEVICES index
 * lists), which is what sdm_bind_host_thread tries first: -2 if that cannot tell (it then asks HIP, i.e. brings the runtime
 * up before the thread is moved). */
int32_t sdm_host_numa_node_early(int32_t device);

/* ---- life cycle: SemanticDSPMap() / ~SemanticDSPMap() / clear() (semantic_dsp_map.h:25-81),
 * RingBufferOperations::initialize / clear (mc_ring/operations.h:684-767) */
sdm_status sdm_create(const sdm_config *cfg, sdm_map **out);
sdm_status sdm_destroy(sdm_map *m);
sdm_status sdm_clear(sdm_map *m);

/* ---- setters (semantic_dsp_map.h:101-166) */
sdm_status sdm_set_params(sdm_map *m, const sdm_params *p);

/* ---- Gaussian noise table, GaussianRandomCalculator::calculateGaussianTable
 * (utils/basic_algorithms.h:394-402): n floats ~ N(0, stddev^2).  Either generated on
 * the device with rocRAND (Philox4x32-10) or uploaded by the caller. */
sdm_status sdm_generate_noise_table(sdm_map *m, uint64_t seed, int32_t n, float stddev);
sdm_status sdm_upload_noise_table(sdm_map *m, const float *table, int32_t n);
sdm_status sdm_download_noise_table(sdm_map *m, float *table, int32_t n);
/* standard_gaussian_pdf (basic_algorithms.h:405-407), 20000 floats, for cross-checks */
sdm_status sdm_download_pdf_table(sdm_map *m, float *table, int32_t n);

/* ---- the hot path: one call of SemanticDSPMap::subObjectLevelUpdate
 * (semantic_dsp_map.h:576-955) preceded by global_time_stamp += 1 (:173).
 *   depth          H*W float32 metres (cv::Mat CV_32FC1, src/mapping.cpp:180)
 *   cloud          H*W labeled points (output of generateLabeledPointCloud, :227)
 *   cam_pos,cam_q  camera pose in the global frame, q = (w,x,y,z), already cast to float (:584, :745-746)
 *   moves          objects the object layer decided to move this frame (:593-693), caller's order
 *   remove_tracks  lost / floating objects to wipe (:702-736)
 *   stop_after     SDM_STAGE_ALL, or a stage id to stop after (parity debugging)
 * Asynchronous: returns after enqueueing; results are fetched with the getters below.
 * How the frame's ~50 kernels are issued follows the host's speed at issuing launches, measured in sdm_create
 * (sdm_stats.host_enqueue_us): launch by launch on a fast host, replayed from hipGraphs on a slow one.  The environment
 * variable SDM_GRAPH forces one way (0 launch by launch, 4 five chain graphs, 3 one chain graph, 1 one branched graph;
 * read when the map is created), and so does sdm_set_issue_mode at any time between frames; the results are
 * bit-identical. */
#define SDM_ISSUE_LAUNCHES 0   /* launch by launch */
#define SDM_ISSUE_BRANCHED 1   /* one hipGraph, frustum and birth chains as branches */
#define SDM_ISSUE_AUTO 2       /* by the host's measured speed (the default) */
#define SDM_ISSUE_CHAIN 3      /* one hipGraph, a chain of kernel nodes */
#define SDM_ISSUE_PIECES 4     /* five chain hipGraphs on the frame's streams */
/* mode: one of SDM_ISSUE_*.  Graphs captured for another mode are dropped; the next plain frame captures anew. */
sdm_status sdm_set_issue_mode(sdm_map *m, int32_t mode);
sdm_status sdm_update(sdm_map *m, const float *depth, const sdm_labeled_point *cloud,
                      const float cam_pos[3], const float cam_q[4],
                      const sdm_object_move *moves, int32_t n_moves,
                      const int32_t *remove_tracks, int32_t n_remove,
                      uint32_t flags, int32_t stop_after);

/* ---- the step right before the path, PointCloudTools::generateLabeledPointCloud
 * (utils/pointcloud_tools.h:88-310, general non-BOOST / non-ZED2 path), on the device.  Instead of the 20-byte
 * LabeledPoint image the caller hands over what the reference's update() receives: the depth image, the "static" MONO8
 * mask (pixel value + 1 = label id, :137-138; NULL = every pixel Background/65535, :147-156) and one MONO8 mask per
 * movable object (> 0 = object, later entries override earlier ones, :163-213), plus the camera pose in double
 * (the reference back-projects in double and casts to float, :243-249, 298-300).
 * label_to_static_instance[256]: instance id of a static label id (g_label_to_instance_id_map_default), 65535 for
 * label ids without one.  Every mask is H*W bytes, host memory (or device memory with SDM_INPUT_ON_DEVICE). */
typedef struct {
  int32_t track_id;    /* <= max_movable_track */
  int32_t label_id;    /* g_label_id_map_default[label] */
  const uint8_t *mask; /* H*W, > 0 = object */
} sdm_instance_mask;
sdm_status sdm_update_raw(sdm_map *m, const float *depth, const uint8_t *static_mask,
                          const uint16_t label_to_static_instance[256],
                          const sdm_instance_mask *objects, int32_t n_objects,
                          const double cam_pos[3], const double cam_q[4],
                          const sdm_object_move *moves, int32_t n_moves,
                          const int32_t *remove_tracks, int32_t n_remove, uint32_t flags, int32_t stop_after);
/* Preset-specific parts of generateLabeledPointCloud:
 *  - BOOST mode (settings.h:26, 137-143): depth and masks arrive at src_width x src_height and are reduced by `rescale`
 *    with the reference's nearest-neighbour manualResize (pointcloud_tools.h:1104-1133); int(src * rescale) must equal
 *    the configured width / height.  src_width = 0: inputs already have the configured size.
 *  - ZED2 (SETTING 3): pixels whose track id is sky_instance are invalid (:236-242); object_bbox (n_objects x 6
 *    doubles: min x, max x, min y, max y, min z, max z of the object's current key points +- 1 m, global frame,
 *    :174-196) turns points of a movable instance that lie outside their object's box into Background (:254-272).
 *    sky_instance < 0 / object_bbox NULL: off. */
typedef struct {
  int32_t src_width, src_height;
  float rescale;
  int32_t sky_instance;
  const double *object_bbox;
} sdm_raw_options;
sdm_status sdm_update_raw_ex(sdm_map *m, const float *depth, const uint8_t *static_mask,
                             const uint16_t label_to_static_instance[256],
                             const sdm_instance_mask *objects, int32_t n_objects,
                             const double cam_pos[3], const double cam_q[4],
                             const sdm_object_move *moves, int32_t n_moves,
                             const int32_t *remove_tracks, int32_t n_remove, uint32_t flags, int32_t stop_after,
                             const sdm_raw_options *opt);
/* the LabeledPoint image sdm_update_raw generated for the last frame (H*W entries), for cross-checks */
sdm_status sdm_get_labeled_cloud(sdm_map *m, sdm_labeled_point *out);

/* The same frame split at its one cross-shard dependency: sdm_update_begin runs the
 * prediction, visibility/binning and this shard's partial ck image (pass 1 of updateParticles,
 * semantic_dsp_map.h:973-1037) and hands back its device pointer; the partial images of all Z-slab shards are summed
 * IN SLAB ORDER (see sdm_ck_reduce for the exchange that does it with 2 (G-1)/G images of traffic) and the result goes
 * to sdm_update_finish, which forms ck+kappa, updates the weights and runs births, resampling and the occupancy sweep.
 * sdm_update_finish accepts either n_parts whole partial images in slab order (it adds them itself) or, with
 * n_parts = 1, the image already summed.  sdm_update == begin + finish with the shard's own image. */
sdm_status sdm_update_begin(sdm_map *m, const float *depth, const sdm_labeled_point *cloud,
                            const float cam_pos[3], const float cam_q[4],
                            const sdm_object_move *moves, int32_t n_moves,
                            const int32_t *remove_tracks, int32_t n_remove,
                            uint32_t flags, int32_t stop_after, const float **ck_part_dev);
sdm_status sdm_update_finish(sdm_map *m, const float *ck_parts_dev, int32_t n_parts, uint32_t flags,
                             int32_t stop_after);
/* sdm_update_begin itself in three steps, for sharded maps whose moving objects cross slab borders
 * (moveParticlesInSetsByTransformations, mc_ring/operations.h:321-362, needs the other shards twice):
 *   sdm_frame_start   -> all-gather of the per-object member counts (SDM_HALO_OBJ int32 per shard)
 *   sdm_frame_moves   -> all-to-all of the export segments: a copy whose target voxel lies in another slab goes to
 *                        the shard that owns that slab, and to nobody else
 *   sdm_frame_predict -> chunk-owner exchange of the partial ck images (below) -> sdm_update_finish
 * The exchange buffers are device memory owned by the caller and registered once with sdm_set_halo_buffers:
 *   counts_local  SDM_HALO_OBJ int32                      counts_all  shard_count x SDM_HALO_OBJ int32 (gathered)
 *   send, recv    shard_count segments of SDM_HALO_HEADER_BYTES + cap_records x SDM_HALO_RECORD_BYTES each; segment d of
 *                 send is addressed to shard d, segment s of recv is what shard s addressed to this one.
 * cap_records is the capacity PER DESTINATION; more slab-crossing copies towards one shard in one frame than that is a
 * capacity error (SDM_ERR_CAPACITY at the next sdm_synchronize; the copies beyond it are dropped like copies whose
 * target voxel is full, mc_ring/operations.h:357). */
#define SDM_HALO_OBJ 64
#define SDM_HALO_RECORD_BYTES 36
#define SDM_HALO_HEADER_BYTES 16
#define SDM_HALO_DEFAULT_CAP 4096
sdm_status sdm_frame_start(sdm_map *m, const float *depth, const sdm_labeled_point *cloud,
                           const float cam_pos[3], const float cam_q[4],
                           const sdm_object_move *moves, int32_t n_moves,
                           const int32_t *remove_tracks, int32_t n_remove, uint32_t flags, int32_t stop_after);
sdm_status sdm_frame_moves(sdm_map *m);
/* More than SDM_MAX_MOVES moving objects on a Z-slab shard: sdm_frame_moves applies ONE batch and, if objects are left,
 * issues the next batch's member count and publishes its per-object counts in counts_local; *pending = 1 then says: gather
 * the count rows once more (the same all-gather as behind sdm_frame_start) and call sdm_frame_moves again.  The export
 * segments collect the copies of all batches; they are exchanged once, when nothing is pending any more. */
sdm_status sdm_frame_moves_pending(sdm_map *m, int32_t *pending);
sdm_status sdm_frame_predict(sdm_map *m, const float **ck_part_dev);
sdm_status sdm_set_halo_buffers(sdm_map *m, int32_t *counts_local, const int32_t *counts_all, void *send,
                                const void *recv_all, int32_t cap_records);
/* the HIP stream (hipStream_t) all work of this map is enqueued on; sdm_set_stream moves the map onto a
 * caller-owned stream (e.g. the one RCCL collectives are issued on, so that no host sync is needed between
 * sdm_update_begin, the all-gather and sdm_update_finish); NULL restores the map's own stream. */
sdm_status sdm_stream(sdm_map *m, void **stream_out);
sdm_status sdm_set_stream(sdm_map *m, void *hip_stream);
/* caller-provided device buffer that sdm_update_begin writes this shard's partial ck image to (H*W floats; for the
 * chunk-owner exchange shard_count x sdm_ck_chunk_elems floats, the image padded to whole chunks); NULL restores the
 * internal buffer. */
sdm_status sdm_set_ck_buffer(sdm_map *m, float *dev_buffer);
/* Chunk-owner exchange of the partial ck images: the image is cut into shard_count chunks of `chunk` pixels
 * (sdm_ck_chunk_elems: ceil(H*W / shard_count) rounded up to 64), shard r owns chunk r.
 *   1. all-to-all: chunk d of every shard's partial image goes to shard d  -> stage (shard_count x chunk floats, part s
 *      from shard s)
 *   2. sdm_ck_reduce: the owner adds the parts in slab order (the float sums a single map split into the same slabs
 *      forms, oracle `ck_slabs`) into chunk `shard_rank` of full (shard_count x chunk floats)
 *   3. all-gather of the summed chunks (in place on full)               -> sdm_update_finish(m, full, 1, ...)
 * 2 (G-1)/G images received per shard instead of the G-1 whole images of a plain all-gather. */
sdm_status sdm_ck_chunk_elems(sdm_map *m, int64_t *chunk_out);
sdm_status sdm_ck_reduce(sdm_map *m, const float *stage_dev, float *full_dev);

/* ---- native multi-GPU path: RCCL over xGMI, one process per GPU.  Rank 0 draws an id, the caller hands it to
 * every rank (any out-of-band channel), every rank calls sdm_comm_init on its shard map (collective), then
 * sdm_update_sharded runs whole frames: the exchanges above are issued on the map's streams, nothing waits on the host. */
sdm_status sdm_comm_unique_id(uint8_t out[128]);
/* halo_cap_records: capacity per destination shard of the export segments, 0 = SDM_HALO_DEFAULT_CAP */
sdm_status sdm_comm_init(sdm_map *m, const uint8_t id[128], int32_t halo_cap_records);
/* The same exchanges WITHOUT RCCL, through peer-mapped memory (xGMI is point to point: a shard writes its pieces straight
 * into the peers' HBM and raises a flag; one small kernel per exchange, no collective launch, no host call but that).
 * sdm_ipc_create allocates this shard's receive arena and returns its hipIpc handle (64 bytes); the caller gathers the
 * handles of all shards - any transport, the order is the shard order - and sdm_ipc_connect maps the peers' arenas.
 * sdm_update_sharded then runs on them (sdm_comm_init is not needed; a map uses one or the other).  A shard that is missing
 * or out of step: SDM_ERR_COMM at the next sdm_synchronize, after SDM_COMM_TIMEOUT_MS (sdm_comm_set_options(m, -1, ms)). */
sdm_status sdm_ipc_create(sdm_map *m, int32_t halo_cap_records, uint8_t handle_out[64]);
sdm_status sdm_ipc_connect(sdm_map *m, const uint8_t *handles_all);
/* How sdm_update_sharded combines the shards' partial ck images, and how long sdm_synchronize waits for a sharded frame.
 * ck_exchange: 0 = chunk-owner reduction (all-to-all of the chunks to their owners, slab-ordered sum there, all-gather of the
 * summed chunks: 2 (G-1)/G images received, two collectives on the critical path; the default), 1 = ONE all-gather of the
 * whole partial images and the slab-ordered sum on every shard (G-1 images received, one collective), -1 = leave as it is
 * (environment: SDM_CK_EXCHANGE=allgather).  Both form the same float sums.  All shards must choose alike.
 * timeout_ms > 0: bound of sdm_synchronize's wait (default 30000, SDM_COMM_TIMEOUT_MS): beyond it - a peer died or fell out
 * of step - the communicator is aborted and SDM_ERR_COMM returned instead of hanging. */
sdm_status sdm_comm_set_options(sdm_map *m, int32_t ck_exchange, int32_t timeout_ms);
sdm_status sdm_update_sharded(sdm_map *m, const float *depth, const sdm_labeled_point *cloud,
                              const float cam_pos[3], const float cam_q[4],
                              const sdm_object_move *moves, int32_t n_moves,
                              const int32_t *remove_tracks, int32_t n_remove, uint32_t flags);
/* HIP events around the four collectives of sdm_update_sharded frames; sdm_get_comm_times waits for the last frame and
 * returns their GPU times in microseconds: [0] member counts (all-gather), [1] slab-crossing copies (all-to-all),
 * [2] partial ck chunks to their owners (all-to-all), [3] summed ck chunks (all-gather); 0 for one the frame did not
 * issue.  The time of a collective includes waiting for the slowest shard to arrive at it. */
sdm_status sdm_comm_timing(sdm_map *m, int32_t on);
sdm_status sdm_get_comm_times(sdm_map *m, double out_us[4]);

/* ---- plain device buffers (for frames kept resident in HBM, see SDM_INPUT_ON_DEVICE) */
/* page-locked host memory for input buffers (uploads then run at PCIe speed, beside the previous frame's kernels) */
sdm_status sdm_host_alloc(size_t bytes, void **out);
sdm_status sdm_host_free(void *p);
sdm_status sdm_device_alloc(sdm_map *m, size_t bytes, void **out);
sdm_status sdm_device_free(sdm_map *m, void *p);
sdm_status sdm_device_upload(sdm_map *m, void *dst_dev, const void *src_host, size_t bytes);
sdm_status sdm_device_download(sdm_map *m, void *dst_host, const void *src_dev, size_t bytes);
sdm_status sdm_device_synchronize(sdm_map *m);

/* Wait for all enqueued work of this map; surfaces deferred device-side errors. */
sdm_status sdm_synchronize(sdm_map *m);

/* ---- results (getOccupancyResult, semantic_dsp_map.h:1239-1383) */
/* all voxels in storage order; out has 2^(x_n+y_n+z_n) entries (this shard's slab if sharded) */
sdm_status sdm_get_voxels(sdm_map *m, sdm_voxel_result *out);
/* compacted list of voxels with occ > 0 (occupied) or occ == 0 (free) in increasing storage index.
 * flags: SDM_POINTS_ZERO_CENTER subtracts the camera position (visualize_with_zero_center_, semantic_dsp_map.h:1263-1271);
 * SDM_POINTS_MARK_FOV adds SDM_OCC_OUT_OF_FOV to sdm_point.occ of voxels whose global position fails
 * checkIfPointInFrustum against the last frame's extrinsic (the test that selects the HSV dimming,
 * semantic_dsp_map.h:1339-1342). */
#define SDM_POINTS_ZERO_CENTER 0x1
#define SDM_POINTS_MARK_FOV 0x2
#define SDM_OCC_OUT_OF_FOV 0x40
sdm_status sdm_get_occupied(sdm_map *m, sdm_point *out, size_t cap, size_t *n_out, int32_t flags);
sdm_status sdm_get_freespace(sdm_map *m, sdm_point *out, size_t cap, size_t *n_out, int32_t flags);
/* ---- the same lists coloured and packed on the device, ready for the ROS message.
 * getOccupancyResult's colour rules (semantic_dsp_map.h:1274-1351): Background voxels by height through the jet map
 * (:50-63, :1277-1294), static instances (track id > max_movable_track, or every label when colour_by_label is set:
 * SETTING == 0) by their label colour (utils/data_base.h:216-232), movable ones (160, perm[track & 255], perm[label])
 * or, in evaluation format, (label, track >> 8, track & 255); guessed-occupied voxels white; free space green; outside
 * the evaluation format every colour then takes OpenCV's 8-bit RGB -> HSV -> RGB round trip with V x 0.7 for voxels
 * outside the camera frustum (:1333-1351; restated from OpenCV 4.x, see oracle/colour.py).
 * One point = pcl::PointXYZRGB's 32 bytes: x, y, z, 1.0f, then b, g, r, a = 255, then 12 bytes of padding. */
typedef struct {
  float x, y, z, one;
  uint8_t b, g, r, a;
  uint32_t pad[3];
} sdm_point_xyzrgb;
typedef struct {
  uint8_t label_bgr[256][3]; /* g_label_color_map_default, BGR like cv::Vec3b                       */
  uint8_t perm[256];         /* color_map_int_256_ (semantic_dsp_map.h:45-48)                         */
  int32_t background_label;  /* g_label_id_map_default["Background"]                                */
  int32_t colour_by_label;   /* SETTING == 0 (:1296-1302)                                            */
  int32_t jet_axis;          /* 0: jet index from -z + 2 (default), 1: from y + 2 (SETTING == 3)       */
  int32_t evaluation_format; /* if_out_evaluation_format_ (:130-134)                                 */
} sdm_colour_config;
sdm_status sdm_set_colours(sdm_map *m, const sdm_colour_config *c);
/* flags: SDM_POINTS_ZERO_CENTER as above (the frustum test always uses the uncentred position) */
sdm_status sdm_get_occupied_rgb(sdm_map *m, sdm_point_xyzrgb *out, size_t cap, size_t *n_out, int32_t flags);
sdm_status sdm_get_freespace_rgb(sdm_map *m, sdm_point_xyzrgb *out, size_t cap, size_t *n_out, int32_t flags);
/* device pointer to the per-voxel result array (valid until the next update) */
sdm_status sdm_voxels_device_ptr(sdm_map *m, const sdm_voxel_result **out);

/* ---- owner sets of the object layer: ObjectParticleHashMap (object_layer.h:20-52) */
sdm_status sdm_object_particle_count(sdm_map *m, int32_t track_id, int64_t *count);
/* The keys of ObjectParticleHashMap::indices_map whose sets are not empty: every track id that owns at least one slot of
 * this map (this shard), ascending, at most `cap` of them; *n_out = how many there are.  What the reference's floating-object
 * check iterates over (semantic_dsp_map.h:712-736: an owner set without a tracked object is wiped): an object layer on the
 * C ABI lists these, drops the ids it tracks, and passes the rest to sdm_update as remove_tracks.  Synchronises the map's
 * stream. */
sdm_status sdm_tracks_with_particles(sdm_map *m, int32_t *out, int32_t cap, int32_t *n_out);

/* ---- introspection / checkpoint (tests, fixtures; SURVEY.md §5 checkpoint row) */
sdm_status sdm_get_stats(sdm_map *m, sdm_stats *out, int32_t count_live);
sdm_status sdm_set_profiling(sdm_map *m, int32_t on);
/* test hook: always run the generic 3-D frustum flood instead of the line-graph flood (both are exact) */
sdm_status sdm_debug_force_generic_flood(sdm_map *m, int32_t on);
sdm_status sdm_get_ring_state(sdm_map *m, sdm_ring_state *out);
sdm_status sdm_set_ring_state(sdm_map *m, const sdm_ring_state *in);
sdm_status sdm_get_stamps(sdm_map *m, uint32_t *sx, uint32_t *sy, uint32_t *sz);
sdm_status sdm_set_stamps(sdm_map *m, const uint32_t *sx, const uint32_t *sy, const uint32_t *sz);
/* SoA dump/load of every slot of this shard (V*S entries per array; NULL = skip on dump) */
sdm_status sdm_dump_state(sdm_map *m, float *px, float *py, float *pz, float *w, uint16_t *ts,
                          uint16_t *track, uint8_t *label, uint8_t *status, uint8_t *forget,
                          uint16_t *owner);
sdm_status sdm_load_state(sdm_map *m, const float *px, const float *py, const float *pz,
                          const float *w, const uint16_t *ts, const uint16_t *track,
                          const uint8_t *label, const uint8_t *status, const uint8_t *forget,
                          const uint16_t *owner);
/* diagnostics of the last update: ck+kappa image, per-pixel bin counts, extrinsic */
sdm_status sdm_get_ck_kappa(sdm_map *m, float *out);
sdm_status sdm_get_bin_counts(sdm_map *m, uint32_t *out);
sdm_status sdm_get_bins(sdm_map *m, uint32_t *out, int64_t cap, int64_t *n_out);
sdm_status sdm_get_extrinsic(sdm_map *m, float *out16);

/* ---- timing of single kernels for bench.py's roofline line: runs the occupancy sweep
 * `iters` times on the map's stream bracketed by HIP events, returns average ms per launch. */
sdm_status sdm_time_occupancy_sweep(sdm_map *m, int32_t iters, float *avg_ms);
/* bench hook: overwrite the map with the dense case of SURVEY.md 8(d) (every slot live, every voxel observed) */
sdm_status sdm_debug_fill_dense(sdm_map *m);
/* mode 0 (= sdm_debug_fill_dense): every slot draws one of eight track ids - four to five different ones per voxel, the
 * worst case for the track vote; mode 1: every voxel draws one (a voxel of a real map holds particles of one surface),
 * one voxel in 16 two */
sdm_status sdm_debug_fill_dense_ex(sdm_map *m, int32_t mode);
/* Test hook: how many groups of 512 voxels carry the non-incremental sweep's "every chunk was dense" hint right now, i.e.
 * will be skipped by its classification launch and classified by the evaluating launch itself next time
 * (tests/test_sweep_dense_gpu.py makes sure its repeated sweeps do take that path). */
sdm_status sdm_debug_hinted_groups(sdm_map *m, int64_t *n_out);
/* Test hook.  A non-incremental sweep evaluates the voxels of its sparse chunks either in its first launch or through
 * per-tile lists and a launch of their own; the library picks per sweep from what the sweep before found (speed only:
 * the results are the same).  mode 1 / 0: always / never the lists; -1: the library picks again. */
sdm_status sdm_debug_sweep_lists(sdm_map *m, int32_t mode);
/* Test hook: how the next non-incremental sweep would be issued, from what the last one reported (waits for the map's
 * stream).  Bit 0: the sparse voxels go through per-tile lists; bit 1: every group of 512 voxels was dense, the
 * classification launch is left out and the evaluation launch takes every group (one launch instead of two; the
 * environment variable SDM_SWEEP_SKIP_SCAN=0 keeps both).  Either way every voxel gets the same result. */
sdm_status sdm_debug_sweep_mode(sdm_map *m, int32_t *mode_out);
/* Test hook.  The table of older owner-set memberships (sdm_stats.alias_entries) takes 65536 entries; `cap` (1..65536)
 * makes it report its overflow earlier, so that a test can reach it on a small map.  Call before the map's first frame. */
sdm_status sdm_debug_alias_cap(sdm_map *m, int32_t cap);

/* ---- device-side unit tests of the hand-written primitives (tests/test_primitives_gpu.py) */
sdm_status sdm_test_scan(const uint32_t *in, uint32_t *out, int64_t n);
sdm_status sdm_test_sort_pairs(const uint32_t *keys_in, const uint32_t *vals_in, uint32_t *keys_out,
                               uint32_t *vals_out, int64_t n, int32_t nbits);

const char *sdm_last_error(void);
const char *sdm_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SDM_H_ */
