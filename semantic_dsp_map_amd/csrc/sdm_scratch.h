// sdm_scratch.h — per-frame device work buffers and the stage launchers (internal).
#pragma once
#include "sdm_internal.h"

namespace sdm {

// Raster order of the birth loop (semantic_dsp_map.h:778-800): pass p = 3*row_start + col_start,
// off[p] = first sequence number of the pass, cols[p] = visited columns per row in the pass.
struct BirthOrder {
  int off[10];
  int cols[9];
};

// one moved copy of a particle on its way to its new voxel (operations.h:339-349)
struct MoveCopy {
  float x, y, z;
  uint32_t forget_bits;
  float w;
  uint16_t ts, track, owner;
  uint8_t label, status;
  uint32_t voxel;  // shard-local index of the target voxel
};
static_assert(sizeof(MoveCopy) == 32, "MoveCopy layout");

// object moves / removals (moves.hip)
constexpr int MAX_MOVE_OBJECTS = 48;  // keeps FrameArgs under the 4 KB kernel-argument limit
constexpr int MAX_REMOVE_TRACKS = 128;
constexpr int HALO_OBJ = 64;              // ints per shard in the gathered count matrix (>= MAX_MOVE_OBJECTS)
constexpr int HALO_RECORD_BYTES = 36;
constexpr int HALO_HEADER_BYTES = 16;
// one segment of an export / import buffer: header (record count) + cap records
__host__ __device__ constexpr size_t halo_segment_bytes(uint32_t cap) { return HALO_HEADER_BYTES + (size_t)cap * HALO_RECORD_BYTES; }
struct MoveSet {
  int n;
  float T[MAX_MOVE_OBJECTS][12];  // rows 0..2 of the 4x4 (row-major)
  uint16_t track[MAX_MOVE_OBJECTS];
};

// Everything that changes from frame to frame, in one block of device memory that every kernel of the frame reads its
// frame scalars from (uniform loads).  The host fills a copy and hands it to the frame's first kernel (k_set_frame) BY
// VALUE, which stores it: no host-to-device copy, and - what it is for - a frame whose launch sequence is captured in
// a hipGraph needs exactly one kernel-node parameter update per frame.  Two blocks alternate outside a graph, so that
// the next frame's pose-only chains (frustum, member count) can start while this frame's sweep still runs.
struct FrameArgs {
  Frame f;
  StampUpdates su;
  MoveSet ms;
  int n_obj;           // moving objects of this frame (= ms.n)
  int n_remove;
  uint32_t mv_seq;     // number of frames with moving objects so far: its parity selects the per-object totals (Scratch::mv_tot)
  // A frame with more than MAX_MOVE_OBJECTS moving objects runs member count + k_move_apply once per BATCH of objects (the
  // block's object list is rewritten between them), then ONE k_move_replay: all copies are taken and invalidated before any
  // is re-inserted, and the global ranks run on across the batches (operations.h:321-362).  mv_batch = index of the batch
  // the block holds; batch b starts its ranks where batch b - 1 ended (Counters::n_moved_b).
  // In such a frame the MAIN block is rewritten between the batches (k_move_members_v, launch_set_frame for long removal
  // lists), on the main stream, behind the kernels that read the batch before: afterwards ms / n_obj / mv_seq / mv_batch /
  // n_remove / remove[] hold the LAST batch only.  Nothing behind the move and removal stages reads them (k_move_replay reads
  // n_obj as "did anything move": every batch has objects), and the chains on the side streams read their own copy of the
  // block (fa_side, fa_moves), of which they use the pose, the inputs and the first batch's list.
  uint32_t mv_batch;
  int force_generic;   // test hook: always run the generic 3-D frustum flood
  const float *depth;  // this frame's inputs (device)
  const sdm_labeled_point *cloud;
  uint16_t remove[MAX_REMOVE_TRACKS];
};
static_assert(sizeof(FrameArgs) <= 3400, "FrameArgs travels as a kernel argument (4 KB limit, with the launch's other arguments)");

constexpr int MV_ROW = 16;              // words of a voxel's row of Scratch::mv_row: one 64-byte load
constexpr int MV_DIRECT = MV_ROW - 2;   // ranks held in the row itself (word 0: arrivals - 1; last word: head of the chain of the others)
constexpr uint32_t MV_FIRST = 0x100u;   // MoveCopy::forget_bits: this copy was the first to arrive at its voxel and replays the voxel's list
struct Scratch {
  const FrameArgs *fa = nullptr;       // this frame's block, as the main-stream kernels see it (written by k_frame_begin)
  const FrameArgs *fa_side = nullptr;  // the same for the chains that start before it: frustum, member count (k_set_frame)
  const FrameArgs *fa_moves = nullptr; // the member-count chain's own copy when it runs on its own stream (k_move_members_v: Z-slab shards)
  // frustum vertex bitsets, one line of wpl 64-bit words per (y,z)
  uint64_t *vmask = nullptr, *reach = nullptr;
  int wpl = 0;
  // per-line bitmaps over (y,z), wy 64-bit words per z row: non-empty, overlaps next line in y / z, reached
  uint64_t *line_ne = nullptr, *line_ey = nullptr, *line_ez = nullptr, *line_reach = nullptr;
  int wy = 0;
  // per-pixel bins
  // bin_count: per pixel (H*W + 1 entries), directly behind it row_cnt (H * ROW_SUBS entries), zeroed together at frame
  // begin.  bin_start: W + 1 offsets per image row (the last one is the row's end); the rows' blocks lie in the bin-order
  // arrays in the order the rows' workgroups reserved them (k_bin_rows).
  uint32_t *bin_count = nullptr, *bin_start = nullptr;
  // the visible particles as k_visibility lists them: per image row ROW_SUBS lists of row_cap entries
  // {particle index, column | place in the pixel's bin << ROW_COL_BITS}
  uint32_t *row_cnt = nullptr;  // H * ROW_SUBS counters, ROW_CNT_STRIDE words apart
  uint8_t *row_win = nullptr;   // per pixel: particles its image row puts into a window centred in its column (saturated)
  uint2 *row_list = nullptr;
  uint32_t row_cap = 0;
  uint32_t cap_vis = 0;
  // visible particles in bin order
  uint32_t *bin_idx = nullptr, *vpix = nullptr;
  float4 *vp4 = nullptr;     // x, y, z, weight
  uint32_t *vtf = nullptr;   // track | forget_count << 16
  // per pixel, packed for pass 2 of the weight update: {x, y, z, ck+kappa} and track | is_valid << 16
  float4 *pix4 = nullptr;
  uint32_t *pixt = nullptr;
  float *ck_kappa = nullptr;
  uint8_t *ck_class = nullptr;   // per pixel: CK_DONE / CK_LIGHT / CK_HEAVY (k_ck_classify -> k_ck)
  uint32_t *ck_heavy = nullptr;  // sharded list of pixels with long windows (cap_heavy entries per shard)
  uint32_t cap_heavy = 0;
  // births
  uint32_t *b_valid = nullptr, *b_rank = nullptr;
  uint32_t *bkey_a = nullptr, *bval_a = nullptr, *bkey_b = nullptr, *bval_b = nullptr;
  float4 *bpos = nullptr;
  // object moves
  // Member count of the moving objects (k_move_members, moves.hip), all per listed chunk ("position" = rank of the chunk
  // among the flagged ones, ascending): mv_list[pos] the chunk, mv_cnt[obj * MV_LIST_CAP + pos] its members per moving
  // object, mv_nmem[pos] their number (MV_COMPLEX: the chunk holds older set memberships and is ranked by the apply
  // kernel's alias path), mv_mem[pos * OWNER_CHUNK + i] the members in ascending slot order:
  // slot in chunk | object << 12 | rank among the object's members in this chunk << 18.
  // mv_tot[parity][obj] (one 128-byte line each): members per moving object on this shard, added up with atomics.
  uint32_t *mv_cnt = nullptr;
  uint32_t *mv_list = nullptr, *mv_nlist = nullptr;
  uint32_t *mv_nmem = nullptr, *mv_mem = nullptr, *mv_tot = nullptr;
  MoveCopy *mv_copy = nullptr;     // the moved copies, indexed by global rank (32 B each: two 16-byte accesses)
  uint32_t cap_move = 0;
  // slab-crossing copies: one segment [u32 count, pad to 16 B][halo_cap records] per shard.  Segment d of halo_send holds
  // the copies addressed to shard d, segment s of halo_recv what shard s addressed to this one (all-to-all)
  unsigned char *halo_send = nullptr;
  const unsigned char *halo_recv = nullptr;
  uint32_t halo_cap = 0;
  uint32_t halo_world = 1;
  uint8_t *track_to_obj = nullptr; // 65536 entries: moving-object rank of a track id, 0xFF = not moving
  // generic
  uint32_t *scan_scratch = nullptr, *sort_scratch = nullptr;
  uint32_t *scan_scratch_b = nullptr;   // births run on their own stream
  // re-insertion of the moved copies: per target voxel of the shard a row of MV_ROW words (mv_row) - word 0 the number of
  // copies that arrived, less one (all ones when idle), then the ranks of the first MV_DIRECT arrivals, in the last word the
  // head of the chain of the later ones (all ones when idle; mv_next per rank).  A copy knows its target voxel
  // (MoveCopy::voxel) and whether it arrived first (MV_FIRST): that one replays the list and leaves the row idle
  // (move_link, k_move_replay).
  uint32_t *mv_row = nullptr, *mv_next = nullptr;
  Counters *cnt = nullptr;
  Cursors *cur = nullptr;
};

// N1: arguments of the labeled-point-cloud kernel (same layout as the kernel-side struct)
constexpr int MAX_CLOUD_OBJECTS = 64;
struct CloudArgsHost {
  double R[9], t[3];
  double ifx, icx, ify, icy;
  double dmin, dmax;
  float sigma0, sigma1;
  int consider_depth_noise, consider_instance, n_objects, has_static;
  int sky_instance, has_bbox;  // ZED2 preset (pointcloud_tools.h:174-196, 236-242, 254-272)
  int track[MAX_CLOUD_OBJECTS], label[MAX_CLOUD_OBJECTS];
};
void launch_labeled_cloud(const Dims &d, const CloudArgsHost &h, const float *depth, const uint8_t *static_mask,
                          const uint16_t *label_to_inst, const uint8_t *obj_masks, const double *bbox,
                          sdm_labeled_point *cloud, hipStream_t s);
// manualResize (pointcloud_tools.h:1104-1133): nearest-neighbour reduction of a src_w x src_h image to d.W x d.H
void launch_manual_resize(const Dims &d, const void *src, void *dst, int src_w, int src_h, float scale, int elem_bytes,
                          hipStream_t s);

void launch_set_frame(FrameArgs *fa_dev, const FrameArgs &fa, hipStream_t s);
const void *set_frame_kernel();  // the kernel behind launch_set_frame (hipGraph kernel-node parameter updates)
// k_frame_begin's arguments as one object: the same kernel is launched directly and replayed as a hipGraph node whose
// parameters are replaced every frame
// what the member count of the moving objects (move_members_body, moves.hip) writes and reads besides the map's state
struct MembersArgs {
  uint32_t *mv_cnt, *mv_list, *mv_nlist, *mv_nmem, *mv_mem, *mv_tot;
  Cursors *cur;
  uint32_t n_flags;  // chunks of OWNER_CHUNK slots (= owner_flag bytes in use)
  size_t n_slots;
};
struct FrameBeginLaunch {
#ifndef SDM_FB_GRID
#define SDM_FB_GRID 1024  // (round 6: 512 -> 1024 workgroups halve the chunks a workgroup of the member count ranks one after the other: driven -3.7 us, headline -1 us, tools/gpu_driven_ab.sh)
#endif
  static constexpr unsigned GRID = SDM_FB_GRID, BLOCK = 256;
  Counters *cnt;
  uint32_t *bin_count;
  uint32_t n_bins;
  State st;
  FrameArgs fa;
  FrameArgs *dst_main, *dst_side;
  Dims d;
  uint32_t slab_max;
  MembersArgs ma;
  int with_members;  // the launch is also the member count of the frame's moving objects (whole maps; shards run it as a chain of its own)
  void *argv[11];
  void set(const Dims &d, const State &st, const Scratch &sc, const FrameArgs &fa, bool with_side, bool with_members);
  static const void *kernel();
};
// done: recorded when the kernel has completed (the launch packet's own completion signal: no marker packet behind it)
void launch_frame_begin(FrameBeginLaunch &a, hipStream_t s, hipEvent_t done = nullptr);
void launch_moves_batch(const Dims &d, const State &st, const Scratch &sc, const FrameArgs &fa_batch, int32_t *counts_local, hipStream_t s);
// lists (non-incremental sweeps only): the tiles' sparse voxels go through State::occ_list and a launch of their own
// mode (non-incremental sweeps only): OCC_LISTS | OCC_SKIP_SCAN - every group of the map was hinted when the last such
// sweep ran (State::grp_hint), so the first launch would be workgroups that leave after four bytes: it is left out and
// k_occupancy_dense takes every group as hinted, whatever its byte says by now (right for any map, fast for a dense one)
constexpr int OCC_LISTS = 1, OCC_SKIP_SCAN = 2;
void launch_occupancy(const Dims &d, const Filter &flt, const State &st, Counters *cnt, int all_dirty, const FrameArgs *fa, uint32_t remark,
                      hipStream_t s, int mode = 0);
size_t tile_mark_bytes(const Dims &d);  // State::tile_dirty, padded for the sweep's tile scan
void launch_frustum(const Dims &d, const Scratch &sc, hipStream_t s);
// visibility + binning; its last kernel also classifies the pixels for launch_ck (same ck_out / finish)
void launch_visibility(const Dims &d, const Filter &flt, const State &st, const Scratch &sc, float *ck_out, int finish, hipStream_t s,
                       hipEvent_t vis_done = nullptr);  // vis_done: recorded when k_visibility - the first of the three launches - has completed
void launch_ck(const Dims &d, const Filter &flt, const State &st, const Scratch &sc, float *ck_out, int finish, hipStream_t s);
// part_stride: floats between two partial images (0 = H*W)
void launch_ck_finish(const Dims &d, const Filter &flt, const Scratch &sc, const float *parts, int n_parts, size_t part_stride, hipStream_t s);
void launch_ck_reduce_chunk(const float *stage, const float *own_part, float *full, uint32_t chunk, int world, int rank, hipStream_t s);
// ck_raw: the summed (not yet finished) ck image of a sharded frame, nullptr = pix4 holds ck + kappa
void launch_weight(const Dims &d, const Filter &flt, const State &st, const Scratch &sc, hipStream_t s, const float *ck_raw = nullptr);
int launch_birth_prepare(const Dims &d, const Filter &flt, const BirthOrder &bo, const State &st,
                         const Scratch &sc, hipStream_t s);
void launch_birth_replay(const Dims &d, const Filter &flt, const State &st, const Scratch &sc, int which, bool literal,
                         hipStream_t s);
void launch_count_live(const Dims &d, const State &st, unsigned long long *out, hipStream_t s);
void launch_count_owner(const Dims &d, const State &st, uint16_t track, unsigned long long *out, hipStream_t s);
void launch_pack_pos4(float4 *pos4, uint8_t *forget_plane, const float *px, const float *py, const float *pz, const uint8_t *forget, size_t n,
                      hipStream_t s);
void launch_unpack_pos4(const float4 *pos4, const uint8_t *forget_plane, float *px, float *py, float *pz, uint8_t *forget, size_t n, hipStream_t s);
// the reference's slot-order arrays <-> records + voxel stamps (state export / import; slot 0 = the time particle)
void launch_rec_pack(const Dims &d, const State &st, const float *w, const uint16_t *ts, const uint16_t *track,
                     const uint8_t *label, const uint8_t *status, hipStream_t s);
void launch_rec_unpack(const Dims &d, const State &st, float *w, uint16_t *ts, uint16_t *track, uint8_t *label, uint8_t *status,
                       hipStream_t s);
void launch_fill_dense(const Dims &d, const State &st, uint32_t stamp, int mode, hipStream_t s);
void launch_vflag_from_records(const Dims &d, const State &st, hipStream_t s);
// N2: colour tables on the device (sdm_colour_config + the two division tables of OpenCV's 8-bit RGB2HSV)
struct ColourTables {
  sdm_colour_config cfg;
  int32_t sdiv[256], hdiv180[256];
};
// scratch of the result lists (k_emit_mark / k_emit_write, kernels.hip)
struct EmitScratch {
  uint8_t *mask = nullptr;      // emit_mask_bytes: per thread of k_emit_mark, the voxels it selected
  uint32_t *blk_cnt = nullptr;  // emit_block_elems: selected voxels per workgroup
  uint32_t *total = nullptr;    // the list's length
};
size_t emit_mask_bytes(const Dims &d);
size_t emit_block_elems(const Dims &d);
// the list's length only
void launch_emit_count(const Dims &d, const State &st, const EmitScratch &e, int want_free, hipStream_t s);
void launch_emit_points_rgb(const Dims &d, const Frame &f, const State &st, const ColourTables *ct, const EmitScratch &e,
                            sdm_point_xyzrgb *out, uint32_t cap, int want_free, const float sub[3], hipStream_t s);
void launch_emit_points(const Dims &d, const Frame &f, const State &st, const EmitScratch &e, sdm_point *out, uint32_t cap, int want_free,
                        const float sub[3], int mark_fov, hipStream_t s);

size_t move_blocks(const Dims &d);
size_t move_count_elems();
size_t move_member_elems(size_t n_slots);
size_t move_total_elems();
void launch_tracks_with_particles(const Dims &d, const State &st, uint32_t *bitmap, hipStream_t s);
void launch_owner_flags(const Dims &d, const State &st, hipStream_t s);
void launch_moves_count(const Dims &d, const State &st, const Scratch &sc, int32_t *counts_local, hipStream_t s,
                        const FrameArgs *by_value = nullptr);
void launch_moves_transform(const Dims &d, const Filter &flt, const State &st, const Scratch &sc, const int32_t *counts_all, int world,
                            int rank, hipStream_t s);
void launch_moves_finish(const Dims &d, const Filter &flt, const State &st, const Scratch &sc, int world, int rank, hipStream_t s);
void launch_remove(const Dims &d, const State &st, const Scratch &sc, hipStream_t s);

}  // namespace sdm
