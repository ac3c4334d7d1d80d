// sdm_internal.h — shared declarations of libsdm_hip (not part of the C ABI).
//
// Data layout in HBM (particle index = voxel << p_n | slot, the reference's, mc_ring/operations.h:370,788):
//   dense, per slot     pos4   float4  x, y, z, 0
//                       forget u8      forget count
//                       owner  u16     track id of the owner set holding this index, 0xFFFF = none
//   dense, per voxel    vts    u16     observation stamp (the reference's slot-0 time particle; its only home)
//                       vflag  u8      0 = every slot INVALID
//                       res    8 B     result of the occupancy sweep
//   one record per voxel (10*(S-1) bytes)  weight, time stamp, track, label, status of its S-1 particle slots (SlotRef below)
// plus the three per-axis slab stamp arrays.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sdm.h"

#pragma clang fp contract(off)

namespace sdm {

constexpr uint32_t INVALID_INDEX = 0xffffffffu;  // INVALID_PARTICLE_INDEX, operations.h:35
constexpr uint16_t OWNER_NONE = 0xFFFF;
constexpr int PDF_NUM = 20000;                     // GAUSSIAN_PDF_NUM, basic_algorithms.h:378
#define SDM_OCC_INIT_WEIGHT 0.05f                  // C_PARTICLE_OCC_INIT_WEIGHT, settings.h:147
#define SDM_MIN_RIGHT_PDF 0.1f                     // c_min_rightly_updated_pdf, settings.h:149

// Particle_Status, mc_ring/buffer.h:43-50
enum : uint8_t { ST_INVALID = 0, ST_UPDATED = 1, ST_REGULAR_BORN = 2, ST_GUESSED_BORN = 3, ST_COPIED = 4, ST_TIMEPTC = 5 };

// Everything a kernel needs to know about the grid and the camera (passed by value).
struct Dims {
  int x_n, y_n, z_n, p_n;
  uint32_t NX, NY, NZ, S;
  uint32_t V;            // voxels of the whole map
  uint32_t v_begin;      // first storage voxel index owned by this shard
  uint32_t v_count;      // voxels owned by this shard (Z-slab in ring-index space)
  uint32_t rz_begin, rz_count;
  float voxel_size, recip;
  float pmin[3], pmax[3];
  int W, H;
  float fx, fy, cx, cy;
  float dmin, dmax;
  float tanx, tany;
  float occl_coeff;      // g_depth_error_stddev_at_one_meter + 1.f, operations.h:1387
  int window_half;
  int max_movable;
};

// Ring-buffer state of the current frame (mc_ring/buffer.h:97-120) + frame scalars.
struct Frame {
  int eq[3];
  float center[3];
  uint32_t gts;          // global_time_stamp
  uint32_t epoch;        // sweep epoch 1..255 (State::tile_dirty): the number the next sweep looks for
  float E[16];           // extrinsic, row-major
  int bb0[3], bb1[3];    // conservative map-index bounding box of the frustum, [bb0, bb1) in voxels
  int start_v[3];        // BFS start vertex
  int start_ok;          // start vertex inside the vertex grid
};

// Filter parameters in device-friendly form.
struct Filter {
  float p_detect, noise_number, occ_threshold, id_transition;
  float forget[8];       // getForgettingFactor(count) for count 0..7
  int independent;
  int consider_depth_noise;
  int nb;                // births per valid pixel actually generated
  int use_rng;           // births draw from the noise table
  int noise_n;           // table length
};

// Device-side counters / flags of one frame (zeroed at frame start, except cursors).
struct Counters {
  uint32_t n_vis;
  uint32_t n_birth_success;
  uint32_t n_birth_attempts;
  uint32_t n_resampled;
  uint32_t n_moved;
  uint32_t n_move_reinserted;
  uint32_t n_frustum_voxels;
  uint32_t n_occupied;
  uint32_t n_free;
  uint32_t overflow;
  uint32_t n_valid_px;
  uint32_t n_halo_dropped;  // slab-crossing copies beyond the export capacity of their destination (dropped, SDM_ERR_CAPACITY)
  // the flood flags (below) as THIS frame's visibility pass found them: the frustum chain of the next frame may start -
  // and reset them - while this frame's later stages run, so what the host reads after the frame are these copies
  uint32_t vis_flood_complex, vis_flood_rounds, vis_start_in_frustum;
  uint32_t n_moved_b[2];  // members moved up to and including batch b of this frame's object list, by batch parity (FrameArgs::mv_batch)
  uint32_t pad[2];
  // Atomics on one cache line retire one at a time (~12 ns each on MI355X) - same address or not.  Counters that
  // every wave bumps are therefore sharded by block index, one 128-byte line per shard; the per-shard
  // visible-particle counters also index per-shard regions of the work list.
  struct alignas(128) ShardLine {
    uint32_t vis;       // visible particles appended to this shard's part of the work list
    uint32_t fv;        // frustum voxels handled
    uint32_t heavy;     // weight-update pass 1: pixels handed to the row-parallel kernel
    uint32_t birth;     // successful births
    uint32_t resample;  // voxels resampled
    uint32_t sweep;     // voxels the occupancy sweep evaluated in full (record fetched) in its last launch
    uint32_t sweep_tiles;  // tiles that sweep looked into
    uint32_t heavy_ticket; // weight-update pass 1: next batch of this shard's heavy-pixel list to hand out
    uint32_t pad[24];
  };
  ShardLine shard[64];
  // Written by the frustum chain, which may run ahead of the frame's k_frame_begin (it starts when the previous
  // frame's particles are final): not zeroed with the rest, the chain's first kernel resets them.
  uint32_t flood_complex;  // some x-line of the frustum mask is not one contiguous run
  uint32_t flood_rounds;
  uint32_t start_in_frustum;
  uint32_t flood_pad[29];
};
static_assert(sizeof(Counters::ShardLine) == 128, "one cache line per shard");
constexpr uint32_t VIS_SHARDS = 64;
constexpr uint32_t ROW_SUBS = 8;  // sub-lists per image row of the visible particles (Scratch::row_list)
// a row-list entry's second word: column | place in the pixel's bin << ROW_COL_BITS (images up to 4095 pixels wide - round 6; 2047
// before -, up to 2^20 particles in one pixel's bin)
constexpr int ROW_COL_BITS = 12, ROW_PIB_BITS = 32 - ROW_COL_BITS;
constexpr uint32_t ROW_CNT_STRIDE = 32;  // uint32 per sub-list counter: a cache line each (atomics on one line retire one at a time)
constexpr uint32_t OWNER_CHUNK = 4096;  // slots per owner_flag byte (= slots per block of the move sweep)
constexpr uint32_t OWNER_GROUP = 64;    // chunks per owner_flag2 byte
// sizes of the two flag levels for n_slots slots, padded so that the member count reads them in whole 16-byte pieces
// without range checks: the fine level to whole groups, the coarse level to whole tiles of 4096 groups
__host__ __device__ constexpr size_t owner_flag_bytes(size_t n_slots) {
  return ((n_slots + OWNER_CHUNK - 1) / OWNER_CHUNK + OWNER_GROUP - 1) / OWNER_GROUP * OWNER_GROUP;
}
__host__ __device__ constexpr size_t owner_flag2_bytes(size_t n_slots) { return (owner_flag_bytes(n_slots) / OWNER_GROUP + 4095) / 4096 * 4096; }
// Slab stamps written by this frame's ring shift (mc_ring/operations.h:1131-1181), applied on the device by the
// frame-begin kernel; all stamps of one frame carry the same value (the frame's global_time_stamp).
constexpr int MAX_STAMP_UPDATES = 96;
struct StampUpdates {
  int n;
  uint32_t value;
  uint16_t entry[MAX_STAMP_UPDATES];  // axis << 12 | ring index
};
struct Cursors {
  int32_t birth_cursor;
  int32_t move_cursor;
  // set by the member count of the move stage when its chunk list overflows; that count may run before the frame's
  // counters are zeroed (it is started as soon as the previous frame's particles are final), so it cannot use them
  uint32_t move_list_overflow;
};

// Slot attributes that are only touched where a particle lives - weight, time stamp, track id, label, status - share
// one record per voxel.  Slot 0 of a voxel is its time particle (mc_ring/buffer.h:57-79, operations.h:824-837): of its
// fields the map only ever needs the time stamp, which lives in the dense array State::vts; the record holds the
// L = S - 1 PARTICLE slots and nothing else,
//   [w: 4L | ts: 2L | track: 2L | label: L | status: L]        10 L bytes (70 B at S = 8), records back to back.
// These are exactly the (S - 1) * 10 bytes per voxel SURVEY.md 8(d) counts for the occupancy sweep, so a sweep over a
// dense map moves the algorithmic bytes, the 2-byte stamp, the 8-byte result and one flag byte: 81 B/voxel (rounds 2-5
// kept a slot-0 row in the record - 80 B records, 92 B/voxel moved, 1.15 x the survey's figure).  A record is 2-byte
// aligned and no more: every access below is written as an under-aligned access (gfx950 runs with unaligned access mode
// on for global memory and LDS; a chunk of 64 records - 640 L bytes - starts on a 128-byte line, which is what the
// chunk-wide loads of the non-incremental sweep go by).  sdm_dump_state reports slot 0 as {TIMEPTC, w 0, ts = the
// voxel's observation stamp, track 0, label 0}, which is all the reference's slot 0 ever holds; sdm_load_state takes
// the stamp from slot 0 of the stamp array and ignores its other fields.
// What whole-map sweeps stream stays dense: the per-voxel stamp and "something here" flag, owner, positions.
constexpr size_t REC_BYTES_PER_SLOT = 10;
__host__ __device__ constexpr uint32_t rec_bytes(uint32_t S) { return (uint32_t)REC_BYTES_PER_SLOT * (S - 1u); }
// bytes the record array needs for v_count voxels: one chunk of padding (the sweep's chunk-wide loads need no clamp at
// the map's end) and a piece more for the whole-record fetch, which reads up to the next multiple of 8 bytes
__host__ __device__ constexpr size_t rec_array_bytes(size_t v_count, uint32_t S) { return (v_count + 64) * rec_bytes(S) + 64; }
typedef float rec_f32 __attribute__((aligned(2)));  // a float at a 2-byte aligned address (typed: not a char access)
typedef uint32_t rec_v4u __attribute__((ext_vector_type(4), aligned(2)));
typedef uint32_t rec_v2u __attribute__((ext_vector_type(2), aligned(2)));

// One particle slot of a voxel's record: r = the record, L = S - 1, k = slot - 1 (slot 1..S-1; slot 0 has no entry).
struct SlotRef {
  unsigned char *r;
  uint32_t L, k;
  __host__ __device__ __forceinline__ float w() const { return *reinterpret_cast<const rec_f32 *>(r + 4u * k); }
  __host__ __device__ __forceinline__ uint16_t ts() const { return *reinterpret_cast<const uint16_t *>(r + 4u * L + 2u * k); }
  __host__ __device__ __forceinline__ uint16_t track() const { return *reinterpret_cast<const uint16_t *>(r + 6u * L + 2u * k); }
  __host__ __device__ __forceinline__ uint8_t label() const { return r[8u * L + k]; }
  __host__ __device__ __forceinline__ uint8_t status() const { return r[9u * L + k]; }
  __host__ __device__ __forceinline__ void set_w(float v) const { *reinterpret_cast<rec_f32 *>(r + 4u * k) = v; }
  __host__ __device__ __forceinline__ void set_ts(uint16_t v) const { *reinterpret_cast<uint16_t *>(r + 4u * L + 2u * k) = v; }
  __host__ __device__ __forceinline__ void set_track(uint16_t v) const { *reinterpret_cast<uint16_t *>(r + 6u * L + 2u * k) = v; }
  __host__ __device__ __forceinline__ void set_label(uint8_t v) const { r[8u * L + k] = v; }
  __host__ __device__ __forceinline__ void set_status(uint8_t v) const { r[9u * L + k] = v; }
};

struct State {
  float4 *pos4 = nullptr;        // x, y, z, 0
  // forget count per slot.  Its own byte plane since round 5 (rounds 1-4: the positions' fourth word): clear() resets the
  // positions and leaves the forget counts alone (operations.h:697-722), so k_clear_map STORES the positions without
  // reading them; the weight update changes a byte instead of reading and rewriting a position.
  uint8_t *forget = nullptr;
  unsigned char *rec = nullptr;  // v_count records of rec_bytes(S) = 10 (S - 1) bytes
  // observation stamp of every voxel = time stamp of its slot-0 "time particle" (operations.h:824-837), kept as a
  // dense array of its own: the sweeps read it for every voxel, the slot rows only where something lives
  uint16_t *vts = nullptr;
  // one byte per voxel.  Bits 0-1: VF_EMPTY = every slot is INVALID; VF_CLEAN = the voxel holds something and nothing
  // it holds has changed since the occupancy sweep last evaluated it (its result stands); VF_DIRTY = it holds
  // something that was written since.  Every kernel that writes a slot, and a ring shift that re-stamps the voxel's
  // slab, sets VF_DIRTY; the sweep sets VF_CLEAN / VF_EMPTY.  Bits 2-3, kept by the sweep: what the voxel's result
  // entry holds right now - VR_UNOBSERVED, VR_EMPTY (the two constant results) or 0 = something else.  Together they
  // let the sweep finish an unobserved, empty or unchanged voxel from 3 bytes read and nothing written.
  uint8_t *vflag = nullptr;
  uint8_t *tile_dirty = nullptr;  // per tile of 2^TILE_SHIFT voxels: the sweep epoch that has to look into it (mark_tile); two arrays
  uint32_t tile_stride = 0;       // bytes of one of the two
  // per chunk of 64 voxels: the voxels the non-incremental sweep's first launch left to its second (dense chunks)
  unsigned long long *occ_need = nullptr;
  // The voxels of a tile's sparse chunks that hold something, listed by the non-incremental sweep's first launch (index
  // inside the tile, OCC_LIST_CAP entries per tile, occ_list_n of them) for the launch behind it, which evaluates them
  // one workgroup per unit of OCC_LIST_UNIT entries, all units at once.  A unit is a (tile, first entry) pair in one of
  // OCC_LIST_SHARDS arrays (occ_unit_cap pairs each), its slot handed out by the shard's counter, which also counts the
  // tiles that listed anything; the sweep's last launch zeroes the counters and turns the tile count into
  // occ_shard[OCC_LIST_SHARDS].word, the word the host reads before the next non-incremental sweep: 2 = few tiles list
  // anything (surfaces), the lists pay; 1 = none or most do, evaluate in the first launch (map.hip, sweep_lists).
  uint16_t *occ_list = nullptr;
  uint32_t *occ_list_n = nullptr;
  uint2 *occ_unit = nullptr;
  uint32_t occ_unit_cap = 0;
  struct alignas(128) OccListShard {
    unsigned long long word;  // low half: units handed out; high half: tiles that listed anything (one atomic for both)
    // entry OCC_LIST_SHARDS only (the sweep's word to the host, sweep_mode_latch): aux[0] - k_occupancy_scan entered a
    // tile (some group of the map was not hinted); aux[1] - 2: the sweep's first launch had nothing to do, the next
    // non-incremental sweep may leave it out (launch_occupancy, OCC_SKIP_SCAN); 1: it may not
    uint32_t aux[30];
  };
  OccListShard *occ_shard = nullptr;
  // per group of 512 voxels (what one wave of the non-incremental sweep's kernels takes): 1 = every chunk of the group
  // was dense in the last non-incremental sweep; the next one leaves the group to the second launch whole
  // (k_occupancy_scan skips it, k_occupancy_dense classifies it itself).  A hint about speed only: both launches read
  // the same bytes, and either way every voxel gets the same result.
  uint8_t *grp_hint = nullptr;
  uint16_t *owner = nullptr;
  // one byte per OWNER_CHUNK consecutive slots: 1 if any slot of the chunk may have an owner.  Lets the object-move
  // and removal sweeps skip the (vast) part of the map no dynamic object ever touched.
  uint8_t *owner_flag = nullptr;
  // one byte per OWNER_GROUP consecutive chunks: 1 if any of their owner_flag bytes may be set.  The member count of the
  // move stage finds the flagged chunks of a map of any size from a few hundred bytes (every workgroup reads this level
  // whole, then the 64 flag bytes of the groups that are marked).  Both levels are plain byte stores by whoever gives a
  // slot an owner (flag_owner_chunk); cleared by the member count when it finds a chunk / a group without owners.
  uint8_t *owner_flag2 = nullptr;
  // Extra memberships: the reference's owner sets (object_layer.h:20-52) are real sets, and a slot can sit in two of
  // them - a stale index of object A whose slot is taken by a particle of object B stays in A's set until A moves or
  // is removed.  owner[] holds the latest owner; every older membership that the reference still has is an entry
  // (slot index, track) here.  alias[0] = number of entries (deleted ones carry track OWNER_NONE until the next
  // compaction), entries from alias[2]: index, track.  Nearly always empty.
  uint32_t *alias = nullptr;
  uint32_t alias_cap = 0;  // entries the table takes before it reports an overflow (ALIAS_CAP unless a test lowered it)
  // 65536 bits, one per hash of a slot index: set when an entry for the slot goes into `alias`, rebuilt by the table's
  // garbage collection.  Every insertion into / removal from an owner set has to look for older memberships of its slot;
  // with the bit clear there is none and the table is not walked (a drive with a dozen objects keeps a few hundred entries
  // alive, and every one of 20 k re-insertions per frame walked them all: alias_may_hold).
  uint32_t *alias_filter = nullptr;
  sdm_voxel_result *res = nullptr;
  uint32_t *stamps_x = nullptr, *stamps_y = nullptr, *stamps_z = nullptr;
  float *pdf = nullptr;
  float *noise = nullptr;
};

// one byte per tile of 2^TILE_SHIFT voxels (State::tile_dirty): set by whoever sets VF_DIRTY on a voxel of the tile or
// changes its observation stamp; the sweep returns at once from a tile whose byte is 0 and clears the byte otherwise
constexpr int TILE_SHIFT = 11;
constexpr uint32_t OCC_LIST_CAP = 2048, OCC_LIST_UNIT = 256, OCC_LIST_SHARDS = 64;  // State::occ_list (kernels.hip: 32 sparse chunks of fewer than 64 voxels each)
__host__ __device__ constexpr size_t occ_list_tiles(size_t v_count) { return (v_count + ((size_t)1 << TILE_SHIFT) - 1) >> TILE_SHIFT; }
// bytes of State::grp_hint for v_count voxels: whole tiles (4 groups), so that a workgroup reads its four bytes as one word
__host__ __device__ constexpr size_t grp_hint_bytes(size_t v_count) { return (((v_count + (1u << 11) - 1) >> 11) * 4 + 15) / 16 * 16; }  // (whole 16-byte pieces)
enum : uint8_t { VF_EMPTY = 0, VF_CLEAN = 1, VF_DIRTY = 2, VF_STATE = 3, VR_UNOBSERVED = 1 << 2, VR_EMPTY = 2 << 2, VR_MASK = 3 << 2 };
// State::tile_dirty holds, per tile, the sweep epoch in which something in the tile was last written or stamped: the
// frame's kernels mark with the epoch of the frame's sweep (Frame::epoch), the sweep looks for exactly that number and
// marks what it has to see again with the next one.  There are two arrays, odd and even epochs: while a sweep reads the
// marks of its epoch - every workgroup of k_occupancy reads all of them - what it has to see again goes into the other
// array, and so do the next frame's marks; nothing is ever cleared.  Epochs run 1..254 and start over (0 = never marked;
// an even count, so that consecutive epochs always differ in parity): a mark left from 254 sweeps ago sends the sweep
// through one tile in which it then finds nothing to do.
__device__ __host__ __forceinline__ uint32_t next_epoch(uint32_t e) { return e % 254u + 1u; }
__device__ __forceinline__ uint8_t *tile_marks(const State &st, uint32_t epoch) { return st.tile_dirty + (epoch & 1u) * st.tile_stride; }
__device__ __forceinline__ void mark_tile(const State &st, size_t lv, uint32_t epoch) { tile_marks(st, epoch)[lv >> TILE_SHIFT] = (uint8_t)epoch; }
constexpr uint32_t ALIAS_CAP = 65536;  // entries allocated (State::alias_cap of them in use: sdm_debug_alias_cap lowers it for the overflow tests)
constexpr uint32_t ALIAS_FILTER_WORDS = 2048;
__device__ __forceinline__ uint32_t alias_hash(size_t li) { return ((uint32_t)li * 2654435761u) >> 16; }
// may the table hold an entry for slot li?  (read past the vector L1: the bit may have been set by this very kernel)
__device__ __forceinline__ bool alias_may_hold(const State &st, size_t li) {
  const uint32_t h = alias_hash(li);
  return (__hip_atomic_load(st.alias_filter + (h >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (h & 31u)) & 1u;
}
__device__ __forceinline__ void alias_note(const State &st, size_t li) {
  const uint32_t h = alias_hash(li);
  atomicOr(st.alias_filter + (h >> 5), 1u << (h & 31u));
}
// the slot with shard-local index li has (or may have) an owner: both levels of the chunk flags
__device__ __forceinline__ void flag_owner_chunk(const State &st, size_t li) {
  const size_t c = li / OWNER_CHUNK;
  st.owner_flag[c] = 1;
  st.owner_flag2[c / OWNER_GROUP] = 1;
}
// position of the n-th (0-based) set bit of m
__device__ __forceinline__ int nth_set_bit(unsigned long long m, uint32_t n) {
  int pos = 0;
#pragma unroll
  for (int w = 32; w >= 1; w >>= 1) {
    const uint32_t c = (uint32_t)__popcll(m & ((1ull << w) - 1ull));
    if (n >= c) {
      n -= c;
      m >>= w;
      pos += w;
    }
  }
  return pos;
}

// The entries (li, track) among the table's first n go (track = OWNER_NONE: deleted; alias_compact_wave collects them).
// Sixteen entries are requested before the first is looked at, and nothing is stored inside the walk: written as
// `if (entry matches) entry.track = NONE` per entry, every load came after a store that might have hit it, and the walk
// was one round trip PER ENTRY - a table of a hundred older memberships cost an insertion of k_move_replay 10-20 us, the
// 100-250 us spikes of that kernel in the middle of the `driven` drive (tools/probes/timers_moves.py, SDM_TIMERS_FROM).
// (Not inlined: it sits behind every insertion of the replays, whose loops are unrolled - fourteen copies of it in
// k_move_replay alone doubled the code objects.  The table comes as a pointer to GLOBAL memory, by value: through a
// reference to State a function that is not inlined sees a generic pointer and loads through FLAT instructions.)
__device__ __attribute__((noinline)) void alias_drop_global(__attribute__((address_space(1))) uint32_t *alias, uint32_t n, uint32_t li, uint32_t track) {
  const __attribute__((address_space(1))) unsigned long long *e = (const __attribute__((address_space(1))) unsigned long long *)(alias + 2);  // (8-byte aligned: the table starts at word 2; low word = slot, high word = track)
  uint32_t hits = 0, last = 0;
  for (uint32_t k0 = 0; k0 < n; k0 += 16) {
    unsigned long long v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = e[k0 + u < n ? k0 + u : n - 1];
    const unsigned long long want = ((unsigned long long)track << 32) | li;
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (k0 + u < n && v[u] == want) {
        ++hits;
        last = k0 + u;
      }
  }
  if (hits == 1) {
    alias[3 + 2 * last] = OWNER_NONE;
  } else if (hits > 1) {  // (a set holds an index once: not expected - the plain walk takes them all)
    for (uint32_t k = 0; k < n; ++k)
      if (alias[2 + 2 * k] == li && alias[3 + 2 * k] == track) alias[3 + 2 * k] = OWNER_NONE;
  }
}
__device__ __forceinline__ void alias_drop(const State &st, uint32_t n, uint32_t li, uint32_t track) {
  alias_drop_global((__attribute__((address_space(1))) uint32_t *)st.alias, n, li, track);
}

// ObjectParticleHashMap::addParticleToObj (object_layer.h:31-33): slot li joins track's set.  li is the shard-local slot
// index.  Returns false when the alias table is full.
__device__ __forceinline__ bool owner_insert(const State &st, size_t li, uint16_t track) {
  const uint16_t prev = st.owner[li];
  st.owner[li] = track;
  uint32_t n = st.alias[0];
  if (n > st.alias_cap) n = st.alias_cap;
  if (n && alias_may_hold(st, li))
    alias_drop(st, n, (uint32_t)li, track);  // a set holds an index once
  if (prev == OWNER_NONE || prev == track) return true;
  const uint32_t k = atomicAdd(&st.alias[0], 1u);  // prev's set keeps the index
  if (k >= st.alias_cap) return false;
  st.alias[2 + 2 * k] = (uint32_t)li;
  st.alias[3 + 2 * k] = prev;
  alias_note(st, li);
  return true;
}
// removeParticleFromObj (object_layer.h:35-37): slot li leaves track's set
__device__ __forceinline__ void owner_erase(const State &st, size_t li, uint16_t track) {
  if (st.owner[li] == track) {
    st.owner[li] = OWNER_NONE;
    return;
  }
  uint32_t n = st.alias[0];
  if (n > st.alias_cap) n = st.alias_cap;
  if (n && alias_may_hold(st, li))
    alias_drop(st, n, (uint32_t)li, track);
}

// The same two for a kernel in which ONE thread owns all slots of a voxel (the ordered replays): `own` is the thread's
// register copy of the slot's owner entry (loaded with the voxel's other rows; the replay works on the register copy of the
// whole voxel and stores its rows once, at the end - round 6: a replay that stored ten fields per insertion ran into the 64
// memory operations a wave may have in flight, and its longest voxels paid a microsecond per insertion for it), n_alias the table length read once at
// kernel start, touched = this thread has added an entry since.  Entries other threads add meanwhile concern other
// voxels' slots and cannot match li, so the length read at the start serves until this thread adds one itself.  What the
// generic versions load between the stores of one insertion and the next - the owner entry, the table length: two
// dependent round trips per insertion - is already there.
// fbits: the filter bits of the voxel's slots (bit i = the table may hold an entry for slot i), fetched for all slots in
// ONE round before the replay starts (alias_filter_bits) and kept up to date by this thread's own additions.  Read per
// insertion - alias_may_hold, a load past the L1 - it was a dependent round trip in front of every insertion of a voxel's
// replay: seventeen in a row for the longest list of a frame of the `driven` workload, 20 of that head's 30 us.  Nobody
// else can add an entry for one of this voxel's slots while the kernel runs, and a bit another voxel's slot shares with
// one of these only ever costs a walk.
template <int S>
__device__ __forceinline__ uint32_t alias_filter_bits(const State &st, size_t base) {
  uint32_t w[S], bits = 0;
#pragma unroll
  for (int i = 1; i < S; ++i) w[i] = __hip_atomic_load(st.alias_filter + (alias_hash(base + i) >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int i = 1; i < S; ++i) bits |= ((w[i] >> (alias_hash(base + i) & 31u)) & 1u) << i;
  return bits;
}
__device__ __forceinline__ bool owner_insert_local(const State &st, size_t li, uint16_t track, uint16_t &own, uint32_t n_alias,
                                                   bool &touched, uint32_t &fbits, int slot) {
  const uint16_t prev = own;
  own = track;  // (the caller stores the voxel's owner row once, when its replay is through)
  if ((fbits >> slot) & 1u) {
    uint32_t n = touched ? st.alias[0] : n_alias;
    if (n > st.alias_cap) n = st.alias_cap;
    alias_drop(st, n, (uint32_t)li, track);  // a set holds an index once
  }
  if (prev == OWNER_NONE || prev == track) return true;
  const uint32_t k = atomicAdd(&st.alias[0], 1u);  // prev's set keeps the index
  touched = true;
  if (k >= st.alias_cap) return false;
  st.alias[2 + 2 * k] = (uint32_t)li;
  st.alias[3 + 2 * k] = prev;
  alias_note(st, li);
  fbits |= 1u << slot;
  return true;
}
__device__ __forceinline__ void owner_erase_local(const State &st, size_t li, uint16_t track, uint16_t &own, uint32_t n_alias,
                                                  bool touched, uint32_t fbits, int slot) {
  if (own == track) {
    own = OWNER_NONE;  // (stored with the voxel's owner row by the caller)
    return;
  }
  if ((fbits >> slot) & 1u) {
    uint32_t n = touched ? st.alias[0] : n_alias;
    if (n > st.alias_cap) n = st.alias_cap;
    alias_drop(st, n, (uint32_t)li, track);
  }
}

// Garbage collection of the table of older memberships: deleted entries (track OWNER_NONE) go, the live ones keep their
// order.  Called by ONE wave of a kernel that runs while nobody else touches the table (k_bin_rows: after the frame's
// moves and removals, before its births).  Batches of 64 entries: a batch is read whole before its survivors are
// written, and they land at or below the batch's own positions.
__device__ __forceinline__ void alias_compact_wave(const State &st) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t na = st.alias[0];
  if (na == 0) return;
  if (na > st.alias_cap) {
    // entries beyond the table's capacity were dropped: the sets are incomplete from here on.  alias[1] keeps saying so
    // (SDM_ERR_CAPACITY at every synchronisation, sdm_stats.alias_overflowed) until sdm_clear / sdm_load_state - this
    // very function used to erase the only trace of it by writing the clamped count back.
    if (lane == 0) st.alias[1] = 1u;
    na = st.alias_cap;
  }
  for (uint32_t k = lane; k < ALIAS_FILTER_WORDS; k += 64) st.alias_filter[k] = 0u;  // rebuilt from the survivors below
  __threadfence();
  uint32_t keep = 0;
  for (uint32_t b0 = 0; b0 < na; b0 += 64) {
    const uint32_t k = b0 + lane;
    uint32_t idx = INVALID_INDEX, trk = OWNER_NONE;
    if (k < na) {
      idx = st.alias[2 + 2 * k];
      trk = st.alias[3 + 2 * k];
    }
    const bool live = trk != OWNER_NONE;
    const unsigned long long m = __ballot(live);
    if (live) {
      const uint32_t at = keep + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      st.alias[2 + 2 * at] = idx;
      st.alias[3 + 2 * at] = trk;
      alias_note(st, idx);
    }
    keep += (uint32_t)__popcll(m);
  }
  for (uint32_t k = keep + lane; k < na; k += 64) {
    st.alias[2 + 2 * k] = INVALID_INDEX;
    st.alias[3 + 2 * k] = OWNER_NONE;
  }
  if (lane == 0) st.alias[0] = keep;
}

// the record of local voxel lv / slot `slot` (1..S-1) of it / the slot with shard-local index li = lv << p_n | slot
__host__ __device__ __forceinline__ unsigned char *rec_ptr(const State &st, uint32_t S, size_t lv) { return st.rec + lv * rec_bytes(S); }
__host__ __device__ __forceinline__ SlotRef slot_ref(const State &st, uint32_t S, size_t lv, uint32_t slot) {
  return SlotRef{rec_ptr(st, S, lv), S - 1u, slot - 1u};
}
__host__ __device__ __forceinline__ SlotRef slot_ref_li(const State &st, int p_n, size_t li) {
  const uint32_t S = 1u << p_n;
  return SlotRef{rec_ptr(st, S, li >> p_n), S - 1u, ((uint32_t)li & (S - 1u)) - 1u};
}

// ---- device helpers ------------------------------------------------------------------
// PINNED: 4x4 row-major times [x y z 1] evaluated ((m0*x + m1*y) + m2*z) + m3
__host__ __device__ __forceinline__ float row4(const float *r, float x, float y, float z) {
  return ((r[0] * x + r[1] * y) + r[2] * z) + r[3];
}

// operations.h:1037-1070 (single +-N correction)
__host__ __device__ __forceinline__ uint32_t axis_correct(int idx, uint32_t n) {
  if (idx < 0) return (uint32_t)(idx + (int)n);
  if (idx >= (int)n) return (uint32_t)(idx - (int)n);
  return (uint32_t)idx;
}

// operations.h:890-923 row-major storage index
__host__ __device__ __forceinline__ uint32_t ring_to_voxel(const Dims &d, uint32_t rx, uint32_t ry, uint32_t rz) {
  return (((rz << d.y_n) | ry) << d.x_n) | rx;
}
__host__ __device__ __forceinline__ void voxel_to_ring(const Dims &d, uint32_t v, uint32_t &rx, uint32_t &ry, uint32_t &rz) {
  rx = v & (d.NX - 1);
  ry = (v >> d.x_n) & (d.NY - 1);
  rz = (v >> (d.x_n + d.y_n)) & (d.NZ - 1);
}

// PINNED float -> index cast (operations.h:867-869): (-1,0) truncates to 0 and is accepted.
__host__ __device__ __forceinline__ bool float_to_idx(float f, uint32_t n, uint32_t &out) {
  if (!(f > -1.0f && f < (float)n)) return false;
  out = (uint32_t)(int32_t)f;
  return out < n;
}

// operations.h:849-883: global position -> storage voxel index (INVALID_INDEX if outside the map)
__host__ __device__ __forceinline__ uint32_t global_pos_to_voxel(const Dims &d, const Frame &f, float px, float py,
                                                                 float pz, uint32_t &rx, uint32_t &ry, uint32_t &rz) {
  float mx = px - f.center[0], my = py - f.center[1], mz = pz - f.center[2];
  uint32_t ix = 0, iy = 0, iz = 0;
  bool ok = float_to_idx((mx - d.pmin[0]) * d.recip, d.NX, ix);
  ok = float_to_idx((my - d.pmin[1]) * d.recip, d.NY, iy) && ok;
  ok = float_to_idx((mz - d.pmin[2]) * d.recip, d.NZ, iz) && ok;
  if (!ok) return INVALID_INDEX;
  rx = axis_correct((int)ix + f.eq[0], d.NX);
  ry = axis_correct((int)iy + f.eq[1], d.NY);
  rz = axis_correct((int)iz + f.eq[2], d.NZ);
  return ring_to_voxel(d, rx, ry, rz);
}

// operations.h:1267-1290 (PINNED: K*p/z as (fx*x + cx*z)/z, (fy*y + cy*z)/z)
__device__ __forceinline__ bool project_to_image(const Dims &d, const Frame &f, float px, float py, float pz, int &row,
                                                 int &col, float &cam_z) {
  float x = row4(f.E + 0, px, py, pz);
  float y = row4(f.E + 4, px, py, pz);
  float z = row4(f.E + 8, px, py, pz);
  if (z < d.dmin || z > d.dmax) return false;
  float u = (d.fx * x + d.cx * z) / z;
  float v = (d.fy * y + d.cy * z) / z;
  row = (int)v;
  col = (int)u;
  if (row < 0 || row >= d.H || col < 0 || col >= d.W) return false;
  cam_z = z;
  return true;
}

// operations.h:1240-1258
__device__ __forceinline__ bool point_in_frustum(const Dims &d, const Frame &f, float px, float py, float pz) {
  float x = row4(f.E + 0, px, py, pz);
  float y = row4(f.E + 4, px, py, pz);
  float z = row4(f.E + 8, px, py, pz);
  if (z < d.dmin || z > d.dmax) return false;
  if (fabsf(x) > z * d.tanx) return false;
  if (fabsf(y) > z * d.tany) return false;
  return true;
}

// basic_algorithms.h:417-422 (PINNED: NaN -> 1e-9f)
__device__ __forceinline__ float query_pdf(const float *__restrict__ pdf, float x, float mu, float sigma) {
  float c = (x - mu) / sigma;
  if (!(c <= 9.9f && c >= -9.9f)) return 1e-9f;
  return pdf[(int)(c * 1000 + 10000)];
}

// The weight update divides three coordinates by the same sigma for every particle-pixel pair, and its kernels are
// bound by instruction issue (profiles/r02_a7_sq.json).  The compiler's IEEE division is v_div_scale x 2, v_rcp, two
// Newton steps on the reciprocal, the quotient with two residual corrections, v_div_fmas, v_div_fixup - 11
// instructions, of which the scaling and the fix-up only matter at extreme exponents and the reciprocal depends on the
// denominator alone.  div_recip + div_by are the same arithmetic without them: the reciprocal once per sigma, five
// instructions per quotient.  tools/probes/div_probe.hip compares the bits with a / b for EVERY float b in
// [2^-10, 2^10] and numerators 0, 1e-7, around 9.9 b and with exponents -40 .. 20 (1.3e9 pairs, no mismatch apart from
// -0 / b = -0 coming out as +0, which selects the same table entry).  Outside that range of sigma div_recip returns 0
// and the plain division is used.  A numerator beyond the probed exponents cannot select a different entry either:
// smaller ones land on the table's centre with any rounding, larger ones (or an overflow to NaN) outside its range.
__device__ __forceinline__ float div_recip(float b) {
  if (!(b >= 0.0009765625f && b <= 1024.f)) return 0.f;
  const float r0 = __builtin_amdgcn_rcpf(b);
  const float e0 = __builtin_fmaf(-b, r0, 1.0f);
  return __builtin_fmaf(e0, r0, r0);
}
__device__ __forceinline__ float div_by(float a, float b, float r) {
  const float q0 = a * r;
  const float e1 = __builtin_fmaf(-b, q0, a);
  const float q1 = __builtin_fmaf(e1, r, q0);
  const float e2 = __builtin_fmaf(-b, q1, a);
  return __builtin_fmaf(e2, r, q1);
}
// query_pdf with the reciprocal of sigma from div_recip.  FAST = the caller has checked, for the whole wave, that no
// lane's reciprocal is 0 (a test per quotient costs the scalar unit what the shorter division saves the vector unit).
template <bool FAST>
__device__ __forceinline__ float query_pdf_r(const float *__restrict__ pdf, float x, float mu, float sigma, float rsig) {
  const float c = FAST ? div_by(x - mu, sigma, rsig) : (x - mu) / sigma;
  if (!(fabsf(c) <= 9.9f)) return 1e-9f;
  return pdf[(uint32_t)(int)(c * 1000 + 10000)];  // 100 .. 19900: an unsigned offset spares the 64-bit address arithmetic
}

typedef uint32_t sdm_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_result(sdm_voxel_result *dst, const sdm_voxel_result &r) {
  sdm_v2u v;
  __builtin_memcpy(&v, &r, 8);
  __builtin_nontemporal_store(v, reinterpret_cast<sdm_v2u *>(dst));  // written once, read by the next stage only
}


// a ring shift re-stamped these slabs: what the voxels there hold has just become stale (operations.h:1131-1181), so
// their results turn into "unobserved" although nobody wrote to them.  That is all the sweep would do for such a voxel
// (isVoxelValid fails: its observation stamp is older than the new slab stamp), so it is done right here, voxel by
// voxel, and the tile needs no mark: an x shift touches one voxel of every x row, i.e. every tile of the map, and
// would otherwise send the next sweep through all of them.  Should the visibility pass observe the voxel again in
// this very frame, it marks the tile itself.  t = update k * slab_max + j-th voxel of its slab.
__device__ __forceinline__ void mark_slab_voxel_dirty(const Dims &d, const State &st, const StampUpdates &su, uint32_t slab_max,
                                                      uint32_t t) {
  const uint32_t k = t / slab_max, j = t - k * slab_max;
  if ((int)k >= su.n) return;
  const uint32_t e = su.entry[k], axis = e >> 12, idx = e & 0xfffu;
  uint32_t rx, ry, rz;
  if (axis == 0) {
    if (j >= d.NY * d.NZ) return;
    rx = idx;
    ry = j % d.NY;
    rz = j / d.NY;
  } else if (axis == 1) {
    if (j >= d.NX * d.NZ) return;
    ry = idx;
    rx = j % d.NX;
    rz = j / d.NX;
  } else {
    if (j >= d.NX * d.NY) return;
    rz = idx;
    rx = j % d.NX;
    ry = j / d.NX;
  }
  if (rz < d.rz_begin || rz >= d.rz_begin + d.rz_count) return;  // another shard's slab
  const uint32_t lv = ring_to_voxel(d, rx, ry, rz) - d.v_begin;
  const uint8_t fl = st.vflag[lv];
  const uint8_t state = fl & VF_STATE;
  // a CLEAN voxel's stored result is gone with this: it is evaluated again when the voxel is seen again
  const uint8_t nf = (uint8_t)((state == VF_CLEAN ? VF_DIRTY : state) | VR_UNOBSERVED);
  if (nf != fl) st.vflag[lv] = nf;
  if ((fl & VR_MASK) != VR_UNOBSERVED) {
    sdm_voxel_result out;
    out.wsum = -1.f;
    out.track = 0;
    out.label = 0;
    out.occ = -1;
    store_result(st.res + lv, out);
  }
}


// ---- primitives (primitives.hip) ------------------------------------------------------
// exclusive prefix sum of n uint32; in == out allowed. scratch must hold scan_scratch_elems(n) uint32.  If n_dev is not
// null the element count is min(*n_dev, n) read on the device (n is then the capacity the launch is sized for).
size_t scan_scratch_elems(size_t n);
void exclusive_scan_u32(const uint32_t *in, uint32_t *out, size_t n, uint32_t *scratch, hipStream_t s, const uint32_t *n_dev = nullptr);
// stable LSD radix sort of (key,val) pairs on key bits [0,nbits). Result ends in (keys_a, vals_a) if the returned
// value is 0, in (keys_b, vals_b) if 1. scratch must hold sort_scratch_elems(n) uint32.  If n_dev is not null the
// element count is min(*n_dev, n) read on the device (n is then the capacity the launch is sized for).
size_t sort_scratch_elems(size_t n);
int radix_sort_pairs(uint32_t *keys_a, uint32_t *vals_a, uint32_t *keys_b, uint32_t *vals_b, size_t n, int nbits,
                     uint32_t *scratch, hipStream_t s, const uint32_t *n_dev = nullptr);

void launch_clear(const Dims &d, const State &st, hipStream_t s, bool fresh);

}  // namespace sdm
