// moves.hip — object-level prediction of the particle layer (gfx950).
//
// Reference: SemanticDSPMap::subObjectLevelUpdate collection loop + moveParticlesInSetsByTransformations
// (semantic_dsp_map.h:588-699, mc_ring/operations.h:321-362) and ObjectSet::removeObjectByTrackID
// (object_layer.h:414-425).
//
// Owner sets (ObjectParticleHashMap, object_layer.h:20-52) live on the device as a per-slot u16 shadow
// array: slot i belongs to set(T) iff owner[i] == T.  Collecting an object's particles is therefore a
// streaming sweep of that array with a stable multi-bin partition (ballot ranks inside a wave, LDS
// counters across waves, one exclusive scan across blocks) — the result is every moving object's
// particle list in ascending particle index, which is the iteration order the oracle pins for the
// reference's std::unordered_set (DESIGN.md "pinned choices").
#include <hip/hip_ext.h>
#include "sdm_internal.h"
#include "sdm_scratch.h"

#pragma clang fp contract(off)

namespace sdm {

#ifdef SDM_AB_TIMERS
__device__ unsigned long long g_dbg_m[4][4096 * 4];  // [kernel][workgroup][checkpoint], see kernels.hip
void debug_timers_moves(unsigned long long *out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg_m), sizeof(g_dbg_m));
  if (reset) {
    static unsigned long long z[4][4096 * 4];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_m), z, sizeof(z));
  }
}
#define DBGM(k, i, v) do { if (blockIdx.x < 4096 && threadIdx.x == 0) g_dbg_m[k][blockIdx.x * 4 + (i)] = (unsigned long long)(v); } while (0)
#define DBGM_T() wall_clock64()
// every head of k_move_replay (not only thread 0 of a workgroup): maxima / sums over the launch, behind the rows of its 1024 workgroups
#define DBGM_MAX(i, v) atomicMax(&g_dbg_m[1][2048 * 4 + (i)], (unsigned long long)(v))
#define DBGM_ADD(i, v) atomicAdd(&g_dbg_m[1][2048 * 4 + (i)], (unsigned long long)(v))
#else
#define DBGM(k, i, v)
#define DBGM_T() 0ull
#define DBGM_MAX(i, v)
#define DBGM_ADD(i, v)
#endif

namespace {

constexpr int TPB = 256;
constexpr int MV_ITEMS = 16;
constexpr int MV_CHUNK = TPB * MV_ITEMS;  // slots per block
constexpr int MV_WAVES = TPB / 64;
static_assert(MV_CHUNK == (int)OWNER_CHUNK, "one move-sweep block per owner_flag byte");
constexpr uint32_t MV_LIST_CAP = 8192;  // flagged chunks handled per frame (= 33 M slots with an owner nearby)

// rank of the moving object that owns a slot (0xFF if none): the <= 64 moving track ids sit in LDS
__device__ __forceinline__ uint8_t obj_of(uint16_t owner, const uint16_t *tracks, int n_obj) {
  if (owner == OWNER_NONE) return 0xFF;
  uint8_t o = 0xFF;
  for (int k = 0; k < n_obj; ++k)
    if (tracks[k] == owner) o = (uint8_t)k;
  return o;
}

// exclusive prefix of v over the workgroup's TPB threads (thread order); total = the sum.  Two barriers.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *wave_tot, uint32_t &total) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t nb = __shfl_up(inc, off, 64);
    if (lane >= off) inc += nb;
  }
  __syncthreads();  // wave_tot of an earlier call has been read
  if (lane == 63) wave_tot[wid] = inc;
  __syncthreads();
  uint32_t base = 0, t = 0;
#pragma unroll
  for (int w = 0; w < MV_WAVES; ++w) {
    if (w < wid) base += wave_tot[w];
    t += wave_tot[w];
  }
  total = t;
  return base + inc - v;
}

// The member count of the moving objects, ONE launch (round 3: list of flagged chunks, per-chunk counts, device-wide scan -
// three dependent launches on their own stream, 22 us of kernels plus their gaps, and the next frame's k_move_apply
// waited for them 20 us longer than for the main stream).
//   1. Every workgroup finds the flagged chunks itself: the coarse flag level whole (one byte per 64 chunks, 512 bytes
//      for a 256^3 x 8 map), then the 64 flag bytes of every marked group, two block scans - and takes the chunks of
//      rank blockIdx.x, blockIdx.x + gridDim.x, ... ("positions") among them.  Nobody waits for anybody: no list kernel.
//   2. It ranks the chunk's members (ballot ranks per object, as k_move_apply used to) and writes them out as a list
//      in ascending slot order - mv_mem - together with the per-object counts of the chunk.  k_move_apply then has a
//      thread per MEMBER and neither loads owner entries nor ranks.
//   3. No scan: a member's global rank is (members of the objects before its own: mv_tot, added up here with one atomic
//      per chunk and object, every object on a cache line of its own) + (members of its object in the chunks before its
//      own: the workgroup that applies the chunk sums that row of mv_cnt up to its position - a few hundred words) +
//      its rank in the chunk.
// Chunks that also hold older set memberships of moving objects (State::alias) get their primary members listed like any
// other chunk and a flag in mv_nmem; k_move_apply merges the few older memberships into the ranks (its alias path).  A flagged chunk without any owner clears its flag, a marked group without flagged chunk its mark.
constexpr uint32_t MV_COMPLEX = 0xffffffffu;     // (an `ent` of the member count: no member)
constexpr uint32_t MV_COMPLEX_BIT = 0x80000000u;  // in mv_nmem: the chunk also holds older set memberships of moving objects
constexpr uint32_t MV_CLEAR_FLAG = 0x80000000u;  // in mv_list: the chunk holds no owner, k_move_apply clears its flag
constexpr uint32_t MV_GROUP_CAP = 4096;   // marked groups a workgroup lists (x 64 chunks >> MV_LIST_CAP)
constexpr int MV_GRID = (int)FrameBeginLaunch::GRID;
constexpr int MV_PER_WG = (int)(MV_LIST_CAP / MV_GRID);
constexpr uint32_t MV_TOT_STRIDE = 32;    // uint32 per object in mv_tot: one 128-byte line each

__device__ __forceinline__ void move_members_body(const State &st, const MembersArgs &sc, uint32_t n_flags, size_t n_slots, int n_obj,
                                                  const uint16_t *__restrict__ ms_track, uint32_t mv_seq) {
  __shared__ uint16_t tracks[MAX_MOVE_OBJECTS];
  __shared__ uint32_t wave_tot[MV_WAVES];
  __shared__ uint16_t glist[MV_GROUP_CAP];
  __shared__ uint32_t my_chunk[MV_PER_WG];
  __shared__ uint32_t rank_cnt[MV_ITEMS][MV_WAVES][MAX_MOVE_OBJECTS];  // members per (round, wave, object), then running offsets
  __shared__ uint32_t all_cnt[MV_ITEMS][MV_WAVES];                     // members per (round, wave), then running offsets
  __shared__ uint32_t c_obj[MAX_MOVE_OBJECTS];                          // members per object: primary + older memberships
  __shared__ uint32_t any_owner, n_alias_here;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  DBGM(2, 0, DBGM_T());
  if (threadIdx.x < MAX_MOVE_OBJECTS) tracks[threadIdx.x] = (int)threadIdx.x < n_obj ? ms_track[threadIdx.x] : OWNER_NONE;
  // ---- 1a. the coarse level: marked groups, ascending, into glist
  const uint32_t n_groups = (n_flags + OWNER_GROUP - 1) / OWNER_GROUP;
  uint32_t n_marked = 0;
  bool overflow = false;
  for (uint32_t tile = 0; tile < n_groups; tile += TPB * 16) {  // (the level is padded to whole tiles, zero-filled)
    const uint32_t first = tile + threadIdx.x * 16;
    const uint4 q = *reinterpret_cast<const uint4 *>(st.owner_flag2 + first);
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
    uint32_t mask = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if ((w[a] >> (8 * j)) & 0xffu) mask |= 1u << (4 * a + j);
    uint32_t total;
    uint32_t at = n_marked + block_excl_scan((uint32_t)__popc(mask), wave_tot, total);
    while (mask) {
      const int j = __ffs((int)mask) - 1;
      mask &= mask - 1;
      if (at < MV_GROUP_CAP) glist[at] = (uint16_t)(first + (uint32_t)j);
      ++at;
    }
    n_marked += total;
  }
  if (n_marked > MV_GROUP_CAP) {
    n_marked = MV_GROUP_CAP;
    overflow = true;
  }
  __syncthreads();
  // ---- 1b. the flag bytes of the marked groups: this workgroup's positions
  uint32_t n_list = 0;
  for (uint32_t g0 = 0; g0 < n_marked; g0 += TPB) {
    const uint32_t gi = g0 + threadIdx.x;
    unsigned long long m = 0;
    uint32_t grp = 0;
    if (gi < n_marked) {
      grp = glist[gi];
      const uint4 *src = reinterpret_cast<const uint4 *>(st.owner_flag + (size_t)grp * OWNER_GROUP);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const uint4 q = src[a];
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int b4 = 0; b4 < 4; ++b4)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if ((w[b4] >> (8 * j)) & 0xffu) m |= 1ull << (16 * a + 4 * b4 + j);
      }
      if (m == 0ull && blockIdx.x == 0) st.owner_flag2[grp] = 0;  // a mark nobody needs any more
    }
    const uint32_t c = (uint32_t)__popcll(m);
    uint32_t total;
    const uint32_t at = n_list + block_excl_scan(c, wave_tot, total);
    {  // the one position of this workgroup that can fall among the group's (at most 64 < gridDim.x) chunks
      const uint32_t j = at > blockIdx.x ? (at - blockIdx.x + gridDim.x - 1) / gridDim.x : 0u;
      const uint32_t pos = blockIdx.x + j * gridDim.x;
      if (pos < at + c && j < (uint32_t)MV_PER_WG) my_chunk[j] = grp * OWNER_GROUP + (uint32_t)nth_set_bit(m, pos - at);
    }
    n_list += total;
  }
  if (n_list > MV_LIST_CAP) {
    n_list = MV_LIST_CAP;
    overflow = true;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    sc.mv_nlist[0] = n_list;
    sc.cur->move_list_overflow = (overflow || st.alias[0] > st.alias_cap || st.alias[1] != 0u) ? 1u : 0u;  // (alias[1]: the table overflowed in an earlier frame - sticky)
  }
  __syncthreads();
  DBGM(2, 1, DBGM_T());
  // ---- 2. this workgroup's chunks
  const uint32_t na_total = st.alias[0] < st.alias_cap ? st.alias[0] : st.alias_cap;
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  uint32_t *tot = sc.mv_tot + (size_t)(mv_seq & 1u) * MAX_MOVE_OBJECTS * MV_TOT_STRIDE;
  for (int j = 0; j < MV_PER_WG; ++j) {
    const uint32_t pos = blockIdx.x + (uint32_t)j * gridDim.x;
    if (pos >= n_list) break;
    const uint32_t chunk = my_chunk[j];
    const size_t base = (size_t)chunk * MV_CHUNK;
    uint16_t ow[MV_ITEMS];  // the thread's sixteen owner entries, requested together
#pragma unroll
    for (int r = 0; r < MV_ITEMS; ++r) {
      const size_t li = base + (size_t)r * TPB + threadIdx.x;
      ow[r] = li < n_slots ? st.owner[li] : OWNER_NONE;
    }
    __syncthreads();  // the LDS tables of the chunk before have been read
    // (only the columns of this frame's objects are ever read: thread (row t / 4, t % 4) clears its row's columns t % 4, + 4, ...)
    for (int k = (int)(threadIdx.x & 3u); k < n_obj; k += 4) (&rank_cnt[0][0][0])[(threadIdx.x >> 2) * MAX_MOVE_OBJECTS + k] = 0;
    if (threadIdx.x < MV_ITEMS * MV_WAVES) (&all_cnt[0][0])[threadIdx.x] = 0;
    if (threadIdx.x < MAX_MOVE_OBJECTS) c_obj[threadIdx.x] = 0;
    if (threadIdx.x == 0) any_owner = 0, n_alias_here = 0;
    __syncthreads();
    uint32_t ent[MV_ITEMS];  // object << 12 | rank among the object's members in this wave and round << 18; MV_COMPLEX: no member
    uint32_t pre[MV_ITEMS];  // rank among the members of any object in this wave and round
    bool some = false;
#pragma unroll
    for (int r = 0; r < MV_ITEMS; ++r) {
      some = some || ow[r] != OWNER_NONE;
      ent[r] = MV_COMPLEX;
      pre[r] = 0;
      // (a wave's 64 slots are 8 voxels: in nearly every round nobody owns any of them, and a sparse wave pays for every
      // instruction it issues - the seven ballots of a round are 40 of them)
      if (__ballot(ow[r] != OWNER_NONE) == 0ull) continue;
      const uint8_t o = obj_of(ow[r], tracks, n_obj);
      const bool valid = o != 0xFF;
      const uint64_t members = __ballot(valid);
      uint64_t peers = members;
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        const bool bit = (o >> b) & 1u;
        const uint64_t m = __ballot(bit);
        peers &= bit ? m : ~m;
      }
      const uint32_t rank_in_wave = (uint32_t)__popcll(peers & lt_mask);
      if (valid && rank_in_wave == 0) rank_cnt[r][wid][o] = (uint32_t)__popcll(peers);
      if (lane == 0) all_cnt[r][wid] = (uint32_t)__popcll(members);
      ent[r] = valid ? ((uint32_t)o << 12 | rank_in_wave << 18) : MV_COMPLEX;
      pre[r] = (uint32_t)__popcll(members & lt_mask);
    }
    if (some) any_owner = 1;
    if (na_total) {  // older memberships the reference's sets still hold (block-uniform; the table is nearly always empty)
      for (uint32_t k = threadIdx.x; k < na_total; k += TPB) {
        const uint32_t idx = st.alias[2 + 2 * k], trk = st.alias[3 + 2 * k];
        if (trk == OWNER_NONE || idx / MV_CHUNK != chunk) continue;
        any_owner = 1;
        const uint8_t o = obj_of((uint16_t)trk, tracks, n_obj);
        if (o != 0xFF) {
          atomicAdd(&c_obj[o], 1u);
          n_alias_here = 1;
        }
      }
    }
    __syncthreads();
    // Running offsets in (round, wave) order = ascending slot index.  The 64 counts of an object are one per lane of a wave
    // (lane = round * 4 + wave): a wave scan each, the waves share the objects out; "object n_obj" is the count of all
    // members.  (One thread per object walking its 64 counts was 64 dependent LDS round trips: 6 us of the kernel's 17.)
    for (int k = wid; k <= n_obj; k += MV_WAVES) {
      uint32_t *cell = k < n_obj ? &(&rank_cnt[0][0][0])[lane * MAX_MOVE_OBJECTS + k] : &(&all_cnt[0][0])[lane];
      const uint32_t c = *cell;
      uint32_t inc = c;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t nb = __shfl_up(inc, off, 64);
        if (lane >= off) inc += nb;
      }
      *cell = inc - c;
      if (lane == 63) {
        if (k < n_obj) {
          const uint32_t total = inc + c_obj[k];
          sc.mv_cnt[(size_t)k * MV_LIST_CAP + pos] = total;
          if (total) atomicAdd(&tot[k * MV_TOT_STRIDE], total);
        } else {
          sc.mv_nmem[pos] = inc | (n_alias_here ? MV_COMPLEX_BIT : 0u);  // (inc <= MV_CHUNK: the primary members, listed below)
          // A flagged chunk that holds no owner any more loses its flag - in k_move_apply, not here: other workgroups of this
          // launch may still be reading the flags, and they all have to see the same list.
          sc.mv_list[pos] = chunk | (any_owner == 0 ? MV_CLEAR_FLAG : 0u);
        }
      }
    }
    __syncthreads();
    {
      uint32_t *mem = sc.mv_mem + (size_t)pos * MV_CHUNK;
#pragma unroll
      for (int r = 0; r < MV_ITEMS; ++r) {
        if (ent[r] == MV_COMPLEX) continue;
        const uint32_t o = (ent[r] >> 12) & 63u;
        const uint32_t rank = (ent[r] >> 18) + rank_cnt[r][wid][o];
        mem[all_cnt[r][wid] + pre[r]] = ((uint32_t)r * TPB + threadIdx.x) | o << 12 | rank << 18;
      }
    }
    DBGM(2, 2, DBGM_T());
  }
  DBGM(2, 3, DBGM_T());
}
__global__ __launch_bounds__(TPB) void k_move_members(State st, MembersArgs ma, const FrameArgs *__restrict__ fa) {
  const int n_obj = fa->n_obj;
  if (n_obj <= 0) return;  // no object moves in this frame
  move_members_body(st, ma, ma.n_flags, ma.n_slots, n_obj, fa->ms.track, fa->mv_seq);
}
// The same as the first kernel of a chain of its own (sharded maps: the member count starts behind the PREVIOUS frame's
// births on its own stream, and the all-gather of the counts rides that stream): the frame block arrives by value - every
// workgroup reads what it needs from the kernel arguments - and is stored for the chain's other kernel.
__global__ __launch_bounds__(TPB) void k_move_members_v(State st, MembersArgs ma, const FrameArgs src, FrameArgs *__restrict__ dst) {
  if (blockIdx.x == 0) {
    const uint32_t *s4 = reinterpret_cast<const uint32_t *>(&src);
    uint32_t *d4 = reinterpret_cast<uint32_t *>(dst);
    for (uint32_t i = threadIdx.x; i < sizeof(FrameArgs) / 4; i += blockDim.x) d4[i] = s4[i];
  }
  if (src.n_obj <= 0) return;
  move_members_body(st, ma, ma.n_flags, ma.n_slots, src.n_obj, src.ms.track, src.mv_seq);
}

// First kernel of a frame on the main stream: the frame's scalars (pose, ring state, stamp updates, object motions,
// removals, input pointers) arrive by value and are stored where the frame's other kernels read them (FrameArgs,
// sdm_scratch.h) - and, inside a graph, in the side chains' block too; the per-frame counters and the per-pixel bin
// counts are zeroed, the recycled slabs get their stamps and their voxels' results (mark_slab_voxel_dirty).
// with_members: the SAME launch is the member count of the frame's moving objects (move_members_body reads the object
// list from the kernel arguments).  A whole map needs nothing from anybody between the previous frame's sweep and this
// frame's k_move_apply but this one launch; round 3 ran the count as a chain of three launches on a stream of its own
// and k_move_apply started 25-32 us after the sweep had ended (two event hand-overs, the chain's gaps).
__global__ __launch_bounds__(TPB) void k_frame_begin(Counters *cnt, uint32_t *__restrict__ bin_count, uint32_t n_bins,
                                                      State st, const FrameArgs src, FrameArgs *__restrict__ dst_main,
                                                      FrameArgs *__restrict__ dst_side, Dims d, uint32_t slab_max, MembersArgs ma,
                                                      int with_members) {
  const StampUpdates &su = src.su;
  DBGM(3, 0, DBGM_T());
  if (blockIdx.x == gridDim.x - 1) {
    const uint32_t *s4 = reinterpret_cast<const uint32_t *>(&src);
    uint32_t *m4 = reinterpret_cast<uint32_t *>(dst_main), *e4 = reinterpret_cast<uint32_t *>(dst_side);
    for (uint32_t k = threadIdx.x; k < sizeof(FrameArgs) / 4; k += blockDim.x) {
      m4[k] = s4[k];
      if (e4) e4[k] = s4[k];
    }
  }
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  for (uint32_t t = i; t < slab_max * (uint32_t)su.n; t += gridDim.x * blockDim.x) mark_slab_voxel_dirty(d, st, su, slab_max, t);
  if (i < offsetof(Counters, flood_complex) / 4) reinterpret_cast<uint32_t *>(cnt)[i] = 0;  // the flood flags belong to the frustum chain
  if (i < (uint32_t)su.n) {  // this frame's recycled slabs (no host-to-device copy of the stamp arrays)
    const uint32_t e = su.entry[i], axis = e >> 12, idx = e & 0xfffu;
    uint32_t *arr = axis == 0 ? st.stamps_x : (axis == 1 ? st.stamps_y : st.stamps_z);
    arr[idx] = su.value;
  }
  for (; i < n_bins; i += gridDim.x * blockDim.x) bin_count[i] = 0;
  DBGM(3, 1, DBGM_T());
  if (with_members && src.n_obj > 0) move_members_body(st, ma, ma.n_flags, ma.n_slots, src.n_obj, src.ms.track, src.mv_seq);
}

// ---- global ranks across Z-slab shards ------------------------------------------------------------------
// The noise-table cursor and the re-insertion order of the reference run over (object order, ascending particle
// index).  With the map split into Z slabs (ring-z = high index bits) that order is: object, then shard, then local
// index.  Every shard publishes its per-object member counts (HALO_OBJ ints), the counts of all shards are gathered,
// and the global rank of the j-th local member of object k on shard r is
//     e = sum_{k'<k} sum_r' C[r'][k']  +  sum_{r'<r} C[r'][k]  +  j.
// Moved copies are stored at index e and replayed per target voxel in ascending e: the reference's order, without
// knowing where a copy came from.  A copy whose target voxel belongs to another slab is exported as a 36-byte record
// into the segment of the export buffer that is addressed to that slab's shard (one segment per shard: 16-byte header
// with the record count, then halo_cap records); the segments travel as an all-to-all, so a shard receives what was
// meant for it and nothing else.
struct HaloRecord {
  float x, y, z;
  uint32_t forget_bits;
  float w;
  uint32_t voxel;
  uint32_t e;
  uint32_t ts_track;            // ts | track << 16
  uint32_t owner_label_status;  // owner | label << 16 | status << 24
};
static_assert(sizeof(HaloRecord) == HALO_RECORD_BYTES, "halo record layout");

// reset_export: the first batch of a frame's object list (a later batch appends to the frame's export segments)
__global__ void k_move_local_counts(int32_t *counts_local, Scratch sc, const FrameArgs *__restrict__ fa, int reset_export) {
  const int k = threadIdx.x;
  const int n_obj = fa->n_obj;  // (runs with the member count, on its stream)
  // this frame's export counters, one per destination shard
  const uint32_t world = sc.halo_world;
  if (sc.halo_send && reset_export)
    for (uint32_t dst = (uint32_t)k; dst < world; dst += blockDim.x)
      *reinterpret_cast<uint32_t *>(sc.halo_send + (size_t)dst * halo_segment_bytes(sc.halo_cap)) = 0;
  if (k >= HALO_OBJ) return;
  const uint32_t *tot = sc.mv_tot + (size_t)(fa->mv_seq & 1u) * MAX_MOVE_OBJECTS * MV_TOT_STRIDE;
  counts_local[k] = k < n_obj ? (int32_t)tot[k * MV_TOT_STRIDE] : 0;
}

constexpr uint32_t MV_NIL = 0xffffffffu;

// A moved copy of global rank e joins its target voxel's list (Scratch::mv_row): word 0 of the voxel's row hands out
// places; the first MV_DIRECT arrivals write their rank into the row - the replay fetches all of them with one 64-byte
// load - the later ones are chained through the row's last word and mv_next (push-front; the replay restores rank order
// either way).  The copy that arrived FIRST replays the list, and its entry says so (MV_FIRST, set here, behind the
// copy's own store): the replay needs no list of touched voxels, and the thread of any other copy leaves after one load.
// (Rounds 4-6 chained every arrival and made every copy's thread look the voxel up: a list of seventeen copies was
// seventeen dependent loads, and half of the replay's threads fetched a record, an owner row, a forget row and a list head
// for a list that was not theirs - the kernel is bound by the scattered lines it touches, tools/probes/timers_moves.py.)
__device__ __forceinline__ void move_link(const Dims &d, const Scratch &sc, uint32_t v, uint32_t e, uint32_t forget_bits) {
  const uint32_t lv = v - d.v_begin;
  uint32_t *row = sc.mv_row + (size_t)lv * MV_ROW;
  const uint32_t k = atomicAdd(&row[0], 1u) + 1u;  // (idle: all ones)
  if (k < (uint32_t)MV_DIRECT) {
    row[1 + k] = e;
    if (k == 0u) sc.mv_copy[e].forget_bits = forget_bits | MV_FIRST;
  } else {
    const uint32_t prev = atomicExch(&row[MV_ROW - 1], e);  // (MV_NIL when idle: the replay leaves it that way)
    sc.mv_next[e] = prev;
  }
}

// one member of moving object `obj`, global rank e, local slot li
// alias: the membership is an older one kept in State::alias (the slot's owner entry belongs to somebody else and
// stays); copy_invalid: an object earlier in this frame's order already moved this very slot, the reference then
// copies a particle whose status it has just set to INVALID (operations.h:339-349 run object by object).
// (in two halves, so that a thread with several members can request all their loads before the first store goes out:
// one member is a chain of three dependent round trips, and stores between two members keep the compiler from overlapping
// them)
struct MoveLoaded {
  float4 p;
  float n[3];
  float w;
  uint16_t ts, track;
  uint8_t label, status;
};
// (in two steps: the particle's own fields need its slot only, the noise draw its global rank)
__device__ __forceinline__ void move_load_particle(const Dims &d, const State &st, size_t li, bool copy_invalid, MoveLoaded &m) {
  m.p = st.pos4[li];
  m.p.w = __uint_as_float((uint32_t)st.forget[li]);  // (the forget count travels in the copy's fourth word, as it did when it lived there)
  const SlotRef sr = slot_ref_li(st, d.p_n, li);
  m.w = sr.w();
  m.ts = sr.ts();
  m.track = sr.track();
  m.label = sr.label();
  m.status = copy_invalid ? (uint8_t)ST_INVALID : sr.status();
}
__device__ __forceinline__ void move_load_noise(const Filter &flt, const State &st, long long cursor, uint32_t e, MoveLoaded &m) {
  const long long draw = cursor + 3ll * e;
  m.n[0] = st.noise[(draw + 1) % flt.noise_n];
  m.n[1] = st.noise[(draw + 2) % flt.noise_n];
  m.n[2] = st.noise[(draw + 3) % flt.noise_n];
}
__device__ __forceinline__ MoveLoaded move_load(const Dims &d, const Filter &flt, const State &st, long long cursor, uint32_t e,
                                                size_t li, bool copy_invalid) {
  MoveLoaded m;
  move_load_particle(d, st, li, copy_invalid, m);
  move_load_noise(flt, st, cursor, e, m);
  return m;
}
__device__ __forceinline__ void move_store(const Dims &d, const Frame &f, const MoveSet &ms, const State &st, const Scratch &sc,
                                           const MoveLoaded &m, int obj, uint32_t e, size_t li, bool alias) {
  const float4 p = m.p;
  const float *T = ms.T[obj];
  float nx = row4(T + 0, p.x, p.y, p.z);
  float ny = row4(T + 4, p.x, p.y, p.z);
  float nz = row4(T + 8, p.x, p.y, p.z);
  nx = nx + m.n[0];
  ny = ny + m.n[1];
  nz = nz + m.n[2];
  const float pw = m.w;
  const uint16_t pts = m.ts, ptrack = m.track;
  const uint8_t plabel = m.label, pstatus = m.status;
  const uint16_t powner = ms.track[obj];
  slot_ref_li(st, d.p_n, li).set_status(ST_INVALID);  // deleteParticleByIndex
  st.vflag[li >> d.p_n] = VF_DIRTY;
  mark_tile(st, li >> d.p_n, f.epoch);
  if (!alias) st.owner[li] = OWNER_NONE;  // the object's set is replaced by the re-inserted indices (semantic_dsp_map.h:697-699)
  uint32_t rx, ry, rz;
  uint32_t v = global_pos_to_voxel(d, f, nx, ny, nz, rx, ry, rz);
  if (v == INVALID_INDEX) return;  // left the map: dropped (operations.h:799-802)
  if (rz >= d.rz_begin && rz < d.rz_begin + d.rz_count) {
    if (e >= sc.cap_move) return;
    MoveCopy c;
    c.x = nx;
    c.y = ny;
    c.z = nz;
    c.forget_bits = __float_as_uint(p.w) & 0xffu;
    c.w = pw;
    c.ts = pts;
    c.track = ptrack;
    c.owner = powner;
    c.label = plabel;
    c.status = pstatus;
    c.voxel = v - d.v_begin;
    sc.mv_copy[e] = c;
    move_link(d, sc, v, e, c.forget_bits);
  } else if (sc.halo_send) {  // crosses into another slab: export to the shard that owns it
    unsigned char *seg = sc.halo_send + (size_t)(rz / d.rz_count) * halo_segment_bytes(sc.halo_cap);
    uint32_t k = atomicAdd(reinterpret_cast<uint32_t *>(seg), 1u);
    if (k < sc.halo_cap) {
      HaloRecord r;
      r.x = nx;
      r.y = ny;
      r.z = nz;
      r.forget_bits = __float_as_uint(p.w);
      r.w = pw;
      r.voxel = v;
      r.e = e;
      r.ts_track = (uint32_t)pts | ((uint32_t)ptrack << 16);
      r.owner_label_status = (uint32_t)powner | ((uint32_t)plabel << 16) | ((uint32_t)pstatus << 24);
      reinterpret_cast<HaloRecord *>(seg + HALO_HEADER_BYTES)[k] = r;
    } else {
      sc.cnt->overflow = 1;
      atomicAdd(&sc.cnt->n_halo_dropped, 1u);
    }
  }
}
__device__ __forceinline__ void move_one(const Dims &d, const Frame &f, const Filter &flt, const MoveSet &ms, const State &st,
                                         const Scratch &sc, int obj, uint32_t e, size_t li, bool alias = false,
                                         bool copy_invalid = false) {
  const MoveLoaded m = move_load(d, flt, st, (long long)sc.cur->move_cursor, e, li, copy_invalid);
  move_store(d, f, ms, st, sc, m, obj, e, li, alias);
}

// k_move_apply in a chunk that also holds older set memberships of moving objects (State::alias: a slot can then belong to
// several of them).  The member count has listed and ranked the chunk's PRIMARY members (owner[]) like any other chunk's;
// the older memberships - a handful per chunk - are merged in here:
//   rank of a membership (slot s, object o) among o's members in the chunk
//       = primary members of o below s  +  older memberships of o below s.
// A slot that several moving objects hold is moved by each of them, in object order, and all but the first copy a particle
// that has just been invalidated (operations.h:339-349 runs object by object): ONE thread handles all memberships of such a
// slot - the thread of its primary member if it has one, else the thread of its first table entry - so that nobody reads a
// status byte somebody else is writing.  (Rounds 3-5 ranked such a chunk round by round, 256 slots at a time with three
// barriers and a ballot per object and round, its moves a chain of sixteen dependent trips: 30-40 us for ONE chunk, and
// k_move_apply - which ends with its slowest workgroup - took 42 us on the `driven` workload against 14 us on maps without
// such memberships.)
constexpr uint32_t CA_CAP = 512;
struct ChunkAliases {
  const uint32_t *idx;   // slot index (shard-local) of entry c
  const uint32_t *ent;   // its position in State::alias
  const uint8_t *obj;    // rank of its track among the frame's moving objects
  uint32_t n;
};
// older memberships of object o in this chunk below slot index li
__device__ __forceinline__ uint32_t aliases_below(const ChunkAliases &ca, uint32_t o, uint32_t li) {
  uint32_t c = 0;
  for (uint32_t k = 0; k < ca.n; ++k) c += (ca.obj[k] == o && ca.idx[k] < li) ? 1u : 0u;
  return c;
}
// primary members of object o in this chunk below slot-in-chunk s (the member list is in ascending slot order)
__device__ __forceinline__ uint32_t primaries_below(const uint32_t *mem, uint32_t nm, uint32_t o, uint32_t s) {
  uint32_t c = 0;
  for (uint32_t i = 0; i < nm; ++i) {
    const uint32_t en = mem[i];
    if ((en & 4095u) >= s) break;
    c += ((en >> 12) & 63u) == o ? 1u : 0u;
  }
  return c;
}
// all memberships of one slot, in object order.  prim_o: the object of its primary membership (0xFF: none that moves),
// prim_rank: that membership's rank among the primary members of its object in the chunk.
__device__ __forceinline__ void move_slot_memberships(const Dims &d, const Frame &f, const Filter &flt, const MoveSet &ms, const State &st,
                                                      const Scratch &sc, const ChunkAliases &ca, const uint32_t *mem, uint32_t nm,
                                                      const uint32_t *obj_base, size_t li, uint32_t s, uint32_t prim_o, uint32_t prim_rank) {
  int last = -1;  // object of the membership handled last
  bool first = true;
  for (;;) {
    // the membership with the smallest object above `last`
    uint32_t ob = 0xFFu, entry = 0;
    bool is_alias = false;
    if (prim_o != 0xFFu && (int)prim_o > last) ob = prim_o;
    for (uint32_t k = 0; k < ca.n; ++k)
      if (ca.idx[k] == (uint32_t)li && (int)ca.obj[k] > last && ca.obj[k] < ob) {
        ob = ca.obj[k];
        entry = ca.ent[k];
        is_alias = true;
      }
    if (ob == 0xFFu) break;
    const uint32_t below = (!is_alias ? prim_rank : primaries_below(mem, nm, ob, s)) + aliases_below(ca, ob, (uint32_t)li);
    move_one(d, f, flt, ms, st, sc, (int)ob, obj_base[ob] + below, li, is_alias, !first);
    if (is_alias) st.alias[3 + 2 * entry] = OWNER_NONE;  // the object's set is rebuilt from its re-inserted copies
    last = (int)ob;
    first = false;
  }
}

// phase 1 of moveParticlesInSetsByTransformations (operations.h:331-349): copy, transform + table noise, delete the
// original.  One workgroup per listed chunk, one thread per member (k_move_members listed and ranked them).  A member's
// global rank e = rank among all members of all moving objects in (object, shard, index) order:
//   members of the objects before its own (mv_tot, or the gathered counts of all shards)
// + members of its own object on the shards before this one
// + members of its object in the listed chunks before this one (the workgroup adds up that row of mv_cnt)
// + its rank in the chunk.
// The noise cursor advances by three per particle in that order; the copy joins its target voxel's list or is exported.
// Chunks with older set memberships (MV_COMPLEX) are ranked here, round by round, on the alias path.
__global__ __launch_bounds__(TPB) void k_move_apply(Dims d, Filter flt, State st, Scratch sc, const int32_t *__restrict__ counts_all,
                                                    int world, int rank) {
  // Everything whose address depends on nothing but the workgroup's first position goes out before the first value is
  // looked at (a dependent fetch costs 1-2 us in these short kernels, whatever its size): the chunk's member count, its
  // first members, its per-object counts, both parities of the per-object totals.  Beyond the list they are garbage
  // nobody looks at (the arrays hold MV_LIST_CAP positions, the grid is smaller).
  uint32_t pos = blockIdx.x;
  uint32_t nm = sc.mv_nmem[pos], listed = sc.mv_list[pos];
  uint32_t first_entry = sc.mv_mem[(size_t)pos * MV_CHUNK + threadIdx.x];
  uint32_t my_c = threadIdx.x < MAX_MOVE_OBJECTS ? sc.mv_cnt[(size_t)threadIdx.x * MV_LIST_CAP + pos] : 0u;
  uint32_t tot2[2] = {0u, 0u};
  if (threadIdx.x < MAX_MOVE_OBJECTS) {
    tot2[0] = sc.mv_tot[(size_t)threadIdx.x * MV_TOT_STRIDE];
    tot2[1] = sc.mv_tot[((size_t)MAX_MOVE_OBJECTS + threadIdx.x) * MV_TOT_STRIDE];
  }
  const long long cursor = (long long)sc.cur->move_cursor;  // (advanced by k_move_replay, after this kernel)
  const uint32_t n = *sc.mv_nlist;
  const uint32_t na_raw = st.alias[0];
  const int n_obj = sc.fa->n_obj;
  if (n_obj <= 0) return;
  DBGM(0, 0, DBGM_T());
  const Frame f = sc.fa->f;  // a copy (uniform registers): stores of the kernel cannot alias it
  const MoveSet &ms = sc.fa->ms;
  const uint32_t par = sc.fa->mv_seq & 1u;
  // ranks continue where the batch of objects before this one ended (nobody writes that word in this launch: block 0
  // writes the OTHER parity's)
  const uint32_t batch = sc.fa->mv_batch;
  const uint32_t e_base = batch ? sc.cnt->n_moved_b[(batch - 1u) & 1u] : 0u;
  __shared__ uint32_t obj_base[MAX_MOVE_OBJECTS];  // global rank of the object's next member in this chunk
  __shared__ uint32_t obj_first[MAX_MOVE_OBJECTS]; // global rank of the object's first member on this shard
  __shared__ uint32_t c_here[MAX_MOVE_OBJECTS];    // the object's members in this chunk
  __shared__ uint32_t tot_l[MAX_MOVE_OBJECTS];     // the object's members on this shard
  __shared__ uint16_t tracks[MAX_MOVE_OBJECTS];
  __shared__ uint32_t ca_idx[CA_CAP], ca_ent[CA_CAP], ca_n;
  __shared__ uint8_t ca_obj[CA_CAP];
  __shared__ uint32_t cm_mem[MV_CHUNK];  // the member list of a chunk on the alias path
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const size_t n_slots = (size_t)d.v_count * d.S;
  if (threadIdx.x < MAX_MOVE_OBJECTS) {
    tracks[threadIdx.x] = (int)threadIdx.x < n_obj ? ms.track[threadIdx.x] : OWNER_NONE;
    tot_l[threadIdx.x] = (int)threadIdx.x < n_obj ? tot2[par] : 0u;
  }
  __syncthreads();
  if ((int)threadIdx.x < n_obj) {
    uint32_t run = 0;
    if (counts_all) {
      for (int k = 0; k < (int)threadIdx.x; ++k)
        for (int r = 0; r < world; ++r) run += (uint32_t)counts_all[r * HALO_OBJ + k];
      for (int r = 0; r < rank; ++r) run += (uint32_t)counts_all[r * HALO_OBJ + threadIdx.x];
    } else {
      for (int k = 0; k < (int)threadIdx.x; ++k) run += tot_l[k];
    }
    obj_first[threadIdx.x] = e_base + run;
  }
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) {
      uint32_t total = 0;
      if (counts_all) {
        for (int k = 0; k < n_obj; ++k)
          for (int r = 0; r < world; ++r) total += (uint32_t)counts_all[r * HALO_OBJ + k];
      } else {
        for (int k = 0; k < n_obj; ++k) total += tot_l[k];
      }
      sc.cnt->n_moved = e_base + total;
      sc.cnt->n_moved_b[batch & 1u] = e_base + total;
      if (e_base + total > sc.cap_move || sc.cur->move_list_overflow) sc.cnt->overflow = 1;
    }
    // the totals of the next frame with moving objects start at zero (its member count runs after this kernel)
    if (threadIdx.x < MAX_MOVE_OBJECTS) sc.mv_tot[((size_t)(par ^ 1u) * MAX_MOVE_OBJECTS + threadIdx.x) * MV_TOT_STRIDE] = 0;
  }
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const uint32_t na_total = na_raw < st.alias_cap ? na_raw : st.alias_cap;
  while (pos < n) {
    const uint32_t chunk = listed & ~MV_CLEAR_FLAG;
    const size_t base = (size_t)chunk * MV_CHUNK;
    __syncthreads();  // the LDS tables of the chunk before have been read (and obj_first is there)
    if ((int)threadIdx.x < n_obj) c_here[threadIdx.x] = my_c;
    if (threadIdx.x == 0 && (listed & MV_CLEAR_FLAG)) st.owner_flag[chunk] = 0;  // no owner in the chunk any more
    // the thread's member of the chunk (nearly always the only one: a chunk holds a dozen members on average): its own
    // fields are requested before the ranks are known
    MoveLoaded ml;
    const bool simple = (nm & MV_COMPLEX_BIT) == 0u;
    const uint32_t nm_raw = nm;
    nm &= ~MV_COMPLEX_BIT;  // the chunk's primary members
    const bool mine = threadIdx.x < nm;
    if (mine) move_load_particle(d, st, base + (first_entry & 4095u), false, ml);
    __syncthreads();
    if (nm_raw != 0) {  // (workgroup-uniform)
      // the object's members in the chunks before this one: every wave takes the objects of its number
      for (int k = wid; k < n_obj; k += MV_WAVES) {
        if (c_here[k] == 0) continue;  // (wave-uniform)
        uint32_t sum = 0;
        const uint32_t *row = sc.mv_cnt + (size_t)k * MV_LIST_CAP;
        for (uint32_t p = (uint32_t)lane; p < pos; p += 64) sum += row[p];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
        if (lane == 0) obj_base[k] = obj_first[k] + sum;
      }
      __syncthreads();
      DBGM(0, 1, DBGM_T());
      if (simple) {
        if (mine) {
          const uint32_t o = (first_entry >> 12) & 63u;
          const uint32_t e = obj_base[o] + (first_entry >> 18);
          move_load_noise(flt, st, cursor, e, ml);
          move_store(d, f, ms, st, sc, ml, (int)o, e, base + (first_entry & 4095u), false);
        }
        for (uint32_t i = TPB + threadIdx.x; i < nm; i += TPB) {  // (a chunk with more than 256 members)
          const uint32_t en = sc.mv_mem[(size_t)pos * MV_CHUNK + i];
          const uint32_t o = (en >> 12) & 63u;
          move_one(d, f, flt, ms, st, sc, (int)o, obj_base[o] + (en >> 18), base + (en & 4095u));
        }
        DBGM(0, 2, DBGM_T());
      } else {
        // older memberships of moving objects inside this chunk (State::alias): slot, object rank, entry
        if (threadIdx.x == 0) ca_n = 0;
        for (uint32_t i = threadIdx.x; i < nm; i += TPB) cm_mem[i] = i == threadIdx.x ? first_entry : sc.mv_mem[(size_t)pos * MV_CHUNK + i];
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < na_total; k += blockDim.x) {
          const uint32_t idx = st.alias[2 + 2 * k], trk = st.alias[3 + 2 * k];
          if (trk == OWNER_NONE || idx / MV_CHUNK != chunk) continue;
          const uint8_t o = obj_of((uint16_t)trk, tracks, n_obj);
          if (o == 0xFF) continue;
          const uint32_t c = atomicAdd(&ca_n, 1u);
          if (c < CA_CAP) {
            ca_idx[c] = idx;
            ca_obj[c] = o;
            ca_ent[c] = k;
          }
        }
        __syncthreads();
        ChunkAliases ca;
        ca.idx = ca_idx;
        ca.ent = ca_ent;
        ca.obj = ca_obj;
        ca.n = ca_n < CA_CAP ? ca_n : CA_CAP;
        if (threadIdx.x == 0 && ca_n > CA_CAP) sc.cnt->overflow = 1;
        // the primary members: one thread each.  A slot nobody else holds is moved like any other chunk's (the thread's
        // first member: with the particle it requested above), its rank moved up by its object's older memberships below it.
        for (uint32_t i = threadIdx.x; i < nm; i += TPB) {
          const uint32_t en = cm_mem[i], sl = en & 4095u, o = (en >> 12) & 63u;
          const size_t li = base + sl;
          bool shared_slot = false;
          for (uint32_t k = 0; k < ca.n; ++k) shared_slot = shared_slot || ca.idx[k] == (uint32_t)li;
          if (!shared_slot) {
            const uint32_t e = obj_base[o] + (en >> 18) + aliases_below(ca, o, (uint32_t)li);
            if (i == threadIdx.x) {
              move_load_noise(flt, st, cursor, e, ml);
              move_store(d, f, ms, st, sc, ml, (int)o, e, li, false);
            } else {
              move_one(d, f, flt, ms, st, sc, (int)o, e, li);
            }
          } else {
            move_slot_memberships(d, f, flt, ms, st, sc, ca, cm_mem, nm, obj_base, li, sl, o, en >> 18);
          }
        }
        // older memberships of slots whose primary owner is not moving: the thread of the slot's first table entry
        for (uint32_t c = threadIdx.x; c < ca.n; c += TPB) {
          const uint32_t li = ca.idx[c], sl = li - (uint32_t)base;
          bool other = false;
          for (uint32_t k = 0; k < c; ++k) other = other || ca.idx[k] == li;
          for (uint32_t i = 0; i < nm && !other; ++i) {
            const uint32_t s2 = cm_mem[i] & 4095u;
            if (s2 >= sl) {
              other = s2 == sl;
              break;
            }
          }
          if (!other) move_slot_memberships(d, f, flt, ms, st, sc, ca, cm_mem, nm, obj_base, li, sl, 0xFFu, 0u);
        }
      }
    }
    pos += gridDim.x;
    if (pos < n) {
      nm = sc.mv_nmem[pos];
      listed = sc.mv_list[pos];
      first_entry = sc.mv_mem[(size_t)pos * MV_CHUNK + threadIdx.x];
      my_c = threadIdx.x < MAX_MOVE_OBJECTS ? sc.mv_cnt[(size_t)threadIdx.x * MV_LIST_CAP + pos] : 0u;
    }
  }
  DBGM(0, 3, DBGM_T());
}

// import the records other shards exported into this slab (blockIdx.y = source shard; segment s of the receive buffer is
// what shard s addressed to this one)
__global__ __launch_bounds__(TPB) void k_move_import(Dims d, Scratch sc, int world, int rank) {
  if (sc.fa->n_obj <= 0) return;
  const int src = blockIdx.y;
  if (src == rank || src >= world) return;
  const unsigned char *buf = sc.halo_recv + (size_t)src * halo_segment_bytes(sc.halo_cap);
  uint32_t n = *reinterpret_cast<const uint32_t *>(buf);
  if (n > sc.halo_cap) n = sc.halo_cap;
  const HaloRecord *rec = reinterpret_cast<const HaloRecord *>(buf + HALO_HEADER_BYTES);
  uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
    const HaloRecord r = rec[k];
    uint32_t rz = (r.voxel >> (d.x_n + d.y_n)) & (d.NZ - 1);
    if (rz < d.rz_begin || rz >= d.rz_begin + d.rz_count) continue;
    const uint32_t e = r.e;
    if (e >= sc.cap_move) continue;
    MoveCopy c;
    c.x = r.x;
    c.y = r.y;
    c.z = r.z;
    c.forget_bits = r.forget_bits & 0xffu;
    c.w = r.w;
    c.ts = (uint16_t)(r.ts_track & 0xffffu);
    c.track = (uint16_t)(r.ts_track >> 16);
    c.owner = (uint16_t)(r.owner_label_status & 0xffffu);
    c.label = (uint8_t)((r.owner_label_status >> 16) & 0xffu);
    c.status = (uint8_t)(r.owner_label_status >> 24);
    c.voxel = r.voxel - d.v_begin;
    sc.mv_copy[e] = c;
    move_link(d, sc, r.voxel, e, c.forget_bits);
  }
}

// phase 2 (operations.h:351-361): re-insert the copies, first vacant slot, in (object, index) order = ascending global
// rank.  One thread per COPY: the copy that arrived first at its voxel (MV_FIRST; the others leave at once) fetches the
// voxel's row, sorts the ranks, fetches the copies - all of them in one round - and replays them in rank order while the voxel has a vacant slot
// (a copy whose own stamp is older than the slab's leaves its slot vacant, so a list can be longer than the voxel; once
// the voxel is full every later copy is dropped, operations.h:357).  Counter and chain head are left idle again.  (Copies
// that left the map, went to another shard or belong to other shards' members were not written this frame: whatever their
// entry holds, word 0 of no row of this frame names it.)
// A list of more than MV_DIRECT copies (a voxel on the surface of an object that has been tracked for a hundred frames
// receives dozens of copies of which many are vacant themselves - dead members of the set are copied too,
// operations.h:334-349) takes the path of rounds 4-6 for what the row does not hold: the chain is walked once, the ranks
// kept in LDS, and batches of S-1 ranks are selected from there.
constexpr int RP_TPB = 64;
constexpr int RP_GRID = 1024;
constexpr int RP_KEEP = 96;  // ranks of a voxel's list kept in LDS (24 KB per workgroup)
// Batcher's odd-even merge sort of 16 keys: 63 compare-exchanges, no branch (checked with the 0-1 principle when it was generated)
constexpr uint8_t RP_SORT16[63][2] = {
    {0, 1}, {2, 3}, {0, 2}, {1, 3}, {1, 2}, {4, 5}, {6, 7}, {4, 6}, {5, 7}, {5, 6}, {0, 4}, {2, 6}, {2, 4}, {1, 5}, {3, 7}, {3, 5},
    {1, 2}, {3, 4}, {5, 6}, {8, 9}, {10, 11}, {8, 10}, {9, 11}, {9, 10}, {12, 13}, {14, 15}, {12, 14}, {13, 15}, {13, 14}, {8, 12},
    {10, 14}, {10, 12}, {9, 13}, {11, 15}, {11, 13}, {9, 10}, {11, 12}, {13, 14}, {0, 8}, {4, 12}, {4, 8}, {2, 10}, {6, 14}, {6, 10},
    {2, 4}, {6, 8}, {10, 12}, {1, 9}, {5, 13}, {5, 9}, {3, 11}, {7, 15}, {7, 11}, {3, 5}, {7, 9}, {11, 13}, {1, 2}, {3, 4}, {5, 6},
    {7, 8}, {9, 10}, {11, 12}, {13, 14}};
static_assert(MV_ROW == 16, "the row's ranks are sorted by a 16-key network");
template <int S>
__global__ __launch_bounds__(RP_TPB) void k_move_replay(Dims d, Filter flt, State st, Scratch sc) {
  // (the thread's copy: its address depends on the thread's number only, so it is requested before anything else is
  // looked at - stale or garbage beyond this frame's copies, where nobody looks)
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  MoveCopy c0;
  c0.voxel = MV_NIL;
  if (t < sc.cap_move) c0 = sc.mv_copy[t];
  const uint32_t n_alias = st.alias[0];
  if (sc.fa->n_obj <= 0) return;
  const uint32_t n_moved = sc.cnt->n_moved;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // the table cursor of RingBufferOperations::gaussian_random_calculator_ advanced by three per moved particle
    long long c = (long long)sc.cur->move_cursor + 3ll * (long long)n_moved;
    sc.cur->move_cursor = (int32_t)(c % flt.noise_n);
  }
  const uint32_t n_copies = n_moved < sc.cap_move ? n_moved : sc.cap_move;
  const uint32_t epoch = sc.fa->f.epoch;
  const uint32_t stride = gridDim.x * blockDim.x;
  const bool overflow = sc.cnt->overflow != 0;
  bool first_round = true;
  __shared__ uint32_t n_ok_block;  // (statistic: one global atomic per workgroup, not per voxel)
  __shared__ uint32_t kept[RP_KEEP][RP_TPB];  // long lists: the ranks this thread replays (column = thread: conflict-free)
  if (threadIdx.x == 0) n_ok_block = 0;
  __syncthreads();
  // (no `cond ? object : object` on MoveCopy anywhere here: the conditional operator on two lvalues selects an ADDRESS - one
  // of them in global memory, the other the thread's own copy - so the compiler kept the thread's copy in scratch memory,
  // fetched through a FLAT load, and re-read it from scratch for every insertion)
  for (; t < n_copies; t += stride) {
    if (first_round) {
      first_round = false;
    } else {
      c0 = sc.mv_copy[t];
    }
    DBGM(1, 0, DBGM_T());
    [[maybe_unused]] const unsigned long long dbg_t0 = DBGM_T();
    const uint32_t lv = c0.voxel;
    if (lv >= d.v_count || !(c0.forget_bits & MV_FIRST)) continue;  // not the first arrival at its voxel: nothing to do
    // the voxel's row, and - requested with it - the voxel's rows of the map
    uint32_t *const row = sc.mv_row + (size_t)lv * MV_ROW;
    uint32_t rk[MV_ROW];
    {
      const uint4 *rp = reinterpret_cast<const uint4 *>(row);
#pragma unroll
      for (int q = 0; q < MV_ROW / 4; ++q) {
        const uint4 x = rp[q];
        rk[4 * q] = x.x;
        rk[4 * q + 1] = x.y;
        rk[4 * q + 2] = x.z;
        rk[4 * q + 3] = x.w;
      }
    }
    const uint32_t v = d.v_begin + lv;
    uint32_t rx, ry, rz;
    voxel_to_ring(d, v, rx, ry, rz);
    uint32_t a = st.stamps_x[rx], b = st.stamps_y[ry], c = st.stamps_z[rz];
    const size_t base = (size_t)lv * S;
    uint8_t stv[S];
    uint16_t tsv[S];
    unsigned char *const rec = rec_ptr(st, S, lv);
    // The voxel's whole record and its forget counts: the replay works on register copies of the voxel's rows and stores
    // each row ONCE when the list is through.  (Until round 6 an insertion stored its ten fields one by one: a list of
    // seventeen copies is 170 store instructions of a wave that may have 64 memory operations in flight - the head waited
    // for its own stores to retire, a microsecond per insertion, 18 of the slowest head's 30 us on the `driven` workload,
    // tools/probes/timers_moves.py.)
    float wv[S];
    uint16_t trk[S];
    uint8_t lab[S], fg[S];
    {
      stv[0] = ST_TIMEPTC;
      tsv[0] = 0;
      wv[0] = 0.f;
      trk[0] = 0;
      lab[0] = 0;
      __builtin_memcpy(&wv[1], __builtin_assume_aligned(rec, 2), 4 * (S - 1));
      __builtin_memcpy(&tsv[1], rec + 4 * (S - 1), 2 * (S - 1));
      __builtin_memcpy(&trk[1], rec + 6 * (S - 1), 2 * (S - 1));
      __builtin_memcpy(&lab[1], rec + 8 * (S - 1), S - 1);
      __builtin_memcpy(&stv[1], rec + 9 * (S - 1), S - 1);
      __builtin_memcpy(fg, st.forget + base, S);
    }
    float px[S], py[S], pz[S];  // positions of the slots written (touched)
#pragma unroll
    for (int i = 0; i < S; ++i) px[i] = py[i] = pz[i] = 0.f;
    uint32_t touched = 0;
    // the voxel's owner entries and the length of the table of older memberships, with the rows above (owner_insert_local)
    uint16_t own[S];
    __builtin_memcpy(own, st.owner + base, 2 * S);
    // ... and the filter bits of its slots (owner_insert_local; n_alias: wave-uniform)
    uint32_t fbits = n_alias ? alias_filter_bits<S>(st, base) : 0u;
    // (an entry this frame did not write may carry the mark of an older frame: the row says who was first in this one)
    if (rk[0] == MV_NIL || rk[1] != t) continue;
    const uint32_t n_list = rk[0] + 1u;
    row[0] = MV_NIL;
    const uint32_t chain = rk[MV_ROW - 1];
    if (n_list > (uint32_t)MV_DIRECT) row[MV_ROW - 1] = MV_NIL;
    if (overflow) continue;
    uint32_t smax = a > b ? a : b;
    smax = smax > c ? smax : c;
    bool alias_touched = false;
    uint32_t n_ok = 0;
    bool full = false;
    // The vacant slots as a bit mask, kept up to date by the insertions (the lowest set bit is the "first vacant slot" of
    // operations.h:790-796).  A sparse wave pays for every instruction it issues, and the walk over status / stamp /
    // owner arrays per insertion was a few hundred of them per copy - most of this kernel's time.
    uint32_t vac = 0;
#pragma unroll
    for (int i = 1; i < S; ++i)
      if (stv[i] == ST_INVALID || (uint32_t)tsv[i] < smax) vac |= 1u << i;
    // one copy of the list, rank e, into the first vacant slot
    auto insert = [&](uint32_t e, const MoveCopy &cin) {
      const int slot = __ffs((int)vac) - 1;
      MoveCopy c = cin;
      if (e == t) c = c0;
      const uint8_t cs = c.status;
      const uint16_t cts = c.ts;
      uint16_t o = OWNER_NONE;
#pragma unroll
      for (int i = 1; i < S; ++i) o = i == slot ? own[i] : o;
      // the new index joins the object's set
      if (!owner_insert_local(st, base + slot, c.owner, o, n_alias, alias_touched, fbits, slot)) sc.cnt->overflow = 1;
#pragma unroll
      for (int i = 1; i < S; ++i) {  // the slot's row entries (registers: selects, no indexing)
        const bool here = i == slot;
        px[i] = here ? c.x : px[i];
        py[i] = here ? c.y : py[i];
        pz[i] = here ? c.z : pz[i];
        fg[i] = here ? (uint8_t)c.forget_bits : fg[i];
        wv[i] = here ? c.w : wv[i];
        tsv[i] = here ? cts : tsv[i];
        trk[i] = here ? c.track : trk[i];
        lab[i] = here ? c.label : lab[i];
        stv[i] = here ? cs : stv[i];
        own[i] = here ? o : own[i];
      }
      touched |= 1u << slot;
      // (a copy that is itself vacant - deleted before it was copied, or older than the slab's stamp - leaves the slot
      // to the next one)
      if (!(cs == ST_INVALID || (uint32_t)cts < smax)) vac &= ~(1u << slot);
      ++n_ok;
    };
    DBGM_MAX(2, n_list);
    DBGM_ADD(3, n_list);
    DBGM_ADD(4, 1);
    [[maybe_unused]] unsigned long long dbg_sel = 0, dbg_fetch = 0, dbg_ins = 0;
    if (n_list <= (uint32_t)MV_DIRECT) {
      // ---- the usual list: its ranks are in registers.  Sorted once, all copies requested together, replayed in order.
      [[maybe_unused]] const unsigned long long dbg_a = DBGM_T() + (rk[0] == 0xEEEEEEEEu ? 1 : 0);
      uint32_t e[MV_ROW];
#pragma unroll
      for (int j = 0; j < MV_ROW; ++j) e[j] = j < MV_DIRECT && (uint32_t)j < n_list ? rk[1 + j] : MV_NIL;  // (words beyond the count: older frames')
#pragma unroll
      for (int q = 0; q < 63; ++q) {
        const int i0 = RP_SORT16[q][0], i1 = RP_SORT16[q][1];
        const uint32_t lo = e[i0] < e[i1] ? e[i0] : e[i1], hi = e[i0] < e[i1] ? e[i1] : e[i0];
        e[i0] = lo;
        e[i1] = hi;
      }
      MoveCopy cc[MV_DIRECT];
#pragma unroll
      for (int j = 0; j < MV_DIRECT; ++j) cc[j] = sc.mv_copy[e[j] != MV_NIL ? e[j] : t];  // (no branch per fetch: each one was waited for where its branch ended)
      // (the fetches above are to be under way together before the first insertion: without the fence the compiler sinks
      // each fetch to the iteration that uses it - a dependent round trip per copy)
      __asm__ volatile("" ::: "memory");
      [[maybe_unused]] const unsigned long long dbg_b = DBGM_T() + (e[0] == 0xEEEEEEEEu ? 1 : 0);
      DBGM_MAX(0, dbg_a - dbg_t0);
      [[maybe_unused]] const unsigned long long dbg_c = DBGM_T() + (cc[0].voxel == 0xEEEEEEEEu ? 1 : 0) + (cc[MV_DIRECT - 1].voxel == 0xEEEEEEEEu ? 1 : 0);
#pragma unroll
      for (int j = 0; j < MV_DIRECT; ++j) {
        if (e[j] == MV_NIL || vac == 0u) break;  // through, or the voxel is full: the later copies are dropped (operations.h:357)
        insert(e[j], cc[j]);
      }
      [[maybe_unused]] const unsigned long long dbg_d = DBGM_T() + (n_ok == 0xEEEEu ? 1 : 0) + (vac == 0xEEEEEEEEu ? 1 : 0);
      dbg_sel = dbg_b - dbg_a;
      dbg_fetch = dbg_c - dbg_b;
      dbg_ins = dbg_d - dbg_c;
    } else {
      // ---- a long list: the row's ranks and the chain's go to LDS; batches of the next S-1 ranks are selected from there
      // (a list longer than RP_KEEP is gone through again, row and chain, for every batch)
      bool more = true;
      long long last = -1;  // largest rank replayed so far
      bool first_pass = true;
      while (more && !full) {
        [[maybe_unused]] const unsigned long long dbg_a = DBGM_T();
        uint32_t best[S - 1];  // the S-1 smallest ranks above `last`, ascending
#pragma unroll
        for (int i = 0; i < S - 1; ++i) best[i] = MV_NIL;
        uint32_t n_above = 0;  // ranks above `last` on the list
        auto offer = [&](uint32_t cur) {
          if ((long long)cur <= last) return;
          ++n_above;
          uint32_t x = cur;
#pragma unroll
          for (int i = 0; i < S - 1; ++i)
            if (x < best[i]) {
              const uint32_t y = best[i];
              best[i] = x;
              x = y;
            }
        };
        if (first_pass || n_list > (uint32_t)RP_KEEP) {
          uint32_t k = 0;
#pragma unroll
          for (int j = 0; j < MV_DIRECT; ++j) {
            if (first_pass) kept[j][threadIdx.x] = rk[1 + j];
            offer(rk[1 + j]);
          }
          k = MV_DIRECT;
          // (bounded by the count: a chain can only be as long as the arrivals the counter saw)
          for (uint32_t cur = chain; cur != MV_NIL && k < n_list; cur = sc.mv_next[cur]) {
            if (first_pass && k < (uint32_t)RP_KEEP) kept[k][threadIdx.x] = cur;
            ++k;
            offer(cur);
          }
          first_pass = false;
        } else {
          for (uint32_t k = 0; k < n_list; ++k) offer(kept[k][threadIdx.x]);
        }
        DBGM(1, 1, DBGM_T() + (stv[1] == 0xEE ? 1 : 0) + (own[1] == 0xEEEE ? 1 : 0));
        more = n_above > (uint32_t)(S - 1);  // ranks beyond this batch
        MoveCopy cc[S - 1];  // the batch's copies, requested together
#pragma unroll
        for (int u = 0; u < S - 1; ++u) cc[u] = sc.mv_copy[best[u] != MV_NIL ? best[u] : t];
        __asm__ volatile("" ::: "memory");
        [[maybe_unused]] const unsigned long long dbg_b = DBGM_T() + (best[0] == 0xEEEEEEEEu ? 1 : 0);
#pragma unroll
        for (int u = 0; u < S - 1; ++u) {
          const uint32_t e = best[u];
          if (e == MV_NIL || full) break;
          last = e;
          if (vac == 0u) {  // voxel full: this copy and all later ones are dropped (operations.h:357)
            full = true;
            break;
          }
          insert(e, cc[u]);
        }
        [[maybe_unused]] const unsigned long long dbg_d = DBGM_T() + (n_ok == 0xEEEEu ? 1 : 0) + (vac == 0xEEEEEEEEu ? 1 : 0);
        dbg_sel += dbg_b - dbg_a;
        dbg_ins += dbg_d - dbg_b;
      }
    }
    DBGM_MAX(6, dbg_sel);
    DBGM_MAX(7, dbg_fetch);
    DBGM_MAX(8, dbg_ins);
    DBGM(1, 2, DBGM_T());
    DBGM(1, 3, n_ok);
    DBGM_MAX(1, DBGM_T() - dbg_t0 + (n_ok == 0xEEEE ? 1 : 0));
    DBGM_MAX(5, n_ok);
    if (n_ok) {  // the voxel's rows, once
      {
        constexpr int L = S - 1, NW = (10 * L + 3) / 4;
        uint32_t dw[NW];
#pragma unroll
        for (int i = 0; i < NW; ++i) dw[i] = 0;
#pragma unroll
        for (int k = 0; k < L; ++k) {
          dw[k] = __float_as_uint(wv[k + 1]);
          dw[(4 * L + 2 * k) >> 2] |= (uint32_t)tsv[k + 1] << (((4 * L + 2 * k) & 3) * 8);
          dw[(6 * L + 2 * k) >> 2] |= (uint32_t)trk[k + 1] << (((6 * L + 2 * k) & 3) * 8);
          dw[(8 * L + k) >> 2] |= (uint32_t)lab[k + 1] << (((8 * L + k) & 3) * 8);
          dw[(9 * L + k) >> 2] |= (uint32_t)stv[k + 1] << (((9 * L + k) & 3) * 8);
        }
        __builtin_memcpy(__builtin_assume_aligned(rec, 2), dw, 10 * L);
      }
      __builtin_memcpy(st.owner + base, own, 2 * S);
      __builtin_memcpy(st.forget + base, fg, S);
#pragma unroll
      for (int i = 1; i < S; ++i)
        if ((touched >> i) & 1u) st.pos4[base + i] = make_float4(px[i], py[i], pz[i], 0.f);
      flag_owner_chunk(st, base + 1);  // (a voxel's slots lie in one chunk)
      st.vflag[lv] = VF_DIRTY;
      mark_tile(st, lv, epoch);
      atomicAdd(&n_ok_block, n_ok);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && n_ok_block) atomicAdd(&sc.cnt->n_move_reinserted, n_ok_block);
  DBGM(1, 2, DBGM_T());  // (the workgroup's end: the stamp above is thread 0's own list only)
}

// removeObjectByTrackID (object_layer.h:414-425): every index of the set -> INVALID, set erased.
__global__ __launch_bounds__(TPB) void k_remove(State st, size_t n_slots, const FrameArgs *__restrict__ fa, int p_n) {
  const int n = fa->n_remove;
  if (n <= 0) return;  // nothing to wipe in this frame
  const uint32_t epoch = fa->f.epoch;
  const uint16_t *__restrict__ tracks = fa->remove;
  if (blockIdx.x == 0) {  // older memberships of the removed objects (State::alias)
    const uint32_t na = st.alias[0] < st.alias_cap ? st.alias[0] : st.alias_cap;
    for (uint32_t k = threadIdx.x; k < na; k += blockDim.x) {
      const uint32_t trk = st.alias[3 + 2 * k];
      if (trk == OWNER_NONE) continue;
      for (int q = 0; q < n; ++q)
        if (tracks[q] == trk) {
          slot_ref_li(st, p_n, st.alias[2 + 2 * k]).set_status(ST_INVALID);
          st.vflag[st.alias[2 + 2 * k] >> p_n] = VF_DIRTY;
          mark_tile(st, st.alias[2 + 2 * k] >> p_n, epoch);
          st.alias[3 + 2 * k] = OWNER_NONE;
          break;
        }
    }
  }
  // workgroups stride over the chunks of OWNER_CHUNK slots (the grid does not depend on the frame)
  const size_t n_chunks = (n_slots + OWNER_CHUNK - 1) / OWNER_CHUNK;
  for (size_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    if (st.owner_flag[chunk] == 0) continue;
    size_t i = chunk * OWNER_CHUNK + threadIdx.x;
    size_t end = (chunk + 1) * OWNER_CHUNK;
    if (end > n_slots) end = n_slots;
    for (; i < end; i += blockDim.x) {
      uint16_t o = st.owner[i];
      if (o == OWNER_NONE || (i & (((size_t)1 << p_n) - 1)) == 0) continue;  // (slot 0, the time particle, is in nobody's set)
      for (int k = 0; k < n; ++k)
        if (tracks[k] == o) {
          slot_ref_li(st, p_n, i).set_status(ST_INVALID);
          st.vflag[i >> p_n] = VF_DIRTY;
          mark_tile(st, i >> p_n, epoch);
          st.owner[i] = OWNER_NONE;
          break;
        }
    }
  }
}

// after sdm_load_state: recompute the chunk flags (both levels; the coarse one was zeroed by the launcher) from the owner array
__global__ __launch_bounds__(TPB) void k_owner_flags(const uint16_t *__restrict__ owner, size_t n_slots,
                                                     uint8_t *__restrict__ owner_flag, uint8_t *__restrict__ owner_flag2) {
  __shared__ uint32_t any_owner;
  if (threadIdx.x == 0) any_owner = 0;
  __syncthreads();
  size_t i = (size_t)blockIdx.x * OWNER_CHUNK + threadIdx.x;
  size_t end = (size_t)(blockIdx.x + 1) * OWNER_CHUNK;
  if (end > n_slots) end = n_slots;
  bool a = false;
  for (; i < end; i += blockDim.x) a = a || owner[i] != OWNER_NONE;
  if (a) any_owner = 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    owner_flag[blockIdx.x] = any_owner ? 1 : 0;
    if (any_owner) owner_flag2[blockIdx.x / OWNER_GROUP] = 1;
  }
}

// sdm_tracks_with_particles: one bit per track id that owns at least one slot (primary owner or an older membership the
// reference's sets still hold) - the keys of ObjectParticleHashMap::indices_map with a non-empty set
// (object_layer.h:20-52, semantic_dsp_map.h:712-736).  bitmap: 2048 uint32, zeroed by the launcher.
__global__ __launch_bounds__(TPB) void k_tracks_with_particles(State st, size_t n_slots, uint32_t *__restrict__ bitmap) {
  // one workgroup per group of OWNER_GROUP chunks: the group's flag, then its chunks' flags in one round, then only the
  // chunks that may hold owners (round 4: a workgroup walked 32 chunk flags one dependent load after the other, 77 us)
  __shared__ uint32_t seen[2048];
  __shared__ uint32_t list[OWNER_GROUP];
  __shared__ uint32_t n_list;
  const size_t n_chunks = (n_slots + OWNER_CHUNK - 1) / OWNER_CHUNK;
  const bool group_set = st.owner_flag2[blockIdx.x] != 0;
  if (!group_set && blockIdx.x != 0) return;  // (workgroup-uniform)
  for (uint32_t k = threadIdx.x; k < 2048; k += TPB) seen[k] = 0;
  if (threadIdx.x == 0) n_list = 0;
  __syncthreads();
  if (blockIdx.x == 0) {
    const uint32_t na = st.alias[0] < st.alias_cap ? st.alias[0] : st.alias_cap;
    for (uint32_t k = threadIdx.x; k < na; k += TPB) {
      const uint32_t trk = st.alias[3 + 2 * k];
      if (trk != OWNER_NONE) atomicOr(&seen[trk >> 5], 1u << (trk & 31u));
    }
  }
  static_assert(OWNER_GROUP == 64, "one wave reads a group's chunk flags");
  if (group_set && threadIdx.x < OWNER_GROUP) {
    const size_t chunk = (size_t)blockIdx.x * OWNER_GROUP + threadIdx.x;
    const bool f = chunk < n_chunks && st.owner_flag[chunk] != 0;
    const unsigned long long b = __ballot(f);
    if (f) list[__popcll(b & ((1ull << threadIdx.x) - 1ull))] = (uint32_t)chunk;
    if (threadIdx.x == 0) n_list = (uint32_t)__popcll(b);
  }
  __syncthreads();
  const uint32_t nl = n_list;
  for (uint32_t q = 0; q < nl; ++q) {
    const size_t c0 = (size_t)list[q] * OWNER_CHUNK;
    constexpr int PER = OWNER_CHUNK / TPB;
    uint16_t o[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const size_t i = c0 + threadIdx.x + (size_t)j * TPB;
      o[j] = i < n_slots ? st.owner[i] : OWNER_NONE;
    }
#pragma unroll
    for (int j = 0; j < PER; ++j)
      if (o[j] != OWNER_NONE && !((seen[o[j] >> 5] >> (o[j] & 31u)) & 1u)) atomicOr(&seen[o[j] >> 5], 1u << (o[j] & 31u));
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < 2048; k += TPB)
    if (seen[k]) atomicOr(&bitmap[k], seen[k]);
}

inline unsigned blocks_for(size_t n, int tpb = TPB) { return (unsigned)((n + tpb - 1) / tpb); }

}  // namespace

void launch_owner_flags(const Dims &d, const State &st, hipStream_t s) {
  const size_t n_slots = (size_t)d.v_count * d.S;
  hipMemsetAsync(st.owner_flag2, 0, owner_flag2_bytes(n_slots), s);
  hipLaunchKernelGGL(k_owner_flags, dim3((unsigned)((n_slots + OWNER_CHUNK - 1) / OWNER_CHUNK)), dim3(TPB), 0, s, st.owner, n_slots,
                     st.owner_flag, st.owner_flag2);
}

void launch_tracks_with_particles(const Dims &d, const State &st, uint32_t *bitmap, hipStream_t s) {
  const size_t n_slots = (size_t)d.v_count * d.S;
  hipMemsetAsync(bitmap, 0, 2048 * sizeof(uint32_t), s);
  const size_t n_chunks = (n_slots + OWNER_CHUNK - 1) / OWNER_CHUNK;
  hipLaunchKernelGGL(k_tracks_with_particles, dim3((unsigned)((n_chunks + OWNER_GROUP - 1) / OWNER_GROUP)), dim3(TPB), 0, s, st, n_slots, bitmap);
}

size_t move_blocks(const Dims &d) { return ((size_t)d.v_count * d.S + MV_CHUNK - 1) / MV_CHUNK; }
size_t move_count_elems() { return (size_t)MAX_MOVE_OBJECTS * MV_LIST_CAP + 1; }
// member lists: one stretch of MV_CHUNK entries per listed chunk.  A map cannot list more chunks than it has; k_move_apply's
// 1024 workgroups request their first position's stretch before they know the list's length, so at least that many
// stretches exist (128 MiB for every map until round 5, 16 MiB for the small ones now).
size_t move_member_elems(size_t n_slots) {
  const size_t n_chunks = (n_slots + MV_CHUNK - 1) / MV_CHUNK;
  return std::min<size_t>(MV_LIST_CAP, std::max<size_t>(n_chunks, 1024)) * MV_CHUNK;
}
size_t move_total_elems() { return (size_t)2 * MAX_MOVE_OBJECTS * MV_TOT_STRIDE; }

// The launch sequence below is the same every frame (hipGraph): kernels of a frame without moving objects / removals
// return at once.
// step 1: list and rank every moving object's members (ascending index) and publish the per-object counts
// by_value != nullptr: the launch runs ahead of the frame's k_frame_begin on a stream of its own; it takes the frame block
// by value and stores it in fa_moves for the chain's other kernel
static MembersArgs members_args(const Dims &d, const Scratch &sc) {
  MembersArgs ma;
  ma.mv_cnt = sc.mv_cnt;
  ma.mv_list = sc.mv_list;
  ma.mv_nlist = sc.mv_nlist;
  ma.mv_nmem = sc.mv_nmem;
  ma.mv_mem = sc.mv_mem;
  ma.mv_tot = sc.mv_tot;
  ma.cur = sc.cur;
  ma.n_flags = (uint32_t)move_blocks(d);
  ma.n_slots = (size_t)d.v_count * d.S;
  return ma;
}
void launch_moves_count(const Dims &d, const State &st, const Scratch &sc, int32_t *counts_local, hipStream_t s, const FrameArgs *by_value) {
  const FrameArgs *fa = by_value ? sc.fa_moves : sc.fa_side;
  const MembersArgs ma = members_args(d, sc);
  if (by_value)
    hipLaunchKernelGGL(k_move_members_v, dim3(MV_GRID), dim3(TPB), 0, s, st, ma, *by_value, const_cast<FrameArgs *>(sc.fa_moves));
  else
    hipLaunchKernelGGL(k_move_members, dim3(MV_GRID), dim3(TPB), 0, s, st, ma, fa);
  // the per-object counts are only needed as a separate row when they are exchanged between shards
  if (d.v_count != d.V) hipLaunchKernelGGL(k_move_local_counts, dim3(1), dim3(HALO_OBJ), 0, s, counts_local, sc, fa, 1);
}

// the next batch of a long object list: the frame block with that batch's objects replaces the main block, and the member
// count of those objects runs - in the same launch (k_move_members_v: one block stores the block, all count)
// A Z-slab shard also publishes the batch's per-object counts (counts_local: exchanged with the other shards before the
// batch's k_move_apply, like the first batch's).
void launch_moves_batch(const Dims &d, const State &st, const Scratch &sc, const FrameArgs &fa_batch, int32_t *counts_local, hipStream_t s) {
  const MembersArgs ma = members_args(d, sc);
  hipLaunchKernelGGL(k_move_members_v, dim3(MV_GRID), dim3(TPB), 0, s, st, ma, fa_batch, const_cast<FrameArgs *>(sc.fa));
  if (d.v_count != d.V) hipLaunchKernelGGL(k_move_local_counts, dim3(1), dim3(HALO_OBJ), 0, s, counts_local, sc, sc.fa, 0);
}

// arguments of k_frame_begin in the order of its parameter list; `fa` is the frame block that goes by value
void FrameBeginLaunch::set(const Dims &d_, const State &st_, const Scratch &sc, const FrameArgs &fa_, bool with_side, bool with_members_) {
  cnt = sc.cnt;
  bin_count = sc.bin_count;
  n_bins = (uint32_t)(d_.W * d_.H + 1 + d_.H * (int)(ROW_SUBS * ROW_CNT_STRIDE));  // the per-pixel counts and, behind them, the per-row list counts
  st = st_;
  fa = fa_;
  dst_main = const_cast<FrameArgs *>(sc.fa);
  dst_side = with_side ? const_cast<FrameArgs *>(sc.fa_side) : nullptr;
  d = d_;
  slab_max = d_.NY * d_.NZ;  // voxels of the largest slab a ring shift can re-stamp
  if (d_.NX * d_.NZ > slab_max) slab_max = d_.NX * d_.NZ;
  if (d_.NX * d_.NY > slab_max) slab_max = d_.NX * d_.NY;
  ma = members_args(d_, sc);
  with_members = with_members_ ? 1 : 0;
  argv[0] = &cnt;
  argv[1] = &bin_count;
  argv[2] = &n_bins;
  argv[3] = &st;
  argv[4] = &fa;
  argv[5] = &dst_main;
  argv[6] = &dst_side;
  argv[7] = &d;
  argv[8] = &slab_max;
  argv[9] = &ma;
  argv[10] = &with_members;
}
const void *FrameBeginLaunch::kernel() { return reinterpret_cast<const void *>(k_frame_begin); }

void launch_frame_begin(FrameBeginLaunch &a, hipStream_t s, hipEvent_t done) {
  if (done) (void)hipExtLaunchKernel(FrameBeginLaunch::kernel(), dim3(FrameBeginLaunch::GRID), dim3(FrameBeginLaunch::BLOCK), a.argv, 0, s, nullptr, done, 0);
  else (void)hipLaunchKernel(FrameBeginLaunch::kernel(), dim3(FrameBeginLaunch::GRID), dim3(FrameBeginLaunch::BLOCK), a.argv, 0, s);
}

// step 2 (after the counts of all shards are known): global ranks, transform, export of slab-crossing copies
void launch_moves_transform(const Dims &d, const Filter &flt, const State &st, const Scratch &sc, const int32_t *counts_all, int world,
                            int rank, hipStream_t s) {
  // (1024 workgroups: move_member_elems counts on it)
  hipLaunchKernelGGL(k_move_apply, dim3(1024), dim3(TPB), 0, s, d, flt, st, sc, d.v_count != d.V ? counts_all : nullptr, world, rank);
}

// step 3 (after the export buffers of all shards are gathered): import, ordered replay per target voxel
void launch_moves_finish(const Dims &d, const Filter &flt, const State &st, const Scratch &sc, int world, int rank, hipStream_t s) {
  if (world > 1 && sc.halo_recv) hipLaunchKernelGGL(k_move_import, dim3(64, world), dim3(TPB), 0, s, d, sc, world, rank);
  dim3 grid(RP_GRID);
  switch (d.p_n) {
    case 1: hipLaunchKernelGGL(k_move_replay<2>, grid, dim3(RP_TPB), 0, s, d, flt, st, sc); break;
    case 2: hipLaunchKernelGGL(k_move_replay<4>, grid, dim3(RP_TPB), 0, s, d, flt, st, sc); break;
    case 3: hipLaunchKernelGGL(k_move_replay<8>, grid, dim3(RP_TPB), 0, s, d, flt, st, sc); break;
    default: hipLaunchKernelGGL(k_move_replay<16>, grid, dim3(RP_TPB), 0, s, d, flt, st, sc); break;
  }
}

void launch_remove(const Dims &d, const State &st, const Scratch &sc, hipStream_t s) {
  const size_t n_slots = (size_t)d.v_count * d.S;
  const size_t n_chunks = (n_slots + OWNER_CHUNK - 1) / OWNER_CHUNK;
  hipLaunchKernelGGL(k_remove, dim3((unsigned)(n_chunks < 1024 ? n_chunks : 1024)), dim3(TPB), 0, s, st, n_slots, sc.fa, d.p_n);
}

}  // namespace sdm
