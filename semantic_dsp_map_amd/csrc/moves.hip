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
#include "sdm_internal.h"
#include "sdm_scratch.h"

#pragma clang fp contract(off)

namespace sdm {

#ifdef SDM_AB_TIMERS
__device__ unsigned long long g_dbg_m[2][4096 * 4];  // [kernel][workgroup][checkpoint], see kernels.hip
void debug_timers_moves(unsigned long long *out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg_m), sizeof(g_dbg_m));
  if (reset) {
    static unsigned long long z[2][4096 * 4];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_m), z, sizeof(z));
  }
}
#define DBGM(k, i, v) do { if (blockIdx.x < 4096 && threadIdx.x == 0) g_dbg_m[k][blockIdx.x * 4 + (i)] = (unsigned long long)(v); } while (0)
#define DBGM_T() wall_clock64()
#else
#define DBGM(k, i, v)
#define DBGM_T() 0ull
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

// pass 0: ascending list of the chunks whose owner_flag is set (one workgroup; each thread takes 32 consecutive
// flag bytes, one block-wide scan of the per-thread counts per 32 K flags).  Dynamic objects touch a few hundred of
// the map's tens of thousands of chunks; everything after this works on the list only.
__device__ __forceinline__ void move_chunks_body(const uint8_t *owner_flag, uint32_t n_flags, uint32_t *__restrict__ list,
                                                 uint32_t *__restrict__ n_list, Cursors *cur, uint32_t *__restrict__ cnt, int n_obj,
                                                 uint32_t n_move_cnt, uint32_t *__restrict__ alias, uint8_t *owner_flag_w) {
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t running, alias_live;
  if (n_obj <= 0) return;  // no object moves in this frame
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  constexpr uint32_t PER = 32;
  // the first 32 K flags are requested before anything else is looked at (the table of older memberships below is a
  // dependent fetch of its own, and nearly always empty); if it does set flags, they are read again
  const bool early = threadIdx.x * PER + PER <= n_flags && ((size_t)(owner_flag + threadIdx.x * PER) & 15) == 0;
  uint4 ea = make_uint4(0, 0, 0, 0), eb = make_uint4(0, 0, 0, 0);
  if (early) {
    ea = *reinterpret_cast<const uint4 *>(owner_flag + threadIdx.x * PER);
    eb = *reinterpret_cast<const uint4 *>(owner_flag + threadIdx.x * PER + 16);
  }
  if (threadIdx.x == 0) {
    running = 0;
    alias_live = 0;
    cur->move_list_overflow = 0;
    // extra set memberships (State::alias): drop the deleted entries, and make sure the chunks of the live ones are
    // on the list even if no slot of theirs has a primary owner any more
    uint32_t na = alias[0], keep = 0;
    if (na > ALIAS_CAP) {
      na = ALIAS_CAP;
      cur->move_list_overflow = 1;
    }
    for (uint32_t k = 0; k < na; ++k) {
      const uint32_t idx = alias[2 + 2 * k], trk = alias[3 + 2 * k];
      if (trk == OWNER_NONE) continue;
      alias[2 + 2 * keep] = idx;
      alias[3 + 2 * keep] = trk;
      ++keep;
      owner_flag_w[idx / OWNER_CHUNK] = 1;
      alias_live = 1;
    }
    for (uint32_t k = keep; k < na; ++k) {
      alias[2 + 2 * k] = INVALID_INDEX;
      alias[3 + 2 * k] = OWNER_NONE;
    }
    alias[0] = keep;
    __threadfence();
  }
  __syncthreads();
  const bool reuse_early = alias_live == 0;
  for (uint32_t tile = 0; tile < n_flags; tile += 1024 * PER) {
    const uint32_t first = tile + threadIdx.x * PER;
    uint32_t mask = 0;  // bit j: flag first + j is set
    if (first + PER <= n_flags && ((size_t)(owner_flag + first) & 15) == 0) {
      uint4 a = ea, b4 = eb;
      if (tile != 0) {
        a = *reinterpret_cast<const uint4 *>(owner_flag + first);
        b4 = *reinterpret_cast<const uint4 *>(owner_flag + first + 16);
      } else if (!reuse_early) {  // thread 0 has just set flags: past this CU's L1, which holds the lines as they were
        uint32_t *wa = reinterpret_cast<uint32_t *>(&a), *wb = reinterpret_cast<uint32_t *>(&b4);
        const uint32_t *src = reinterpret_cast<const uint32_t *>(owner_flag + first);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          wa[q] = __hip_atomic_load(src + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          wb[q] = __hip_atomic_load(src + 4 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      const uint32_t w[8] = {a.x, a.y, a.z, a.w, b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if ((w[q] >> (8 * j)) & 0xffu) mask |= 1u << (4 * q + j);
    } else {
      for (uint32_t j = 0; j < PER; ++j)
        if (first + j < n_flags && owner_flag[first + j]) mask |= 1u << j;
    }
    const uint32_t c = (uint32_t)__popc(mask);
    // block exclusive scan of c
    uint32_t inc = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t nb = __shfl_up(inc, off, 64);
      if (lane >= off) inc += nb;
    }
    if (lane == 63) wave_tot[wid] = inc;
    __syncthreads();
    uint32_t base = running;
    for (int w2 = 0; w2 < wid; ++w2) base += wave_tot[w2];
    uint32_t pos = base + inc - c;
    uint32_t mm = mask;
    while (mm) {
      const int j = __ffs((int)mm) - 1;
      mm &= mm - 1;
      if (pos < MV_LIST_CAP) list[pos] = first + (uint32_t)j;
      else cur->move_list_overflow = 1;
      ++pos;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t t = 0;
      for (int w2 = 0; w2 < 16; ++w2) t += wave_tot[w2];
      running += t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    // The count matrix has one row per object and one column per listed chunk - its row length is the list's length,
    // not the list's capacity: the scan that turns it into offsets covers n_obj x n + 1 elements (2 K at the benchmark's
    // 6 objects x 347 chunks, where rows of MV_LIST_CAP were 49 K, most of them zeros somebody had to write first).
    const uint32_t nl = running < MV_LIST_CAP ? running : MV_LIST_CAP;
    n_list[0] = nl;
    n_list[2] = (uint32_t)n_obj * nl + 1u;   // what the scan covers (read on the device)
    cnt[(size_t)n_obj * nl] = 0;             // terminator of the count matrix (becomes the grand total after the scan)
  }
}
__global__ __launch_bounds__(1024) void k_move_chunks(const uint8_t *owner_flag, uint32_t n_flags,
                                                      uint32_t *__restrict__ list, uint32_t *__restrict__ n_list, Cursors *cur,
                                                      uint32_t *__restrict__ cnt, const FrameArgs *__restrict__ fa,
                                                      uint32_t *__restrict__ alias, uint8_t *owner_flag_w) {
  move_chunks_body(owner_flag, n_flags, list, n_list, cur, cnt, fa->n_obj, fa->n_move_cnt, alias, owner_flag_w);
}
// The same as the first kernel of a chain of its own (the member count of a launch-by-launch frame starts behind the
// PREVIOUS frame's births, on its own stream): the frame block arrives by value and is stored for the chain's other
// kernels - no k_set_frame launch and no event from another stream in front of the chain.
__global__ __launch_bounds__(1024) void k_move_chunks_v(const uint8_t *owner_flag, uint32_t n_flags,
                                                        uint32_t *__restrict__ list, uint32_t *__restrict__ n_list, Cursors *cur,
                                                        uint32_t *__restrict__ cnt, const FrameArgs src, FrameArgs *__restrict__ dst,
                                                        uint32_t *__restrict__ alias, uint8_t *owner_flag_w) {
  const uint32_t *s4 = reinterpret_cast<const uint32_t *>(&src);
  uint32_t *d4 = reinterpret_cast<uint32_t *>(dst);
  for (uint32_t i = threadIdx.x; i < sizeof(FrameArgs) / 4; i += blockDim.x) d4[i] = s4[i];
  move_chunks_body(owner_flag, n_flags, list, n_list, cur, cnt, src.n_obj, src.n_move_cnt, alias, owner_flag_w);
}

// pass 1: per-chunk, per-object member counts.  cnt[obj * n + list position], n = chunks on the list.  A flagged chunk that turns out
// to hold no owner any more clears its flag.
__global__ __launch_bounds__(TPB) void k_move_count(const uint16_t *__restrict__ owner, size_t n_slots,
                                                    const FrameArgs *__restrict__ fa, uint32_t *__restrict__ cnt,
                                                    uint8_t *__restrict__ owner_flag, const uint32_t *__restrict__ list,
                                                    const uint32_t *__restrict__ n_list, const uint32_t *__restrict__ alias) {
  __shared__ uint32_t c[MAX_MOVE_OBJECTS];
  __shared__ uint16_t tracks[MAX_MOVE_OBJECTS];
  __shared__ uint32_t any_owner;
  const int n_obj = fa->n_obj;
  if (n_obj <= 0) return;
  const MoveSet &ms = fa->ms;
  const uint32_t n = *n_list;
  if (threadIdx.x < MAX_MOVE_OBJECTS) tracks[threadIdx.x] = (int)threadIdx.x < n_obj ? ms.track[threadIdx.x] : OWNER_NONE;
  for (uint32_t pos = blockIdx.x; pos < n; pos += gridDim.x) {
    __syncthreads();
    if (threadIdx.x < MAX_MOVE_OBJECTS) c[threadIdx.x] = 0;
    if (threadIdx.x == 0) any_owner = 0;
    __syncthreads();
    const uint32_t chunk = list[pos];
    size_t base = (size_t)chunk * MV_CHUNK;
    // (all 16 owner loads of the thread in flight, then the counting: four at a time were four dependent round trips)
    uint16_t ow[MV_ITEMS];
#pragma unroll
    for (int r = 0; r < MV_ITEMS; ++r) {
      const size_t i = base + (size_t)r * TPB + threadIdx.x;
      ow[r] = i < n_slots ? owner[i] : OWNER_NONE;
    }
#pragma unroll
    for (int r = 0; r < MV_ITEMS; ++r) {
      if (ow[r] != OWNER_NONE) any_owner = 1;
      const uint8_t o = obj_of(ow[r], tracks, n_obj);
      if (o != 0xFF) atomicAdd(&c[o], 1u);
    }
    {  // older memberships the reference's sets still hold (State::alias; nearly always none)
      const uint32_t na = alias[0] < ALIAS_CAP ? alias[0] : ALIAS_CAP;
      for (uint32_t k = threadIdx.x; k < na; k += blockDim.x) {
        const uint32_t idx = alias[2 + 2 * k], trk = alias[3 + 2 * k];
        if (trk == OWNER_NONE || idx / MV_CHUNK != chunk) continue;
        any_owner = 1;
        const uint8_t o = obj_of((uint16_t)trk, tracks, n_obj);
        if (o != 0xFF) atomicAdd(&c[o], 1u);
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < n_obj) cnt[(size_t)threadIdx.x * n + pos] = c[threadIdx.x];
    if (threadIdx.x == 0 && any_owner == 0) owner_flag[chunk] = 0;
  }
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

__global__ void k_move_local_counts(const uint32_t *__restrict__ offs, int32_t *counts_local, Scratch sc, const FrameArgs *__restrict__ fa) {
  int k = threadIdx.x;
  const int n_obj = fa->n_obj;  // (runs with the member count, on its stream)
  // this frame's export counters, one per destination shard
  const uint32_t world = sc.halo_world;
  if (sc.halo_send && (uint32_t)k < world) *reinterpret_cast<uint32_t *>(sc.halo_send + (size_t)k * halo_segment_bytes(sc.halo_cap)) = 0;
  if (k >= HALO_OBJ) return;
  const uint32_t nl = *sc.mv_nlist;  // row length of the count matrix
  counts_local[k] = k < n_obj ? (int32_t)(offs[(size_t)(k + 1) * nl] - offs[(size_t)k * nl]) : 0;
}

constexpr uint32_t MV_NIL = 0xffffffffu;

// A moved copy of global rank e joins the list of its target voxel (push-front; the replay restores rank order).
// The copy that finds the list idle also enters the voxel in this frame's work list.
// (The work list's cursor is ONE word: an atomic per touched voxel on it retires at ~12 ns each, a few thousand per frame -
// that was most of k_move_apply's time.  A workgroup therefore collects its new voxels in LDS - vox_list / vox_n - and
// reserves their places with one atomic, flush_move_voxels; callers without such a list pass nullptr.)
__device__ __forceinline__ void move_link(const Dims &d, const Scratch &sc, uint32_t v, uint32_t e, uint32_t *vox_list = nullptr,
                                          uint32_t *vox_n = nullptr) {
  const uint32_t lv = v - d.v_begin;
  const uint32_t prev = atomicExch(&sc.mv_head[lv], e);
  sc.mv_next[e] = prev;
  if (prev == MV_NIL) {
    if (vox_list) vox_list[atomicAdd(vox_n, 1u)] = lv;
    else sc.mv_vlist[atomicAdd(&sc.cnt->n_move_voxels, 1u)] = lv;
  }
}
// all threads of the workgroup; vox_base is an LDS word
__device__ __forceinline__ void flush_move_voxels(const Scratch &sc, const uint32_t *vox_list, uint32_t *vox_n, uint32_t *vox_base) {
  __syncthreads();
  const uint32_t n = *vox_n;
  if (threadIdx.x == 0 && n) *vox_base = atomicAdd(&sc.cnt->n_move_voxels, n);
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) sc.mv_vlist[*vox_base + i] = vox_list[i];
  __syncthreads();
  if (threadIdx.x == 0) *vox_n = 0;
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
__device__ __forceinline__ MoveLoaded move_load(const Dims &d, const Filter &flt, const State &st, long long cursor, uint32_t e,
                                                size_t li, bool copy_invalid) {
  MoveLoaded m;
  m.p = st.pos4[li];
  const long long draw = cursor + 3ll * e;
  m.n[0] = st.noise[(draw + 1) % flt.noise_n];
  m.n[1] = st.noise[(draw + 2) % flt.noise_n];
  m.n[2] = st.noise[(draw + 3) % flt.noise_n];
  m.w = st.w[rec_index(li, d.p_n, REC_W)];
  m.ts = st.ts[rec_index(li, d.p_n, REC_TS)];
  m.track = st.track[rec_index(li, d.p_n, REC_TRACK)];
  m.label = st.label[rec_index(li, d.p_n, REC_LABEL)];
  m.status = copy_invalid ? (uint8_t)ST_INVALID : st.status[rec_index(li, d.p_n, REC_STATUS)];
  return m;
}
__device__ __forceinline__ void move_store(const Dims &d, const Frame &f, const MoveSet &ms, const State &st, const Scratch &sc,
                                           const MoveLoaded &m, int obj, uint32_t e, size_t li, bool alias, uint32_t *vox_list,
                                           uint32_t *vox_n) {
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
  st.status[rec_index(li, d.p_n, REC_STATUS)] = ST_INVALID;  // deleteParticleByIndex
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
    c.forget_bits = __float_as_uint(p.w);
    c.w = pw;
    c.ts = pts;
    c.track = ptrack;
    c.owner = powner;
    c.label = plabel;
    c.status = pstatus;
    c.pad = 0;
    sc.mv_copy[e] = c;
    move_link(d, sc, v, e, vox_list, vox_n);
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
    }
  }
}
__device__ __forceinline__ void move_one(const Dims &d, const Frame &f, const Filter &flt, const MoveSet &ms, const State &st,
                                         const Scratch &sc, int obj, uint32_t e, size_t li, bool alias = false,
                                         bool copy_invalid = false, uint32_t *vox_list = nullptr, uint32_t *vox_n = nullptr) {
  const MoveLoaded m = move_load(d, flt, st, (long long)sc.cur->move_cursor, e, li, copy_invalid);
  move_store(d, f, ms, st, sc, m, obj, e, li, alias, vox_list, vox_n);
}

// One round (TPB consecutive slots) of k_move_apply in a chunk that holds older set memberships (State::alias): a slot can
// then belong to several moving objects.  mo[0] is its primary membership (owner[]), mo[1..] the older ones.
constexpr uint32_t CA_CAP = 512;
__device__ __forceinline__ void move_round_with_aliases(const Dims &d, const Frame &f, const Filter &flt, const MoveSet &ms,
                                                     const State &st, const Scratch &sc, size_t li, size_t n_slots, int n_obj,
                                                     const uint16_t *tracks, const uint32_t *obj_base,
                                                     uint32_t (*wave_cnt)[MAX_MOVE_OBJECTS], const uint32_t *ca_idx,
                                                     const uint32_t *ca_ent, const uint8_t *ca_obj, uint32_t n_ca, uint64_t lt_mask,
                                                     int wid) {
  constexpr int MAXM = 4;
  uint8_t mo[MAXM] = {0xFF, 0xFF, 0xFF, 0xFF};
  uint32_t ment[MAXM] = {0, 0, 0, 0}, mrank[MAXM] = {0, 0, 0, 0};
  if (li < n_slots) mo[0] = obj_of(st.owner[li], tracks, n_obj);
  int nm = 1;
  for (uint32_t c = 0; c < n_ca; ++c)
    if (ca_idx[c] == (uint32_t)li) {
      if (nm < MAXM) {
        mo[nm] = ca_obj[c];
        ment[nm] = ca_ent[c];
        ++nm;
      } else {
        sc.cnt->overflow = 1;  // a slot in more than four moving sets at once: not handled
      }
    }
  // members of object k in this wave = lanes one of whose memberships is k, in lane (= index) order.  (This path is
  // rare; one ballot per moving object keeps it light on registers.)
  for (int k = 0; k < n_obj; ++k) {
    int q = -1;
#pragma unroll
    for (int c = 0; c < MAXM; ++c)
      if (mo[c] == (uint8_t)k) q = c;
    const uint64_t mm = __ballot(q >= 0);
    if (q >= 0) {
      mrank[q] = (uint32_t)__popcll(mm & lt_mask);
      if (mrank[q] == 0) wave_cnt[wid][k] = (uint32_t)__popcll(mm);
    }
  }
  __syncthreads();
  // a slot that several moving objects hold is moved by each of them, in object order: all but the first copy a
  // particle that has just been invalidated
  bool first = true;
  for (int done_n = 0; done_n < MAXM; ++done_n) {
    int q = -1;
#pragma unroll
    for (int c = 0; c < MAXM; ++c)
      if (mo[c] != 0xFF && (q < 0 || mo[c] < mo[q])) q = c;
    if (q < 0) break;
    const uint8_t ob = mo[q];
    uint32_t e = obj_base[ob] + mrank[q];
#pragma unroll
    for (int w = 0; w < MV_WAVES; ++w)
      if (w < wid) e += wave_cnt[w][ob];
    move_one(d, f, flt, ms, st, sc, (int)ob, e, li, q != 0, !first);
    if (q != 0) st.alias[3 + 2 * ment[q]] = OWNER_NONE;  // the object's set is rebuilt from its re-inserted copies
    mo[q] = 0xFF;
    first = false;
  }
}

// phase 1 of moveParticlesInSetsByTransformations (operations.h:331-349): copy, transform + table noise, delete the
// original.  One workgroup per flagged chunk: the members of every moving object are ranked in ascending index order
// (ballot ranks inside a wave, wave counts through LDS, chunk offsets from the scanned count matrix), which gives each
// its global rank e = rank among all members of all moving objects in (object, shard, index) order.  The noise
// cursor advances by three per particle in that order; the copy joins its target voxel's list or is exported.
__global__ __launch_bounds__(TPB) void k_move_apply(Dims d, Filter flt, State st, Scratch sc, const uint32_t *__restrict__ offs,
                                                    const int32_t *__restrict__ counts_all, int world, int rank) {
  const int n_obj = sc.fa->n_obj;
  if (n_obj <= 0) return;
  DBGM(0, 0, DBGM_T());
  const Frame f = sc.fa->f;  // a copy (uniform registers): stores of the kernel cannot alias it
  const MoveSet &ms = sc.fa->ms;
  __shared__ uint32_t obj_base[MAX_MOVE_OBJECTS];  // global rank of the object's next member in this chunk
  __shared__ uint16_t tracks[MAX_MOVE_OBJECTS];
  __shared__ uint32_t wave_cnt[MV_WAVES][MAX_MOVE_OBJECTS];
  __shared__ uint32_t e_shift[MAX_MOVE_OBJECTS];   // global rank of the object's first local member - its local offset
  __shared__ uint32_t block_total;
  __shared__ uint32_t ca_idx[CA_CAP], ca_ent[CA_CAP], ca_n;
  __shared__ uint8_t ca_obj[CA_CAP];
  __shared__ uint32_t vox_list[MV_CHUNK], vox_n, vox_base;  // voxels that got their first copy from this chunk
  __shared__ uint32_t my_e[MV_ITEMS][TPB];                  // global rank of the member in slot r * TPB + thread of the chunk, MV_NIL: none
  __shared__ uint8_t my_o[MV_ITEMS][TPB];                   // ... and its object
  __shared__ uint32_t rank_cnt[MV_ITEMS][MV_WAVES][MAX_MOVE_OBJECTS];  // members per (round, wave, object), then their running offsets
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (threadIdx.x == 0) vox_n = 0;
  const uint32_t n = *sc.mv_nlist;
  const size_t n_slots = (size_t)d.v_count * d.S;
  if (threadIdx.x < MAX_MOVE_OBJECTS) tracks[threadIdx.x] = (int)threadIdx.x < n_obj ? ms.track[threadIdx.x] : OWNER_NONE;
  if ((int)threadIdx.x < n_obj) {
    // e = sum_{k'<k} sum_r' C[r'][k'] + sum_{r'<rank} C[r'][k] + j.  Single shard: the scanned count matrix already
    // is that prefix (shift 0).
    uint32_t shift = 0;
    if (counts_all) {
      uint32_t run = 0;
      for (int k = 0; k < (int)threadIdx.x; ++k)
        for (int r = 0; r < world; ++r) run += (uint32_t)counts_all[r * HALO_OBJ + k];
      for (int r = 0; r < rank; ++r) run += (uint32_t)counts_all[r * HALO_OBJ + threadIdx.x];
      shift = run - offs[(size_t)threadIdx.x * n];
    }
    e_shift[threadIdx.x] = shift;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    uint32_t total = 0;
    if (counts_all) {
      for (int k = 0; k < n_obj; ++k)
        for (int r = 0; r < world; ++r) total += (uint32_t)counts_all[r * HALO_OBJ + k];
    } else {
      total = offs[(size_t)n_obj * n];
    }
    sc.cnt->n_moved = total;
    if (total > sc.cap_move || sc.cur->move_list_overflow) sc.cnt->overflow = 1;
  }
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const uint32_t na_total = st.alias[0] < ALIAS_CAP ? st.alias[0] : ALIAS_CAP;
  for (uint32_t pos = blockIdx.x; pos < n; pos += gridDim.x) {
    __syncthreads();
    if (threadIdx.x == 0) block_total = 0;
    __syncthreads();
    if ((int)threadIdx.x < n_obj) {
      uint32_t o0 = offs[(size_t)threadIdx.x * n + pos];
      uint32_t o1 = offs[(size_t)threadIdx.x * n + pos + 1];  // next position (or next object's first)
      obj_base[threadIdx.x] = o0 + e_shift[threadIdx.x];
      if (o1 != o0) atomicAdd(&block_total, o1 - o0);
    }
    __syncthreads();
    if (block_total == 0) continue;  // no member of any moving object in this chunk
    const uint32_t chunk = sc.mv_list[pos];
    const size_t base = (size_t)chunk * MV_CHUNK;
    // older memberships of moving objects inside this chunk (State::alias): slot, object rank, entry
    uint32_t n_ca = 0;
    if (na_total) {  // (block-uniform; the table is empty in nearly every frame)
      if (threadIdx.x == 0) ca_n = 0;
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
      n_ca = ca_n < CA_CAP ? ca_n : CA_CAP;
      if (threadIdx.x == 0 && ca_n > CA_CAP) sc.cnt->overflow = 1;
    }
    if (n_ca == 0) {
      // The usual case: every slot belongs to at most one moving object.  Ranks in three steps with two barriers (round 2
      // ranked round by round: 48 barriers per chunk, each owner load behind the barrier before it):
      //   1. every wave ranks its 64 slots of each of the 16 rounds by ballot and notes how many members of which object it
      //      saw - rank_cnt[round][wave][object]; nothing waits for anybody, the 16 owner loads of a thread overlap;
      //   2. one thread per object turns its 64 counts into running offsets, in (round, wave) order = ascending slot index;
      //   3. a member's global rank = the object's first rank in this chunk + the offset of its (round, wave) + its rank in the wave.
      for (uint32_t k = threadIdx.x; k < (uint32_t)(MV_ITEMS * MV_WAVES * MAX_MOVE_OBJECTS); k += TPB) (&rank_cnt[0][0][0])[k] = 0;
      __syncthreads();
      uint16_t ow[MV_ITEMS];  // the thread's sixteen owner entries, requested together
#pragma unroll
      for (int r = 0; r < MV_ITEMS; ++r) {
        const size_t li = base + (size_t)r * TPB + threadIdx.x;
        ow[r] = li < n_slots ? st.owner[li] : OWNER_NONE;
      }
#pragma unroll
      for (int r = 0; r < MV_ITEMS; ++r) {
        const uint8_t o = obj_of(ow[r], tracks, n_obj);
        const bool valid = o != 0xFF;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 6; ++b) {
          bool bit = (o >> b) & 1u;
          uint64_t m = __ballot(bit);
          peers &= bit ? m : ~m;
        }
        const uint32_t rank_in_wave = (uint32_t)__popcll(peers & lt_mask);
        if (valid && rank_in_wave == 0) rank_cnt[r][wid][o] = (uint32_t)__popcll(peers);
        my_e[r][threadIdx.x] = valid ? rank_in_wave : MV_NIL;
        my_o[r][threadIdx.x] = o;
      }
      __syncthreads();
      if ((int)threadIdx.x < n_obj) {
        uint32_t run = obj_base[threadIdx.x];
        for (int r = 0; r < MV_ITEMS; ++r)
#pragma unroll
          for (int w = 0; w < MV_WAVES; ++w) {
            const uint32_t c = rank_cnt[r][w][threadIdx.x];
            rank_cnt[r][w][threadIdx.x] = run;
            run += c;
          }
      }
      __syncthreads();
      DBGM(0, 1, DBGM_T());
    } else {
      for (int r = 0; r < MV_ITEMS; ++r) {
        if (threadIdx.x < MAX_MOVE_OBJECTS) {
#pragma unroll
          for (int w = 0; w < MV_WAVES; ++w) wave_cnt[w][threadIdx.x] = 0;
        }
        __syncthreads();
        const size_t li = base + (size_t)r * TPB + threadIdx.x;
        move_round_with_aliases(d, f, flt, ms, st, sc, li, n_slots, n_obj, tracks, obj_base, wave_cnt, ca_idx, ca_ent, ca_obj, n_ca,
                                lt_mask, wid);
        __syncthreads();
        if (threadIdx.x < MAX_MOVE_OBJECTS) {
          uint32_t add = 0;
#pragma unroll
          for (int w = 0; w < MV_WAVES; ++w) add += wave_cnt[w][threadIdx.x];
          obj_base[threadIdx.x] += add;
        }
        __syncthreads();
      }
    }
    // The moves themselves, outside the rounds: a move is a chain of dependent loads (position, noise, record) in a few
    // lanes, and inside a round every barrier waited for the slowest of them - sixteen times per chunk.  Four members at a
    // time: all their loads, then their stores.  (Ranks and objects wait in LDS: sixteen copies of the move code, which
    // register arrays would need, are 47 K instructions.)
    if (n_ca == 0) {
      const long long cursor = (long long)sc.cur->move_cursor;  // (advanced by k_move_replay, after this kernel)
#pragma unroll 1
      for (int r0 = 0; r0 < MV_ITEMS; r0 += 4) {
        MoveLoaded ml[4];
        uint32_t e4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          e4[u] = my_e[r0 + u][threadIdx.x];  // rank in its wave so far
          if (e4[u] != MV_NIL) e4[u] += rank_cnt[r0 + u][wid][my_o[r0 + u][threadIdx.x]];
          if (e4[u] != MV_NIL) ml[u] = move_load(d, flt, st, cursor, e4[u], base + (size_t)(r0 + u) * TPB + threadIdx.x, false);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (e4[u] != MV_NIL)
            move_store(d, f, ms, st, sc, ml[u], (int)my_o[r0 + u][threadIdx.x], e4[u], base + (size_t)(r0 + u) * TPB + threadIdx.x, false,
                       vox_list, &vox_n);
      }
      DBGM(0, 2, DBGM_T());
      flush_move_voxels(sc, vox_list, &vox_n, &vox_base);  // (n_ca is workgroup-uniform)
      DBGM(0, 3, DBGM_T());
    }
  }
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
    c.forget_bits = r.forget_bits;
    c.w = r.w;
    c.ts = (uint16_t)(r.ts_track & 0xffffu);
    c.track = (uint16_t)(r.ts_track >> 16);
    c.owner = (uint16_t)(r.owner_label_status & 0xffffu);
    c.label = (uint8_t)((r.owner_label_status >> 16) & 0xffu);
    c.status = (uint8_t)(r.owner_label_status >> 24);
    c.pad = 0;
    sc.mv_copy[e] = c;
    move_link(d, sc, r.voxel, e);
  }
}

// phase 2 (operations.h:351-361): re-insert the copies, first vacant slot, in (object, index) order = ascending global
// rank.  One thread per target voxel: it walks the voxel's list, picks the next S-1 ranks in ascending order, replays
// them, and repeats while the voxel still has a vacant slot (a copy whose own stamp is older than the slab's leaves
// its slot vacant, so a list can be longer than the voxel; once the voxel is full every later copy is dropped,
// operations.h:357).  The list head is left idle again.
template <int S>
__global__ __launch_bounds__(TPB) void k_move_replay(Dims d, Filter flt, State st, Scratch sc) {
  if (sc.fa->n_obj <= 0) return;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // the table cursor of RingBufferOperations::gaussian_random_calculator_ advanced by three per moved particle
    long long c = (long long)sc.cur->move_cursor + 3ll * (long long)sc.cnt->n_moved;
    sc.cur->move_cursor = (int32_t)(c % flt.noise_n);
  }
  const uint32_t n_vox = sc.cnt->n_move_voxels;
  const uint32_t epoch = sc.fa->f.epoch;
  const uint32_t stride = gridDim.x * blockDim.x;
  __shared__ uint32_t n_ok_block;  // (statistic: one global atomic per workgroup, not per voxel)
  if (threadIdx.x == 0) n_ok_block = 0;
  __syncthreads();
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_vox; t += stride) {
    DBGM(1, 0, DBGM_T());
    const uint32_t lv = sc.mv_vlist[t];
    const uint32_t head = sc.mv_head[lv];
    sc.mv_head[lv] = MV_NIL;
    if (sc.cnt->overflow) continue;
    const uint32_t v = d.v_begin + lv;
    uint32_t rx, ry, rz;
    voxel_to_ring(d, v, rx, ry, rz);
    uint32_t a = st.stamps_x[rx], b = st.stamps_y[ry], c = st.stamps_z[rz];
    uint32_t smax = a > b ? a : b;
    smax = smax > c ? smax : c;
    const size_t base = (size_t)lv * S;
    uint8_t stv[S];
    uint16_t tsv[S];
    __builtin_memcpy(stv, st.status + base * REC_STATUS, S);
    __builtin_memcpy(tsv, st.ts + base * REC_TS, 2 * S);
    // the voxel's owner entries and the length of the table of older memberships, with the rows above (owner_insert_local)
    uint16_t own[S];
    __builtin_memcpy(own, st.owner + base, 2 * S);
    const uint32_t n_alias = st.alias[0];
    bool alias_touched = false;
    uint32_t n_ok = 0;
    bool more = true, full = false;
    long long last = -1;  // largest rank replayed so far
    while (more && !full) {
      uint32_t best[S - 1];  // the S-1 smallest ranks above `last`, ascending
#pragma unroll
      for (int i = 0; i < S - 1; ++i) best[i] = MV_NIL;
      uint32_t n_above = 0;  // ranks above `last` on the list
      for (uint32_t cur = head; cur != MV_NIL; cur = sc.mv_next[cur]) {
        if ((long long)cur <= last) continue;
        ++n_above;
        uint32_t x = cur;
#pragma unroll
        for (int i = 0; i < S - 1; ++i)
          if (x < best[i]) {
            const uint32_t y = best[i];
            best[i] = x;
            x = y;
          }
      }
      DBGM(1, 1, DBGM_T() + (stv[1] == 0xEE ? 1 : 0) + (own[1] == 0xEEEE ? 1 : 0));
      more = n_above > (uint32_t)(S - 1);  // ranks beyond this batch (a walk of the list is a chain of dependent loads: no second one to find nothing)
      MoveCopy cc[S - 1];  // the batch's copies, requested together
#pragma unroll
      for (int u = 0; u < S - 1; ++u)
        if (best[u] != MV_NIL) cc[u] = sc.mv_copy[best[u]];
#pragma unroll
      for (int u = 0; u < S - 1; ++u) {
        const uint32_t e = best[u];
        if (e == MV_NIL || full) break;
        last = e;
        int slot = -1;
#pragma unroll
        for (int i = S - 1; i >= 1; --i)
          if (stv[i] == ST_INVALID || (uint32_t)tsv[i] < smax) slot = i;
        if (slot < 0) {  // voxel full: this copy and all later ones are dropped (operations.h:357)
          full = true;
          break;
        }
        const MoveCopy c = cc[u];
        const uint8_t cs = c.status;
        const uint16_t cts = c.ts;
        st.pos4[base + slot] = make_float4(c.x, c.y, c.z, __uint_as_float(c.forget_bits));
        st.w[base * REC_W + slot] = c.w;
        st.ts[base * REC_TS + slot] = cts;
        st.track[base * REC_TRACK + slot] = c.track;
        st.label[base * REC_LABEL + slot] = c.label;
        st.status[base * REC_STATUS + slot] = cs;
        {  // the new index joins the object's set.  (ONE copy of the insertion with the slot as a run-time value: inside
           // `if (i == slot)` of an unrolled loop a wave whose lanes fill different slots runs the body once per slot)
          uint16_t o = OWNER_NONE;
#pragma unroll
          for (int i = 1; i < S; ++i) o = i == slot ? own[i] : o;
          if (!owner_insert_local(st, base + slot, c.owner, o, n_alias, alias_touched)) sc.cnt->overflow = 1;
#pragma unroll
          for (int i = 1; i < S; ++i) own[i] = i == slot ? o : own[i];
        }
        st.owner_flag[(base + slot) / OWNER_CHUNK] = 1;
#pragma unroll
        for (int i = 1; i < S; ++i)
          if (i == slot) {
            stv[i] = cs;
            tsv[i] = cts;
          }
        ++n_ok;
      }
    }
    DBGM(1, 2, DBGM_T());
    DBGM(1, 3, n_ok);
    if (n_ok) {
      st.vflag[lv] = VF_DIRTY;
      mark_tile(st, lv, epoch);
      atomicAdd(&n_ok_block, n_ok);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && n_ok_block) atomicAdd(&sc.cnt->n_move_reinserted, n_ok_block);
}

// removeObjectByTrackID (object_layer.h:414-425): every index of the set -> INVALID, set erased.
__global__ __launch_bounds__(TPB) void k_remove(State st, size_t n_slots, const FrameArgs *__restrict__ fa, int p_n) {
  const int n = fa->n_remove;
  if (n <= 0) return;  // nothing to wipe in this frame
  const uint32_t epoch = fa->f.epoch;
  const uint16_t *__restrict__ tracks = fa->remove;
  if (blockIdx.x == 0) {  // older memberships of the removed objects (State::alias)
    const uint32_t na = st.alias[0] < ALIAS_CAP ? st.alias[0] : ALIAS_CAP;
    for (uint32_t k = threadIdx.x; k < na; k += blockDim.x) {
      const uint32_t trk = st.alias[3 + 2 * k];
      if (trk == OWNER_NONE) continue;
      for (int q = 0; q < n; ++q)
        if (tracks[q] == trk) {
          st.status[rec_index(st.alias[2 + 2 * k], p_n, REC_STATUS)] = ST_INVALID;
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
      if (o == OWNER_NONE) continue;
      for (int k = 0; k < n; ++k)
        if (tracks[k] == o) {
          st.status[rec_index(i, p_n, REC_STATUS)] = ST_INVALID;
          st.vflag[i >> p_n] = VF_DIRTY;
          mark_tile(st, i >> p_n, epoch);
          st.owner[i] = OWNER_NONE;
          break;
        }
    }
  }
}

// after sdm_load_state: recompute the chunk flags from the owner array
__global__ __launch_bounds__(TPB) void k_owner_flags(const uint16_t *__restrict__ owner, size_t n_slots,
                                                     uint8_t *__restrict__ owner_flag) {
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
  if (threadIdx.x == 0) owner_flag[blockIdx.x] = any_owner ? 1 : 0;
}

inline unsigned blocks_for(size_t n, int tpb = TPB) { return (unsigned)((n + tpb - 1) / tpb); }

}  // namespace

void launch_owner_flags(const Dims &d, const State &st, hipStream_t s) {
  const size_t n_slots = (size_t)d.v_count * d.S;
  hipLaunchKernelGGL(k_owner_flags, dim3((unsigned)((n_slots + OWNER_CHUNK - 1) / OWNER_CHUNK)), dim3(TPB), 0, s, st.owner, n_slots,
                     st.owner_flag);
}

size_t move_blocks(const Dims &d) { return ((size_t)d.v_count * d.S + MV_CHUNK - 1) / MV_CHUNK; }
size_t move_count_elems() { return (size_t)MAX_MOVE_OBJECTS * MV_LIST_CAP + 1; }

// The launch sequence below is the same every frame (hipGraph): kernels of a frame without moving objects / removals
// return at once, the scan covers n_obj x (chunks on the list) + 1 elements, a number read on the device.
// step 1: collect every moving object's members (ascending index) and publish the per-object counts
// by_value != nullptr: the chain runs ahead of the frame's k_frame_begin on a stream of its own; its first kernel takes
// the frame block by value and stores it in fa_moves for the others
void launch_moves_count(const Dims &d, const State &st, const Scratch &sc, int32_t *counts_local, hipStream_t s, const FrameArgs *by_value) {
  const size_t n_slots = (size_t)d.v_count * d.S;
  const FrameArgs *fa = by_value ? sc.fa_moves : sc.fa_side;
  if (by_value)
    hipLaunchKernelGGL(k_move_chunks_v, dim3(1), dim3(1024), 0, s, st.owner_flag, (uint32_t)move_blocks(d), sc.mv_list, sc.mv_nlist, sc.cur,
                       sc.mv_cnt, *by_value, const_cast<FrameArgs *>(sc.fa_moves), st.alias, st.owner_flag);
  else
    hipLaunchKernelGGL(k_move_chunks, dim3(1), dim3(1024), 0, s, st.owner_flag, (uint32_t)move_blocks(d), sc.mv_list, sc.mv_nlist, sc.cur,
                       sc.mv_cnt, fa, st.alias, st.owner_flag);
  hipLaunchKernelGGL(k_move_count, dim3(1024), dim3(TPB), 0, s, st.owner, n_slots, fa, sc.mv_cnt, st.owner_flag, sc.mv_list, sc.mv_nlist,
                     st.alias);
  exclusive_scan_u32(sc.mv_cnt, sc.mv_cnt, move_count_elems(), sc.scan_scratch_m, s, sc.mv_nlist + 2);
  // the per-object counts are only needed as a separate row when they are exchanged between shards
  if (d.v_count != d.V) hipLaunchKernelGGL(k_move_local_counts, dim3(1), dim3(HALO_OBJ), 0, s, sc.mv_cnt, counts_local, sc, fa);
}

// step 2 (after the counts of all shards are known): global ranks, transform, export of slab-crossing copies
void launch_moves_transform(const Dims &d, const Filter &flt, const State &st, const Scratch &sc, const int32_t *counts_all, int world,
                            int rank, hipStream_t s) {
  hipLaunchKernelGGL(k_move_apply, dim3(1024), dim3(TPB), 0, s, d, flt, st, sc, sc.mv_cnt, d.v_count != d.V ? counts_all : nullptr, world,
                     rank);
}

// step 3 (after the export buffers of all shards are gathered): import, ordered replay per target voxel
void launch_moves_finish(const Dims &d, const Filter &flt, const State &st, const Scratch &sc, int world, int rank, hipStream_t s) {
  if (world > 1 && sc.halo_recv) hipLaunchKernelGGL(k_move_import, dim3(64, world), dim3(TPB), 0, s, d, sc, world, rank);
  dim3 grid(256);
  switch (d.p_n) {
    case 1: hipLaunchKernelGGL(k_move_replay<2>, grid, dim3(TPB), 0, s, d, flt, st, sc); break;
    case 2: hipLaunchKernelGGL(k_move_replay<4>, grid, dim3(TPB), 0, s, d, flt, st, sc); break;
    case 3: hipLaunchKernelGGL(k_move_replay<8>, grid, dim3(TPB), 0, s, d, flt, st, sc); break;
    default: hipLaunchKernelGGL(k_move_replay<16>, grid, dim3(TPB), 0, s, d, flt, st, sc); break;
  }
}

void launch_remove(const Dims &d, const State &st, const Scratch &sc, hipStream_t s) {
  const size_t n_slots = (size_t)d.v_count * d.S;
  const size_t n_chunks = (n_slots + OWNER_CHUNK - 1) / OWNER_CHUNK;
  hipLaunchKernelGGL(k_remove, dim3((unsigned)(n_chunks < 1024 ? n_chunks : 1024)), dim3(TPB), 0, s, st, n_slots, sc.fa, d.p_n);
}

}  // namespace sdm
