// kernels.hip — the per-frame particle/voxel kernels of libsdm_hip (gfx950).
//
// Each kernel restates one CPU loop of the reference (file:line cited per kernel) as a
// data-parallel pass over the layout of sdm_internal.h (dense per-voxel arrays for what whole-map sweeps stream,
// one record per voxel for what only live voxels need).  Sequential-order semantics of the reference
// (first vacant slot, 9-pass stride-3 birth raster, one resample per voxel per frame) are
// preserved by grouping work per voxel and replaying each voxel's ordered list in one thread:
// all state such a list touches is voxel-local.
//
// Float arithmetic that decides an integer (voxel index, pixel, LUT index, resample survivor)
// is written in the reference's operation order with contraction off.
#include <algorithm>

#include <type_traits>

#include <hip/hip_ext.h>
#include "sdm_internal.h"
#include "sdm_scratch.h"

#pragma clang fp contract(off)

namespace sdm {

namespace {

constexpr int TPB = 256;

#ifdef SDM_AB_TIMERS
// development aid (tools/ab_build.sh -DSDM_AB_TIMERS=1): where does a kernel's time go?  100 MHz wall-clock ticks at a few
// checkpoints per wave, reduced with atomics into g_dbg (read with hipMemcpyFromSymbol by tools/probes/timers.py via
// sdm_debug_timers)
}  // namespace
__device__ unsigned long long g_dbg[6][8192 * 4];  // [kernel][workgroup][checkpoint], plain stores by one lane
namespace {
#define DBG_T() wall_clock64()
#define DBG_PUTK(k, i, v) do { if (blockIdx.x < 8192) g_dbg[k][blockIdx.x * 4 + (i)] = (unsigned long long)(v); } while (0)
#define DBG_PUT(i, v) DBG_PUTK(0, i, v)
#define DBG_LANE0(k, i) do { if (threadIdx.x == 0 && threadIdx.y == 0) DBG_PUTK(k, i, DBG_T()); } while (0)
#else
#define DBG_T() 0ull
#define DBG_PUT(i, v)
#define DBG_LANE0(k, i)
#endif

// Whole-voxel fetch: all S slots of one field with the widest aligned vector loads (16 B pieces).  The copy goes
// through a plain vector type so that the compiler cannot narrow it to the bytes it happens to use (slot 0 of
// most arrays is dead, which otherwise splits a 32-byte row into dword/dwordx3 pieces).
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
// A pointer that was itself loaded from memory (the frame's input images: FrameArgs::depth / cloud) has no address space
// the compiler can see: its loads become FLAT instructions, which count on both wait counters and complete out of
// order with everything else - with one of them in flight the compiler waits for ALL outstanding memory operations
// (vmcnt(0) lgkmcnt(0)) before every later use of any loaded value.  These buffers are device memory: say so.
template <typename T>
using gptr = const __attribute__((address_space(1))) T *;
template <typename T>
__device__ __forceinline__ gptr<T> as_global(const T *p) {
  return (gptr<T>)p;
}
// a whole labelled point (20 bytes) from such a buffer
__device__ __forceinline__ sdm_labeled_point load_point(const sdm_labeled_point *cloud, size_t i) {
  static_assert(sizeof(sdm_labeled_point) == 20, "five words");
  gptr<uint32_t> w = (gptr<uint32_t>)(cloud + i);
  uint32_t t[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) t[k] = w[k];
  sdm_labeled_point o;
  __builtin_memcpy(&o, t, sizeof(o));
  return o;
}
// A = alignment the caller guarantees for src; below the natural
// alignment of the widest piece (S <= 4 only) the copy is left to the compiler.
template <int A = 16, typename T, int N>
__device__ __forceinline__ void load_vec(T (&dst)[N], const T *src) {
  constexpr int B = (int)sizeof(T) * N;
  if constexpr (A < (B > 16 ? 16 : B)) {
    __builtin_memcpy(dst, __builtin_assume_aligned(src, A), B);
  } else if constexpr (B >= 16) {
    const v4u *p = reinterpret_cast<const v4u *>(src);
    v4u tmp[B / 16];
#pragma unroll
    for (int i = 0; i < B / 16; ++i) tmp[i] = __builtin_nontemporal_load(p + i);
    __builtin_memcpy(dst, tmp, B);
  } else if constexpr (B == 8) {
    v2u tmp = __builtin_nontemporal_load(reinterpret_cast<const v2u *>(src));
    __builtin_memcpy(dst, &tmp, 8);
  } else if constexpr (B == 4) {
    uint32_t tmp = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(src));
    __builtin_memcpy(dst, &tmp, 4);
  } else {
    uint16_t tmp = __builtin_nontemporal_load(reinterpret_cast<const uint16_t *>(src));
    __builtin_memcpy(dst, &tmp, 2);
  }
}
template <int A = 16, typename T, int N>
__device__ __forceinline__ void store_vec(T *dst, const T (&src)[N]) {
  constexpr int B = (int)sizeof(T) * N;
  constexpr int AL = A < (B > 16 ? 16 : B) ? A : (B > 16 ? 16 : B);
  __builtin_memcpy(__builtin_assume_aligned(dst, AL), src, B);
}

// ---- whole-record access (layout: sdm_internal.h, SlotRef).  A record - the S - 1 particle slots of one voxel, 10 (S - 1)
// bytes at a 2-byte aligned address - is fetched as raw words: 16-byte pieces and one 8-byte piece where the length asks
// for it, each an under-aligned access (global_load_dwordx4 / ds_read_b128 take any address on gfx950), up to 6 bytes
// beyond the record's end (the next record or the array's padding), all requested before anything is looked at; the
// fields are then taken out of the registers with shifts at compile-time offsets.  The arrays the callers work on keep
// one entry per SLOT (index 0 = the time particle: constants nobody reads) so that slot numbers stay what they are in
// the reference.
template <int S>
struct RecRaw {
  static constexpr int L = S - 1, BYTES = 10 * L, N8 = (BYTES + 7) / 8;  // particle slots, bytes, 8-byte units fetched
  uint32_t d[2 * N8];
};
template <int S, bool NT = true>
__device__ __forceinline__ void rec_fetch(RecRaw<S> &raw, const unsigned char *r) {
  constexpr int N8 = RecRaw<S>::N8;
#pragma unroll
  for (int j = 0; j < N8 / 2; ++j) {
    const rec_v4u *q = reinterpret_cast<const rec_v4u *>(r + 16 * j);
    const rec_v4u t = NT ? __builtin_nontemporal_load(q) : *q;
    raw.d[4 * j + 0] = t.x;
    raw.d[4 * j + 1] = t.y;
    raw.d[4 * j + 2] = t.z;
    raw.d[4 * j + 3] = t.w;
  }
  if constexpr (N8 & 1) {
    const rec_v2u *q = reinterpret_cast<const rec_v2u *>(r + 16 * (N8 / 2));
    const rec_v2u t = NT ? __builtin_nontemporal_load(q) : *q;
    raw.d[2 * N8 - 2] = t.x;
    raw.d[2 * N8 - 1] = t.y;
  }
}
__device__ __forceinline__ uint32_t raw_u16(const uint32_t *d, int off) { return (d[off >> 2] >> ((off & 3) * 8)) & 0xffffu; }  // off even
__device__ __forceinline__ uint32_t raw_u8(const uint32_t *d, int off) { return (d[off >> 2] >> ((off & 3) * 8)) & 0xffu; }
template <int S>
__device__ __forceinline__ void rec_unpack(const RecRaw<S> &raw, float (&wv)[S], uint16_t (&ts)[S], uint16_t (&trk)[S], uint8_t (&lab)[S],
                                           uint8_t (&stv)[S]) {
  constexpr int L = S - 1;
  wv[0] = 0.f;
  ts[0] = 0;
  trk[0] = 0;
  lab[0] = 0;
  stv[0] = ST_TIMEPTC;
#pragma unroll
  for (int k = 0; k < L; ++k) {
    wv[k + 1] = __uint_as_float(raw.d[k]);
    ts[k + 1] = (uint16_t)raw_u16(raw.d, 4 * L + 2 * k);
    trk[k + 1] = (uint16_t)raw_u16(raw.d, 6 * L + 2 * k);
    lab[k + 1] = (uint8_t)raw_u8(raw.d, 8 * L + k);
    stv[k + 1] = (uint8_t)raw_u8(raw.d, 9 * L + k);
  }
}
// the whole record of one voxel into per-slot arrays
template <int S>
__device__ __forceinline__ void rec_load(const unsigned char *r, float (&wv)[S], uint16_t (&ts)[S], uint16_t (&trk)[S], uint8_t (&lab)[S],
                                         uint8_t (&stv)[S]) {
  RecRaw<S> raw;
  rec_fetch<S>(raw, r);
  rec_unpack<S>(raw, wv, ts, trk, lab, stv);
}
// status and time-stamp rows only (visibility, the replays): the status bytes are the record's last L bytes - one 8- or
// 16-byte piece that ends with the record -, the stamps 2 L bytes from offset 4 L in 8-byte units
template <int S>
__device__ __forceinline__ void rec_load_st_ts(const unsigned char *r, uint8_t (&stv)[S], uint16_t (&ts)[S]) {
  constexpr int L = S - 1;
  stv[0] = ST_TIMEPTC;
  ts[0] = 0;
  if constexpr (L <= 8) {
    const rec_v2u t = __builtin_nontemporal_load(reinterpret_cast<const rec_v2u *>(r + 10 * L - 8));
    const uint32_t d[2] = {t.x, t.y};
#pragma unroll
    for (int k = 0; k < L; ++k) stv[k + 1] = (uint8_t)raw_u8(d, 8 - L + k);
  } else {
    const rec_v4u t = __builtin_nontemporal_load(reinterpret_cast<const rec_v4u *>(r + 10 * L - 16));
    const uint32_t d[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int k = 0; k < L; ++k) stv[k + 1] = (uint8_t)raw_u8(d, 16 - L + k);
  }
  constexpr int N8 = (2 * L + 7) / 8;
  uint32_t d[2 * N8];
#pragma unroll
  for (int j = 0; j < N8 / 2; ++j) {
    const rec_v4u t = __builtin_nontemporal_load(reinterpret_cast<const rec_v4u *>(r + 4 * L + 16 * j));
    d[4 * j + 0] = t.x;
    d[4 * j + 1] = t.y;
    d[4 * j + 2] = t.z;
    d[4 * j + 3] = t.w;
  }
  if constexpr (N8 & 1) {
    const rec_v2u t = __builtin_nontemporal_load(reinterpret_cast<const rec_v2u *>(r + 4 * L + 16 * (N8 / 2)));
    d[2 * N8 - 2] = t.x;
    d[2 * N8 - 1] = t.y;
  }
#pragma unroll
  for (int k = 0; k < L; ++k) ts[k + 1] = (uint16_t)raw_u16(d, 2 * k);
}
// whole rows back (rare paths: clamp / cull write-backs of the sweep, stale slots deleted by the visibility pass)
template <int S>
__device__ __forceinline__ void rec_store_w(unsigned char *r, const float (&wv)[S]) {
  __builtin_memcpy(__builtin_assume_aligned(r, 2), &wv[1], 4 * (S - 1));
}
template <int S>
__device__ __forceinline__ void rec_store_status(unsigned char *r, const uint8_t (&stv)[S]) {
  __builtin_memcpy(r + 9 * (S - 1), &stv[1], S - 1);
}
// the whole record from per-slot arrays: packed into words, stored as 16-byte pieces and what is left - exactly 10 (S - 1)
// bytes, nothing of the next record is touched
template <int S>
__device__ __forceinline__ void rec_store_all(unsigned char *r, const float (&wv)[S], const uint16_t (&ts)[S], const uint16_t (&trk)[S],
                                              const uint8_t (&lab)[S], const uint8_t (&stv)[S]) {
  constexpr int L = S - 1, NW = (10 * L + 3) / 4;
  uint32_t d[NW];
#pragma unroll
  for (int i = 0; i < NW; ++i) d[i] = 0;
#pragma unroll
  for (int k = 0; k < L; ++k) {
    d[k] = __float_as_uint(wv[k + 1]);
    d[(4 * L + 2 * k) >> 2] |= (uint32_t)ts[k + 1] << (((4 * L + 2 * k) & 3) * 8);
    d[(6 * L + 2 * k) >> 2] |= (uint32_t)trk[k + 1] << (((6 * L + 2 * k) & 3) * 8);
    d[(8 * L + k) >> 2] |= (uint32_t)lab[k + 1] << (((8 * L + k) & 3) * 8);
    d[(9 * L + k) >> 2] |= (uint32_t)stv[k + 1] << (((9 * L + k) & 3) * 8);
  }
  __builtin_memcpy(__builtin_assume_aligned(r, 2), d, 10 * L);
}

__device__ __forceinline__ uint32_t stamp_max(const State &st, uint32_t rx, uint32_t ry, uint32_t rz) {
  uint32_t a = st.stamps_x[rx], b = st.stamps_y[ry], c = st.stamps_z[rz];
  uint32_t m = a > b ? a : b;
  return m > c ? m : c;
}

// ------------------------------------------------------------------------------------ A13
// RingBufferOperations::clear (mc_ring/operations.h:684-723).  A fresh map is all zeros (launch_clear: hipMemsetAsync;
// INVALID = 0, and the time particles' TIMEPTC status is not stored).
// RingBufferOperations::clear on a used map (operations.h:697-722): status, position, weight, time stamp; track id,
// label and forget count stay.  One pass, one thread per slot: everything sdm_clear resets is written here (the
// per-voxel arrays by the slot-0 lane), so the map's bytes cross HBM once.
__global__ __launch_bounds__(TPB) void k_clear_slots(Dims d, State st, size_t n) {
  size_t li = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (li >= n) return;
  st.pos4[li] = make_float4(0.f, 0.f, 0.f, 0.f);  // (the forget count, which clear() does not touch, lives in State::forget)
  uint32_t slot = (uint32_t)li & (uint32_t)(d.S - 1);
  if (slot) {
    const SlotRef r = slot_ref_li(st, d.p_n, li);
    r.set_w(0.f);
    r.set_ts(0);
    r.set_status(ST_INVALID);
  }
  st.owner[li] = OWNER_NONE;
  if (slot == 0) {
    size_t lv = li >> d.p_n;
    st.vts[lv] = 0;
    st.vflag[lv] = 0;
    static_assert(sizeof(sdm_voxel_result) == 8, "the result entry is cleared as one 8-byte store");
    reinterpret_cast<uint2 *>(st.res)[lv] = make_uint2(0u, 0u);
  }
}

// The same reset with every byte it touches on a lane-linear 16-byte access (S >= 8).  k_clear_slots stores a weight, a stamp and a status byte per lane: three store instructions that each
// leave holes in five or six lines of the record array.  Here a workgroup takes CLR_VOX consecutive voxels: their
// positions (stored whole; the forget counts have their own plane and are not touched), the pieces of their records that hold weights, stamps and
// status bytes (the track ids and labels in between are left alone: nothing of a record is read), their owners and the
// per-voxel arrays, 1 KB contiguous per store instruction, all loads of a thread requested before its first store.
// Measured on the C3 map (round 4, tools/probes/clear_time.py with four builds on one box): 1.73 ms per sdm_clear with k_clear_slots, 1.42 ms with
// this kernel (the kernel itself 1.45 -> 1.33 ms; FETCH_SIZE = the positions and nothing else), 1.63 ms when the kept
// pieces are read and written back so that every sector of the record array is written whole - the memory side merges
// the partial sectors better than it serves the extra reads - and 1.62-1.74 ms with plain instead of non-temporal accesses.
constexpr int CLR_VOX = 256;
template <int S>
__global__ __launch_bounds__(TPB) void k_clear_map(Dims d, State st) {
  static_assert(S >= 8 && (CLR_VOX * 10 * (S - 1)) % 16 == 0, "a workgroup's records are whole 16-byte pieces");
  constexpr int L = S - 1, REC = 10 * L;  // particle slots, bytes of one record
  static_assert(ST_INVALID == 0, "every byte clear() resets in a record is a zero byte");
  const uint32_t tid = threadIdx.x;
  const size_t lv0 = (size_t)blockIdx.x * CLR_VOX;
  const uint32_t nv = (uint32_t)(d.v_count - lv0 < (size_t)CLR_VOX ? d.v_count - lv0 : (size_t)CLR_VOX);  // a multiple of 8
  // positions: S pieces per voxel
  constexpr int PP = CLR_VOX * S / TPB;
  v4u *pp = reinterpret_cast<v4u *>(st.pos4 + lv0 * S);
  const uint32_t npos = nv * S;
  // (the forget count, which clear() leaves alone, has its own byte plane - State::forget - since round 5: the positions are
  // stored, not read.  Rounds 1-4 kept it in the positions' fourth word and this kernel read 2.1 GB to write it back;
  // storing x, y, z around it as 8 + 4 bytes was measured too: 1.59 ms per call against 1.50 ms - twelve bytes of every
  // sixteen are a partial write of every sector.)
#pragma unroll
  for (int k = 0; k < PP; ++k) {
    const uint32_t q = k * TPB + tid;
    if (q < npos) __builtin_nontemporal_store(v4u{0u, 0u, 0u, 0u}, pp + q);
  }
  // records: the workgroup's nv records are one block of nv * REC bytes that starts on a 16-byte boundary; piece q of it
  // begins at byte (16 q) mod REC of some record.  In the record [w: 4L | ts: 2L | track: 2L | label: L | status: L] the
  // bytes below 6 L and from 9 L on are reset (to zero), the track ids and labels in between are left alone: nothing of a
  // record is read.  Whole pieces go out as one 16-byte store, the others word by word resp. byte by byte.
  const uint32_t nrec = (nv * REC + 15u) / 16u;  // (nv is a multiple of 8: 80 L bytes, whole pieces)
  constexpr int RPT = (CLR_VOX * REC / 16 + TPB - 1) / TPB;
  unsigned char *rb = st.rec + lv0 * REC;
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    const uint32_t q = k * TPB + tid;
    if (q >= nrec) continue;
    const uint32_t o = (q * 16u) % (uint32_t)REC;
    uint32_t keep = 0;  // bit c: byte c of the piece holds a track id or a label
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      uint32_t off = o + c;
      off = off >= (uint32_t)REC ? off - REC : off;
      keep |= (off >= 6u * L && off < 9u * L ? 1u : 0u) << c;
    }
    if (keep == 0u) {
      __builtin_nontemporal_store(v4u{0u, 0u, 0u, 0u}, reinterpret_cast<v4u *>(rb) + q);
    } else {
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        const uint32_t kb = (keep >> (4 * c4)) & 15u;
        if (kb == 0u) {
          reinterpret_cast<uint32_t *>(rb + (size_t)q * 16)[c4] = 0u;
        } else if (kb != 15u) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (!((kb >> c) & 1u)) rb[(size_t)q * 16 + c4 * 4 + c] = 0;
        }
      }
    }
  }
  // owners (2 B per slot) and the per-voxel arrays
  {
    static_assert(OWNER_NONE == 0xFFFF, "owners are cleared as all-ones pieces");
    static_assert(sizeof(sdm_voxel_result) == 8, "the result entries are cleared as 16-byte pieces");
    const v4u zero{0u, 0u, 0u, 0u}, ones{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    constexpr int OP = (CLR_VOX * S * 2 / 16 + TPB - 1) / TPB;
    v4u *op = reinterpret_cast<v4u *>(st.owner + lv0 * S);
    const uint32_t nown = nv * S * 2 / 16;
#pragma unroll
    for (int k = 0; k < OP; ++k) {
      const uint32_t q = k * TPB + tid;
      if (q < nown) __builtin_nontemporal_store(ones, op + q);
    }
    if (tid < nv * 8 / 16) __builtin_nontemporal_store(zero, reinterpret_cast<v4u *>(st.res + lv0) + tid);
    if (tid < nv * 2 / 16) __builtin_nontemporal_store(zero, reinterpret_cast<v4u *>(st.vts + lv0) + tid);
    if (tid < nv / 8) reinterpret_cast<v2u *>(st.vflag + lv0)[tid] = v2u{0u, 0u};  // (nv is a multiple of 8, not of 16)
  }
}

// first kernel of a frame: the frame's scalars (pose, ring state, stamp updates, object motions, removals, input pointers)
// arrive by value and are stored where the frame's other kernels read them (FrameArgs, sdm_scratch.h)
__global__ __launch_bounds__(TPB) void k_set_frame(FrameArgs *__restrict__ dst, const FrameArgs src) {
  const uint32_t *s4 = reinterpret_cast<const uint32_t *>(&src);
  uint32_t *d4 = reinterpret_cast<uint32_t *>(dst);
  for (uint32_t i = threadIdx.x; i < sizeof(FrameArgs) / 4; i += blockDim.x) d4[i] = s4[i];
}

// ------------------------------------------------------------------------------------ A10
// getOccupancyResult -> determineIfVoxelOccupied -> calculateWeightAndSemanticsInVoxel
// (semantic_dsp_map.h:1239-1257, mc_ring/operations.h:623-639, 390-448).
//
// Two kernels share the per-voxel evaluation below:
//   k_occupancy      the in-frame sweep: incremental (only tiles / voxels written or stamped since the last sweep),
//                    latency-bound, kept lean so that the thousands of workgroups that leave after one byte are
//                    dispatched quickly;
//   k_occupancy_scan + k_occupancy_dense  the non-incremental sweep (first sweep of a state: sdm_load_state,
//                    sdm_set_params, sdm_clear, a wholesale stamp upload): every voxel gets its result, HBM-bound,
//                    records of dense chunks fetched cooperatively.  On a map whose particles sit on surfaces:
//                    k_occupancy_scan_lists + k_occupancy_listed + k_occupancy_dense (the host picks: map.hip, sweep_lists).
//                    On a map whose every group of 512 voxels was dense last time: k_occupancy_dense alone
//                    (launch_occupancy, OCC_SKIP_SCAN).

// A voxel that holds something: weight sum, clamp / cull write-backs and the track vote
// (calculateWeightAndSemanticsInVoxel, operations.h:390-448).  Written without branches - every decision is a select on
// values all lanes compute - so that a wave whose lanes hold different slot patterns runs one instruction stream.
// Skipped terms are added as +0.f: x + 0.f == x bit for bit for every x a sum that starts at +0.f can hold.
// The voxel's result comes back in `out` (the caller stores it: to the result array, or to an LDS stage first); flag
// byte and write-backs are stored here.
// PLAIN = the caller has checked (occupancy_is_plain) that no live slot of the voxel can be clamped, culled or is a
// guessed birth: those rules and their write-backs drop out.  The sweeps test that per wave - one special voxel sends
// the whole wave through the general version - because the rules fire rarely and cost a third of the instructions.
// SINGLE (with PLAIN) = the caller has checked (occupancy_is_plain) that all live slots of the voxel carry ONE track id -
// nearly every voxel of a real map: a voxel holds particles of one surface.  The vote then needs no pair comparisons:
// every voting slot's total is the sum of all voting weights in slot order (the very additions the pair loop performs for
// it, the +0.f terms of the other slots included), the first voting slot wins, nobody beats it (equal total, equal
// track), and the label is the last voting slot's.
// (occupancy_evaluate_core hands the voxel's new flag byte back instead of storing it)
template <int S, bool PLAIN, bool SINGLE = false>
__device__ __forceinline__ void occupancy_evaluate_core(const State &st, float occ_threshold, uint32_t remark, uint32_t lv, uint32_t smax,
                                                        const uint16_t (&ts1)[S], const uint8_t (&st1)[S], const float (&wv_in)[S],
                                                        const uint16_t (&trk16)[S], const uint8_t (&lab8)[S],
                                                        sdm_voxel_result &out, uint8_t &nflag) {
  float wv[S];
  uint32_t trk[S], lab[S], stv[S];
  bool vote[S];
  bool any = false, any_live = false, dirty_w = false, dirty_s = false;
  float weight_sum = 0.f, guessed = 0.f;
#pragma unroll
  for (int i = 1; i < S; ++i) {
    trk[i] = trk16[i];
    lab[i] = lab8[i];
    stv[i] = st1[i];
    const bool present = stv[i] != ST_INVALID;
    const bool live = present && (uint32_t)ts1[i] >= smax;  // !isParticleVacant, operations.h:810-816
    any = any || present;
    any_live = any_live || live;
    const float w = wv_in[i];
    weight_sum += live ? w : 0.f;            // the sum takes the weight as stored ...
    if constexpr (PLAIN) {
      wv[i] = w;
      vote[i] = live;
    } else {
      const bool clamp = live && w > 1.f;    // ... the clamp is written back (operations.h:404-407)
      wv[i] = clamp ? 1.f : w;
      dirty_w = dirty_w || clamp;
      const bool guess = live && stv[i] == ST_GUESSED_BORN;
      guessed += guess ? wv[i] : 0.f;
      const bool cull = live && !guess && stv[i] == ST_UPDATED && wv[i] < SDM_OCC_INIT_WEIGHT;
      stv[i] = cull ? (uint32_t)ST_INVALID : stv[i];
      dirty_s = dirty_s || cull;
      vote[i] = live && !cull;
    }
  }
  // std::map<track, weight> accumulated in slot order; winner = max weight, ties -> smallest track,
  // only weights > 0 (operations.h:429-447).  Slots that do not vote get a track id no slot can hold, so the pair loop
  // needs no second condition; the winner's label (map assignment: the last contributor's label stays) is looked up
  // afterwards.
  uint32_t tv[S];
  float wvote[S];
#pragma unroll
  for (int i = 1; i < S; ++i) {
    tv[i] = vote[i] ? trk[i] : 0xffffffffu - (uint32_t)i;  // distinct per slot: matches no other slot's id
    wvote[i] = vote[i] ? wv[i] : 0.f;
  }
  uint32_t best_t = 0xffffff00u;  // matches no slot, voting or not
  uint32_t best_l = 0;
  bool have = false;
  if constexpr (SINGLE) {
    static_assert(PLAIN, "the single-track vote is only instantiated for plain voxels");
    float tot = 0.f;
    uint32_t t1 = 0, l1 = 0;
    bool any_vote = false;
#pragma unroll
    for (int j = 1; j < S; ++j) {
      tot += wvote[j];
      t1 = vote[j] ? trk[j] : t1;
      l1 = vote[j] ? lab[j] : l1;
      any_vote = any_vote || vote[j];
    }
    have = any_vote && tot > 0.f;
    best_t = t1;
    best_l = have ? l1 : 0u;
  } else if constexpr (PLAIN) {
    // Every voting weight lies in [SDM_OCC_INIT_WEIGHT, 1] here (occupancy_is_plain tests the bit patterns: no NaN, no
    // negative, nothing to clamp), which lets the vote share its comparisons.  e(i, j) = "slots i and j vote for the same
    // track" is symmetric: 21 comparisons instead of 42, kept as the floats 1 / 0; a slot's total is the chain
    //   tot = fma(e(i, j), w[j], tot),  j ascending, its own weight added in its place,
    // the very additions of the pair loop below, bit for bit: 1 * w = w and 0 * w = +0 exactly, so each step rounds
    // tot + w resp. leaves tot alone.  A slot that does not vote ends at +0 (its sentinel id matches nobody), every voter
    // above 0: the winner is the largest total, among equal totals the smallest track id (slots of one track carry the
    // same total), taken with max / compare / min instead of the seven dependent comparisons of the walk.
    float eq[S][S];
#pragma unroll
    for (int i = 1; i < S; ++i)
#pragma unroll
      for (int j = i + 1; j < S; ++j) eq[i][j] = tv[i] == tv[j] ? 1.f : 0.f;
    float tot[S];
    float best_w = 0.f;
#pragma unroll
    for (int i = 1; i < S; ++i) {
      float t = 0.f;
#pragma unroll
      for (int j = 1; j < S; ++j) {
        if (j == i) t += wvote[j];
        else t = __builtin_fmaf(i < j ? eq[i][j] : eq[j][i], wvote[j], t);
      }
      tot[i] = t;
      best_w = fmaxf(best_w, t);
    }
    have = best_w > 0.f;
    uint32_t bt = 0xffffffffu;
#pragma unroll
    for (int i = 1; i < S; ++i) bt = min(bt, tot[i] == best_w ? trk[i] : 0xffffffffu);
    best_t = bt;  // (without a voter: some slot's id - it matches no voting slot below and is replaced by 0 further down)
#pragma unroll
    for (int j = 1; j < S; ++j) best_l = tv[j] == best_t ? lab[j] : best_l;
  } else {
    float best_w = 0.f;
#pragma unroll
    for (int i = 1; i < S; ++i) {
      float tot = 0.f;
#pragma unroll
      for (int j = 1; j < S; ++j) {
        // (a slot always matches itself.  Sharing the 21 symmetric comparisons was tried: the compiler keeps them in
        // SGPR pairs, runs out, and spills to VGPR lanes - more instructions than the 21 comparisons saved)
        if (j == i) tot += wvote[j];
        else tot += tv[j] == trk[i] ? wv[j] : 0.f;
      }
      const bool better = vote[i] && tot > 0.f && (!have || tot > best_w || (tot == best_w && trk[i] < best_t));
      best_w = better ? tot : best_w;
      best_t = better ? trk[i] : best_t;
      have = have || better;
    }
#pragma unroll
    for (int j = 1; j < S; ++j) best_l = tv[j] == best_t ? lab[j] : best_l;
  }
  best_t = have ? best_t : 0u;
  if (!any_live) {  // deleted since it was flagged, or only stale slots: the empty result
    out.wsum = 0.f;
    out.track = 0;
    out.label = 0;
    out.occ = 0.f > occ_threshold ? 1 : 0;
    nflag = (uint8_t)((any ? VF_CLEAN : VF_EMPTY) | VR_EMPTY);  // the entry holds the empty result
    return;
  }
  out.wsum = weight_sum;
  out.track = (uint16_t)best_t;  // 0 / 0 without a winner (PINNED)
  out.label = (uint8_t)best_l;
  out.occ = weight_sum > occ_threshold ? 1 : (guessed >= SDM_OCC_INIT_WEIGHT ? 2 : 0);
  if constexpr (PLAIN) {
    nflag = VF_CLEAN;
    return;
  }
  unsigned char *const rec = rec_ptr(st, S, lv);
  if (dirty_w) {
    float wout[S];
    wout[0] = wv_in[0];
#pragma unroll
    for (int i = 1; i < S; ++i) wout[i] = wv[i];
    rec_store_w<S>(rec, wout);
  }
  uint8_t flag = dirty_w ? VF_DIRTY : VF_CLEAN;  // the sum above used the unclamped weights: the next evaluation differs
  if (dirty_s) {
    uint8_t sout[S];
    sout[0] = st1[0];
    bool left = false;  // the cull may have emptied the voxel
#pragma unroll
    for (int i = 1; i < S; ++i) {
      sout[i] = (uint8_t)stv[i];
      left = left || stv[i] != ST_INVALID;
    }
    rec_store_status<S>(rec, sout);
    flag = left ? VF_DIRTY : VF_EMPTY;  // a culled slot still counted in this sum
  }
  nflag = flag;
  if (flag != VF_CLEAN) mark_tile(st, lv, remark);  // the next sweep evaluates it again resp. writes the empty result
}
template <int S, bool PLAIN, bool SINGLE = false>
__device__ __forceinline__ void occupancy_evaluate(const State &st, float occ_threshold, uint32_t remark, uint32_t lv, uint32_t smax,
                                                   const uint16_t (&ts1)[S], const uint8_t (&st1)[S], const float (&wv_in)[S],
                                                   const uint16_t (&trk16)[S], const uint8_t (&lab8)[S],
                                                   sdm_voxel_result &out) {
  uint8_t nflag;
  occupancy_evaluate_core<S, PLAIN, SINGLE>(st, occ_threshold, remark, lv, smax, ts1, st1, wv_in, trk16, lab8, out, nflag);
  st.vflag[lv] = nflag;
}

// The PLAIN evaluation and the test whether it was allowed, in one pass (the dense sweep: the test alone repeats the
// vacancy test of every slot, and a branch between the two loses the comparison masks).  Nothing is stored: returns true
// if the voxel needs the general version (then `out` / `nflag` mean nothing), which the caller runs for the whole wave.
template <int S>
__device__ __forceinline__ bool occupancy_evaluate_plain_checked(float occ_threshold, uint32_t smax, const uint16_t (&ts1)[S],
                                                                 const uint8_t (&st1)[S], const float (&wv)[S], const uint16_t (&trk16)[S],
                                                                 const uint8_t (&lab8)[S], sdm_voxel_result &out, uint8_t &nflag) {
  constexpr uint32_t LO = 0x3d4ccccdu, HI = 0x3f800000u;  // bit patterns of SDM_OCC_INIT_WEIGHT and 1.f (occupancy_is_plain)
  uint32_t lo = LO, hi = HI;
  bool guess = false, any = false, any_live = false;
  float weight_sum = 0.f;
  uint32_t trk[S], lab[S], tv[S];
  float wvote[S];
#pragma unroll
  for (int i = 1; i < S; ++i) {
    trk[i] = trk16[i];
    lab[i] = lab8[i];
    const bool present = st1[i] != ST_INVALID;
    const bool live = present && (uint32_t)ts1[i] >= smax;  // !isParticleVacant, operations.h:810-816
    any = any || present;
    any_live = any_live || live;
    guess = guess || st1[i] == ST_GUESSED_BORN;
    const uint32_t wb = live ? __float_as_uint(wv[i]) : HI;
    lo = min(lo, wb);
    hi = max(hi, wb);
    wvote[i] = live ? wv[i] : 0.f;
    weight_sum += wvote[i];  // (x + 0.f == x: the sum of the live weights in slot order)
    tv[i] = live ? trk[i] : 0xffffffffu - (uint32_t)i;
  }
  // the vote: see the PLAIN branch of occupancy_evaluate_core
  float eq[S][S];
#pragma unroll
  for (int i = 1; i < S; ++i)
#pragma unroll
    for (int j = i + 1; j < S; ++j) eq[i][j] = tv[i] == tv[j] ? 1.f : 0.f;
  float tot[S];
  float best_w = 0.f;
#pragma unroll
  for (int i = 1; i < S; ++i) {
    float t = 0.f;
#pragma unroll
    for (int j = 1; j < S; ++j) {
      if (j == i) t += wvote[j];
      else t = __builtin_fmaf(i < j ? eq[i][j] : eq[j][i], wvote[j], t);
    }
    tot[i] = t;
    best_w = fmaxf(best_w, t);
  }
  const bool have = best_w > 0.f;
  uint32_t best_t = 0xffffffffu, best_l = 0;
#pragma unroll
  for (int i = 1; i < S; ++i) best_t = min(best_t, tot[i] == best_w ? trk[i] : 0xffffffffu);
#pragma unroll
  for (int j = 1; j < S; ++j) best_l = tv[j] == best_t ? lab[j] : best_l;
  out.wsum = any_live ? weight_sum : 0.f;
  out.track = (uint16_t)(have ? best_t : 0u);
  out.label = (uint8_t)best_l;  // (0 without a voter: no slot's id matches)
  out.occ = any_live ? (weight_sum > occ_threshold ? 1 : 0) : (0.f > occ_threshold ? 1 : 0);
  nflag = any_live ? (uint8_t)VF_CLEAN : (uint8_t)((any ? VF_CLEAN : VF_EMPTY) | VR_EMPTY);
  return guess || lo < LO || hi > HI;
}

// Conservative test for the PLAIN version above: true only if no slot that could be live carries a weight above 1
// (clamp) or below the initial weight (cull candidates), and no slot at all is a guessed birth.  Stale and empty
// slots are ignored for the weights (their weight may be anything), which costs the vacancy test the evaluation
// repeats - still far cheaper than the rules it saves.
// `single` comes back true iff all live slots carry one track id (the voting slots are among the live ones).
template <int S, bool WANT_SINGLE = true>
__device__ __forceinline__ bool occupancy_is_plain(uint32_t smax, const uint16_t (&ts1)[S], const uint8_t (&st1)[S],
                                                   const float (&wv)[S], const uint16_t (&trk)[S], bool &single) {
  // (the weights' bit patterns, compared as unsigned integers: for floats >= +0 that is the float order, and a negative
  // weight, -0, an infinity or a NaN lies above the pattern of 1.f - such a voxel goes through the general version)
  constexpr uint32_t LO = 0x3d4ccccdu, HI = 0x3f800000u;  // SDM_OCC_INIT_WEIGHT, 1.f
  static_assert(SDM_OCC_INIT_WEIGHT == 0.05f, "LO is the bit pattern of the initial weight");
  uint32_t lo = LO, hi = HI;
  bool guess = false;
  uint32_t tmin = 0xffffffffu, tmax = 0u;
#pragma unroll
  for (int i = 1; i < S; ++i) {
    const bool live = st1[i] != ST_INVALID && (uint32_t)ts1[i] >= smax;
    const uint32_t w = live ? __float_as_uint(wv[i]) : HI;
    lo = min(lo, w);
    hi = max(hi, w);
    guess = guess || st1[i] == ST_GUESSED_BORN;
    if constexpr (WANT_SINGLE) {
      const uint32_t t = trk[i];
      tmin = min(tmin, live ? t : 0xffffffffu);
      tmax = max(tmax, live ? t : 0u);
    }
  }
  single = WANT_SINGLE && tmin >= tmax;  // one track id (or no live slot at all: 0xffffffff >= 0)
  return !guess && lo >= LO && hi <= HI;
}

// one voxel per lane; `mine` = this lane has a voxel to evaluate (its arrays are loaded)
template <int S>
__device__ __forceinline__ void occupancy_evaluate_wave(const State &st, float occ_threshold, uint32_t remark, bool mine, uint32_t lv, uint32_t smax,
                                                        const uint16_t (&ts1)[S], const uint8_t (&st1)[S], const float (&wv)[S],
                                                        const uint16_t (&trk)[S], const uint8_t (&lab)[S],
                                                        sdm_voxel_result &out) {
  bool single = true;
  const bool special = mine && !occupancy_is_plain<S>(smax, ts1, st1, wv, trk, single);
  if (__ballot(special) == 0ull) {  // wave-uniform
    if (__ballot(mine && !single) == 0ull) {
      if (mine) occupancy_evaluate<S, true, true>(st, occ_threshold, remark, lv, smax, ts1, st1, wv, trk, lab, out);
    } else {
      if (mine) occupancy_evaluate<S, true, false>(st, occ_threshold, remark, lv, smax, ts1, st1, wv, trk, lab, out);
    }
  } else {
    if (mine) occupancy_evaluate<S, false>(st, occ_threshold, remark, lv, smax, ts1, st1, wv, trk, lab, out);
  }
}

// Classification of a voxel from its observation stamp and flag byte alone: 0 = nothing to do, 1 = write the constant
// result `out` and the flag byte `nflag`, 2 = evaluate (phase 2).  isVoxelValid: operations.h:824-837.
__device__ __forceinline__ int occupancy_classify(uint32_t t0, uint32_t flag, uint32_t smax, float occ_threshold, int all_dirty,
                                                  sdm_voxel_result &out, uint8_t &nflag) {
  const uint32_t state = flag & VF_STATE, held = flag & VR_MASK;
  out.track = 0;
  out.label = 0;
  if (t0 == 0 || t0 < smax) {
    if (held == VR_UNOBSERVED && !all_dirty) return 0;  // the result entry already says so
    out.wsum = -1.f;
    out.occ = -1;
    // a CLEAN voxel's stored result is gone with this: it is evaluated again when the voxel is seen again
    nflag = (uint8_t)((state == VF_CLEAN ? VF_DIRTY : state) | VR_UNOBSERVED);
    return 1;
  }
  if (state == VF_EMPTY) {  // every slot INVALID: weight sum 0, no vote, nothing to clamp or cull
    if (held == VR_EMPTY && !all_dirty) return 0;
    out.wsum = 0.f;
    out.occ = 0.f > occ_threshold ? 1 : 0;
    nflag = (uint8_t)(VF_EMPTY | VR_EMPTY);
    return 1;
  }
  if (state == VF_CLEAN && !all_dirty) return 0;  // nothing it holds has changed: the result of the last sweep stands
  return 2;
}

// In-frame sweep.  A workgroup takes one tile of 2^TILE_SHIFT voxels that carries this sweep's mark (something in it was
// written or stamped since the last sweep; how the tiles are dealt out: below).  Two phases per tile.  Phase 1 streams the
// voxel stamps and flag bytes (OCC_VPT consecutive voxels per thread, everything requested before the first value is
// looked at) and finishes every voxel that is unobserved, empty or unchanged - the vast majority - from 3 bytes; a
// result entry is written only when it does not already hold that constant (bits 2-3 of the flag byte).  The others are
// listed in LDS and handled in phase 2 with all lanes busy: the voxel's record (status, slot stamps, weights, tracks,
// labels), vote, write-backs.  (Draining the list in a separate kernel was measured in round 1: the scattered fetches
// then take longer than the whole fused sweep.  Giving this kernel the cooperative record fetch of k_occupancy_dense for
// dense tiles was measured in round 2: the registers and LDS it needs cut the resident workgroups from 8 to 5 per CU
// and the launch - 8192 workgroups of which most leave at once - went from 22 to 33 us.)
constexpr int OCC_VPT = 8;  // consecutive voxels of one thread: one 16-byte load of stamps, one 8-byte load of flags
constexpr int OCC_TILE = TPB * OCC_VPT;  // voxels of one workgroup
static_assert(OCC_TILE == (1 << TILE_SHIFT), "a workgroup takes one tile of State::tile_dirty at a time");

// Which tiles?  The launch has OCC_GRID workgroups whatever the map's size, and every one of them reads the whole array of
// tile marks (8 KB for 256^3 voxels, from L2), counts the tiles that carry this sweep's epoch and takes the b-th, the
// (b + OCC_GRID)-th, ... of them: a few hundred tiles of a frame are then one tile per workgroup, all of them resident at
// once, and the workgroups without a tile leave after that scan.  (Round 2/3 launched one workgroup per tile - 8192 of
// which 7600 left after one byte - and spent 13 of the launch's 22 us on dispatching them; several tiles per workgroup
// at fixed positions were measured too: tiles in need come in clusters, 35-61 us.)  Nobody writes the array of this
// sweep's epoch while the sweep runs (mark_tile), so the scan sees the same marks in every workgroup.  Maps with more than 64 * TPB tiles: one workgroup
// per tile as before (seg = 0).
#ifndef SDM_OCC_GRID
#define SDM_OCC_GRID 1024
#endif
constexpr uint32_t OCC_GRID = SDM_OCC_GRID;
constexpr uint32_t OCC_SEG_MAX = 64;  // tile marks per thread of the scan (one 64-bit mask)

// (OCC_GRID workgroups = 4 waves per SIMD: 128 registers are free; S = 16 takes more and runs at 2)
template <int S>
__global__ __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(S <= 8 ? 4 : 2, S <= 8 ? 4 : 2))) void k_occupancy(Dims d, float occ_threshold, State st, Counters *cnt,
                                                                                             const FrameArgs *__restrict__ fa, uint32_t n_tiles, uint32_t seg) {
  __shared__ uint16_t live_list[OCC_TILE];
  __shared__ uint32_t n_live;
  __shared__ uint32_t wave_total[TPB / 64];
  __shared__ uint32_t sel_tile;
  DBG_LANE0(5, 0);
  const uint32_t epoch = fa->f.epoch, remark = next_epoch(epoch);
  // ---- the tiles this sweep has to look into: this thread's stretch of the marks as a bit mask, ranks by a block scan
  unsigned long long mask = 0;
  uint32_t my_rank = 0, n_dirty = 0;
  const uint8_t *__restrict__ marks = tile_marks(st, epoch);  // (what this sweep marks goes into the other array)
  if (seg) {
    const uint32_t t0 = threadIdx.x * seg;
    for (uint32_t j = 0; j < seg; j += 16) {
      if (t0 + j >= n_tiles) break;
      const v4u w = *reinterpret_cast<const v4u *>(marks + t0 + j);  // (the array is padded to whole stretches)
      const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (((ww[q >> 2] >> (8 * (q & 3))) & 0xffu) == epoch && t0 + j + q < n_tiles) mask |= 1ull << (j + q);
    }
    const uint32_t mine = (uint32_t)__popcll(mask);
    uint32_t inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(inc, off, 64);
      if ((threadIdx.x & 63u) >= (uint32_t)off) inc += t;
    }
    if ((threadIdx.x & 63u) == 63u) wave_total[threadIdx.x >> 6] = inc;
    __syncthreads();
    my_rank = inc - mine;
#pragma unroll
    for (int w = 0; w < TPB / 64; ++w) {
      if ((uint32_t)w < (threadIdx.x >> 6)) my_rank += wave_total[w];
      n_dirty += wave_total[w];
    }
  } else {
    n_dirty = marks[blockIdx.x] == epoch ? gridDim.x : 0;  // (rank = tile: the loop below runs once, for this tile)
  }
  if (blockIdx.x >= n_dirty) {
    DBG_LANE0(5, 1);
    return;
  }
 for (uint32_t r = blockIdx.x; r < n_dirty; r += gridDim.x) {
  uint32_t tile = blockIdx.x;
  if (seg) {
    if (r >= my_rank && r < my_rank + (uint32_t)__popcll(mask)) {
      unsigned long long m = mask;
      for (uint32_t k = my_rank; k < r; ++k) m &= m - 1ull;  // drop the set bits below the wanted one
      sel_tile = threadIdx.x * seg + (uint32_t)__builtin_ctzll(m);
    }
    if (threadIdx.x == 0) n_live = 0;
    __syncthreads();
    tile = sel_tile;
  } else {
    if (threadIdx.x == 0) n_live = 0;
    __syncthreads();
  }
  const uint32_t blk0 = tile * OCC_TILE;
  if (threadIdx.x == 0) atomicAdd(&cnt->shard[tile & (VIS_SHARDS - 1)].sweep_tiles, 1u);
  const uint32_t lv0 = blk0 + threadIdx.x * OCC_VPT;  // v_count is a multiple of 8: whole groups only
  if (lv0 < d.v_count) {
    uint16_t t0v[OCC_VPT];
    uint8_t flag[OCC_VPT];
    uint32_t sx[OCC_VPT], yz = 0;
    const bool rows = d.x_n >= 3;  // a group lies in one x row of the ring: one y and one z stamp, eight consecutive x stamps
    load_vec(t0v, st.vts + lv0);
    load_vec(flag, st.vflag + lv0);
    if (rows) {
      uint32_t rx, ry, rz;
      voxel_to_ring(d, d.v_begin + lv0, rx, ry, rz);
      const uint32_t b = st.stamps_y[ry], c = st.stamps_z[rz];
      yz = b > c ? b : c;
      load_vec(sx, st.stamps_x + rx);
    }
    uint8_t nflag[OCC_VPT];
    bool flags_changed = false;
    v2u outw[OCC_VPT];       // the constant results this thread has to write ...
    uint32_t want = 0;       // ... for these of its voxels
#pragma unroll
    for (int u = 0; u < OCC_VPT; ++u) {
      uint32_t smax;
      if (rows) {
        smax = sx[u] > yz ? sx[u] : yz;
      } else {
        uint32_t rx, ry, rz;
        voxel_to_ring(d, d.v_begin + lv0 + u, rx, ry, rz);
        smax = stamp_max(st, rx, ry, rz);
      }
      nflag[u] = flag[u];
      outw[u] = v2u{0u, 0u};
      sdm_voxel_result out;
      const int cls = occupancy_classify(t0v[u], flag[u], smax, occ_threshold, 0, out, nflag[u]);
      if (cls == 1) {
        __builtin_memcpy(&outw[u], &out, 8);
        want |= 1u << u;
        flags_changed = true;
      } else if (cls == 2) {
        live_list[atomicAdd(&n_live, 1u)] = (uint16_t)(threadIdx.x * OCC_VPT + u);
      }
    }
    if (flags_changed) store_vec(st.vflag + lv0, nflag);  // phase 2 rewrites the bytes of the listed voxels after the barrier
    if (want == (1u << OCC_VPT) - 1u) {  // the whole group (recycled slab rows): 64 contiguous bytes
      v4u *dst = reinterpret_cast<v4u *>(st.res + lv0);
#pragma unroll
      for (int u = 0; u < OCC_VPT; u += 2)
        __builtin_nontemporal_store(v4u{outw[u].x, outw[u].y, outw[u + 1].x, outw[u + 1].y}, dst + u / 2);
    } else {
#pragma unroll
      for (int u = 0; u < OCC_VPT; ++u)
        if (want & (1u << u)) __builtin_nontemporal_store(outw[u], reinterpret_cast<v2u *>(st.res + lv0 + u));
    }
  }
  __syncthreads();
  const uint32_t nl = n_live;
  if (threadIdx.x == 0 && nl) atomicAdd(&cnt->shard[tile & (VIS_SHARDS - 1)].sweep, nl);
  for (uint32_t k = threadIdx.x; k < nl; k += TPB) {
    const uint32_t lv = blk0 + live_list[k];
    uint16_t ts1[S], trk[S];
    uint8_t st1[S], lab[S];
    float wv[S];
    rec_load<S>(rec_ptr(st, S, lv), wv, ts1, trk, lab, st1);  // the whole record (one or two lines at S = 8) in one go
    uint32_t rx, ry, rz;
    voxel_to_ring(d, d.v_begin + lv, rx, ry, rz);
    const uint32_t smax = stamp_max(st, rx, ry, rz);
#ifndef SDM_AB_NO_SCHED_BARRIER
    // everything is requested before anything is looked at (left alone the compiler sinks some of the loads to their
    // first use: two more dependent round trips in a kernel that is nothing but latency)
    __builtin_amdgcn_sched_barrier(0);
#endif
    // (latency-bound here, not issue-bound: the general version only - the plain one would cost registers, i.e.
    // resident workgroups, i.e. dispatch time of the many workgroups that leave at once)
    sdm_voxel_result out;
    occupancy_evaluate<S, false>(st, occ_threshold, remark, lv, smax, ts1, st1, wv, trk, lab, out);
    store_result(st.res + lv, out);
  }
  __syncthreads();  // the list and the selection are reused by the next tile of this workgroup
 }
  DBG_LANE0(5, 3);
}

// Non-incremental sweep: every voxel's result entry is written (first sweep of a state).  Two launches:
//
// k_occupancy_scan   streams stamps and flags of every voxel like the in-frame kernel (8 consecutive voxels per thread,
//   wide loads) and finds the constant results.  Voxels that hold something: chunks of 64 voxels with fewer than
//   OCC_DENSE_MIN of them put them on the workgroup's list and they are evaluated here, one record per lane - or, in
//   the variant k_occupancy_scan_lists, handed to k_occupancy_listed, a launch in between that gives every 256 listed
//   voxels of a tile a workgroup of their own (the evaluation loop says when which); denser
//   chunks are left to the second kernel as a 64-bit mask per chunk.  All results of the tile - constant or evaluated -
//   are collected in LDS and leave as lane-linear 16-byte stores, 1 KB contiguous per instruction, every line written
//   whole and once (stored from registers they are 8-byte pieces 64 bytes apart with holes where the evaluated voxels
//   are: measured 75 us against 46 us for the same bytes on the benchmark state).  On a sparse map this launch is all
//   there is.
// k_occupancy_dense  a wave owns 8 consecutive chunks, lane = voxel, and leaves at once when none of them has a mask.
//   A chunk's 64 records are one contiguous block of 640*(S-1) bytes: fetched with lane-linear 16-byte loads (1 KB per
//   instruction), passed through LDS, where every lane picks up its own record (record stride 70 B at S = 8);
//   a per-lane fetch of the records would touch all of the block's lines with every
//   load instruction (measured: 0.70 ms for the dense case against 0.33 ms).  The loads of the wave's next two chunks
//   are in flight while one is evaluated.
// What bounds the dense case (rocprofv3 SQ counters, profiles/r02_dense_pmc.txt): not the bytes alone - the
// evaluation costs about 650 instructions per chunk and wave, the SIMDs issue for more than 80 % of the kernel's time.
// Hence the branch-free evaluation and its PLAIN variant above.
constexpr int OCC_CHUNK = 64;
constexpr int OCC_CHUNKS = OCC_TILE / OCC_CHUNK;
constexpr int OCC_WAVES = TPB / 64;
constexpr int OCC_CPW = OCC_CHUNKS / OCC_WAVES;  // chunks per wave
#ifndef SDM_OCC_DENSE_MIN
#define SDM_OCC_DENSE_MIN 32
#endif
constexpr uint32_t OCC_DENSE_MIN = SDM_OCC_DENSE_MIN;

struct OccScanInputs {  // what one thread reads of a tile: 8 voxel stamps, 8 flag bytes, the ring stamps of its row
  uint16_t t0v[OCC_VPT];
  uint8_t flag[OCC_VPT];
  uint32_t sx[OCC_VPT], yz;
};

__device__ __forceinline__ void occ_scan_fetch(const Dims &d, const State &st, uint32_t tile, uint32_t n_tiles, uint32_t tid,
                                               OccScanInputs &in) {
  const uint32_t lv0 = tile * OCC_TILE + tid * OCC_VPT;
  if (tile >= n_tiles || lv0 >= d.v_count) return;
  load_vec(in.t0v, st.vts + lv0);
  load_vec(in.flag, st.vflag + lv0);
  if (d.x_n >= 3) {  // a group lies in one x row of the ring: one y and one z stamp, eight consecutive x stamps
    uint32_t rx, ry, rz;
    voxel_to_ring(d, d.v_begin + lv0, rx, ry, rz);
    const uint32_t b = st.stamps_y[ry], c = st.stamps_z[rz];
    in.yz = b > c ? b : c;
    load_vec(in.sx, st.stamps_x + rx);
  }
}

// (registers: told to fit 8 waves per SIMD the compiler finds an allocation with 64 registers and no spills at S <= 8;
// left alone it takes 91, i.e. 5 resident workgroups per CU instead of 8 - this launch lives on resident workgroups)
template <int S, bool LISTS>
__device__ __forceinline__ void occupancy_scan_tile(const Dims &d, float occ_threshold, const State &st, Counters *cnt, unsigned long long *__restrict__ need,
                                                    uint32_t n_tiles, uint32_t remark, const uint32_t tile) {
  // the tile's results, 64 bytes (8 voxels) per thread; the 16-byte pieces of a row are swizzled so that neither the
  // row-wise writes nor the lane-linear reads run into bank conflicts
  __shared__ v4u res_stage[TPB * 4];
  __shared__ uint16_t live_list[OCC_CHUNKS * (OCC_DENSE_MIN - 1)];  // a sparse chunk lists fewer than OCC_DENSE_MIN
  static_assert(OCC_CHUNKS * (OCC_DENSE_MIN - 1) <= OCC_LIST_CAP, "a tile's list fits its segment of State::occ_list");
  __shared__ unsigned long long chunk_mask[OCC_CHUNKS];
  __shared__ uint32_t n_live;
  auto stage_slot = [&](uint32_t v) -> v2u * {  // where voxel v of the tile sits in the stage
    const uint32_t t = v >> 3, q = (v >> 1) & 3u;
    return reinterpret_cast<v2u *>(&res_stage[t * 4 + (q ^ ((t >> 1) & 3u))]) + (v & 1u);
  };
  // (One tile per workgroup.  A persistent version - 1024 workgroups walking the tiles, the next tile's inputs pulled
  // into L2 or registers meanwhile - was measured: what the loop keeps alive costs 30 registers, one resident workgroup
  // per CU less, 90 us against 75 us on the benchmark state.)
  const uint32_t tid = threadIdx.x;
  // a wave's 512 voxels whose chunks were all dense in the last non-incremental sweep are left to the second launch
  // whole: that one then classifies them itself (State::grp_hint - a hint about speed only; whatever the voxels hold by
  // now, the result is the same).  A tile of four such groups: nothing to do here.
  static_assert(OCC_WAVES == 4 && OCC_CPW * OCC_CHUNK == 512, "one hint byte per wave of this kernel and of k_occupancy_dense");
  const uint32_t hint4 = reinterpret_cast<const uint32_t *>(st.grp_hint)[tile];
  if (hint4 == 0x01010101u) return;
  if (tid == 0) st.occ_shard[OCC_LIST_SHARDS].aux[0] = 1u;  // this launch was needed (k_occupancy_dense passes it on)
  const bool hinted = (hint4 >> (8 * (tid >> 6))) & 1u;
  const uint32_t lane = tid & 63u, wave = tid >> 6;
  if (tid == 0) n_live = 0;
  OccScanInputs in;
  if (!hinted) occ_scan_fetch(d, st, tile, n_tiles, tid, in);
  __syncthreads();
  {
    const uint32_t blk0 = tile * OCC_TILE;
    if (tid == 0) atomicAdd(&cnt->shard[tile & (VIS_SHARDS - 1)].sweep_tiles, 1u);
    const uint32_t lv0 = blk0 + tid * OCC_VPT;  // v_count is a multiple of 8: whole groups only
    const bool in_range = lv0 < d.v_count && !hinted;
    v2u outw[OCC_VPT];
    uint32_t want = 0, listed = 0;
    uint8_t nflag[OCC_VPT] = {};
    bool flags_changed = false;
    if (in_range) {
      const bool rows = d.x_n >= 3;
#pragma unroll
      for (int u = 0; u < OCC_VPT; ++u) {
        uint32_t smax;
        if (rows) {
          smax = in.sx[u] > in.yz ? in.sx[u] : in.yz;
        } else {
          uint32_t rx, ry, rz;
          voxel_to_ring(d, d.v_begin + lv0 + u, rx, ry, rz);
          smax = stamp_max(st, rx, ry, rz);
        }
        nflag[u] = in.flag[u];
        outw[u] = v2u{0u, 0u};
        sdm_voxel_result out;
        // (non-incremental: 1 or 2, never 0)
        const int cls = occupancy_classify(in.t0v[u], in.flag[u], smax, occ_threshold, 1, out, nflag[u]);
        if (cls == 1) {
          __builtin_memcpy(&outw[u], &out, 8);
          want |= 1u << u;
          flags_changed = true;
        } else {
          listed |= 1u << u;
        }
      }
    }
    // which way do this thread's listed voxels go?  Its chunk = the aligned group of eight lanes it sits in.
    bool whole;  // every result of this thread's group is known in this launch: it leaves through the stage
    {
      uint32_t in_chunk = (uint32_t)__popc(listed);
      in_chunk += __shfl_xor(in_chunk, 1, 64);
      in_chunk += __shfl_xor(in_chunk, 2, 64);
      in_chunk += __shfl_xor(in_chunk, 4, 64);
      const bool dense = in_chunk >= OCC_DENSE_MIN;
      whole = in_range && (!dense || listed == 0);
      reinterpret_cast<uint8_t *>(chunk_mask)[tid] = dense ? (uint8_t)listed : (uint8_t)0;
      if (!dense && listed) {
        uint32_t k = atomicAdd(&n_live, (uint32_t)__popc(listed));
#pragma unroll
        for (int u = 0; u < OCC_VPT; ++u)
          if (listed & (1u << u)) live_list[k++] = (uint16_t)(tid * OCC_VPT + u);
      }
      uint32_t nw = (uint32_t)__popc(listed);
      for (int off = 32; off > 0; off >>= 1) nw += __shfl_down(nw, off, 64);
      if (lane == 0 && nw) atomicAdd(&cnt->shard[(tile + wave) & (VIS_SHARDS - 1)].sweep, nw);
      // The flag bytes leave here, eight per thread (the evaluation of a listed voxel rewrites its byte later).  Those of
      // the voxels left to the second launch are set to what that launch nearly always decides - VF_CLEAN - so that it
      // stores a flag byte only where it decides otherwise: one-byte stores of a whole map cost a tenth of the dense
      // sweep's time on some boxes (tools/probes/ring_probe.hip), and a sweep that finds the bytes as it leaves them
      // writes none.
      if (dense && listed) {
#pragma unroll
        for (int u = 0; u < OCC_VPT; ++u)
          if ((listed & (1u << u)) && nflag[u] != VF_CLEAN) {
            nflag[u] = VF_CLEAN;
            flags_changed = true;
          }
      }
      if (flags_changed) store_vec(st.vflag + lv0, nflag);
    }
    if (whole) {  // constant results into the stage (the slots of listed voxels are filled in by the evaluation)
      const uint32_t sw = (tid >> 1) & 3u;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        res_stage[tid * 4 + (q ^ sw)] = v4u{outw[2 * q].x, outw[2 * q].y, outw[2 * q + 1].x, outw[2 * q + 1].y};
    } else if (in_range) {  // a group with voxels the second launch evaluates: its constant results one by one
#pragma unroll
      for (int u = 0; u < OCC_VPT; ++u)
        if (want & (1u << u)) __builtin_nontemporal_store(outw[u], reinterpret_cast<v2u *>(st.res + lv0 + u));
    }
    __syncthreads();
    if (tid < OCC_CHUNKS && blk0 + tid * OCC_CHUNK < d.v_count)
      need[(size_t)tile * OCC_CHUNKS + tid] = chunk_mask[tid];
    const uint32_t nl = n_live;
    // The voxels the tile's sparse chunks listed.  Two ways, chosen by the host per sweep from what the sweep before it
    // found (State::occ_shard; every voxel gets the same result either way):
    //
    // LISTS - the tile hands its list over (State::occ_list, one unit of State::occ_unit per OCC_LIST_UNIT entries) and
    // k_occupancy_listed, the launch behind this one, gives every unit a workgroup of its own, all at once; the result
    // slots leave here as zeros with the rest of the tile.  This is for maps the filter grew: particles on surfaces, a
    // few hundred tiles with dozens to hundreds of listed voxels among thousands with none.  Evaluated here, each such
    // tile went round trip, evaluation, round trip, evaluation while its 16 KB of results waited in LDS, and the launch
    // ran on when the rest had drained: 65 us against 40 us for the empty map on the `driven` map (36 K voxels that hold
    // something), 41 + 7 us with the lists.
    //
    // not LISTS - the loop below.  For a map where no tile or nearly every tile lists something the extra launch is 4 us
    // for nothing or worse: on the benchmark map (291 K voxels that hold something, strewn evenly - 35 per tile, none
    // in a dense chunk) the list launch takes 32 us - 17 us for 291 K records of 80 bytes at random places (two 128-byte
    // lines each: 73 MB, DRAM rows opened for 128 bytes), 10 us for the 8-byte results written into lines this launch
    // has just stored, 1.5 us for the evaluation itself - 83 us in all against 75 with the loop, which hides part of
    // the records behind the stream (41 us for this launch with the loop cut out, 65 us with it).  Also measured there
    // and not kept (tools/gpu_full_split.sh): the rows nothing is evaluated in stored before the loop (73 against 68);
    // the plain evaluation with its admission test (spills at 64 registers, 74 against 69 at 72); the loop dealt out
    // to the four waves (79 against 73) or to the wave tile % 4 (no change).
    if constexpr (LISTS) {
      if (nl) {
        if (tid == 0) {
          // (a bijection on every 64 consecutive tiles - no shard gets more than its share - that also spreads tiles
          // which agree in their low bits: the tiles of a ground plane do, and went to two shards)
          const uint32_t sh = (tile ^ (tile >> 6) ^ (tile >> 12)) & (OCC_LIST_SHARDS - 1);
          st.occ_list_n[tile] = nl;
          const uint32_t units = (nl + OCC_LIST_UNIT - 1) / OCC_LIST_UNIT;
          const uint32_t slot = (uint32_t)atomicAdd(&st.occ_shard[sh].word, (1ull << 32) | units);  // (a tile and its units)
          for (uint32_t u = 0; u < units && slot + u < st.occ_unit_cap; ++u)
            st.occ_unit[(size_t)sh * st.occ_unit_cap + slot + u] = make_uint2(tile, u * OCC_LIST_UNIT);
        }
        for (uint32_t k = tid; k < nl; k += TPB) st.occ_list[(size_t)tile * OCC_LIST_CAP + k] = live_list[k];
      }
    } else {
      // (whether lists would pay next time is a matter of one tile in eight, counted eightfold: an atomic per tile, 128 to
      // a line, was 3 us of this launch on a map where every tile lists something)
      if (nl && tid == 0 && ((tile * 2654435761u) >> 29) == 0u)
        atomicAdd(&st.occ_shard[(tile ^ (tile >> 6) ^ (tile >> 12)) & (OCC_LIST_SHARDS - 1)].word, 8ull << 32);
#ifdef SDM_SCAN_NOEVAL  // (development, timing only: the launch without its evaluations)
      if (nl) return;
#endif
#pragma unroll 1
      for (uint32_t k0 = 0; k0 < nl; k0 += TPB) {  // workgroup-uniform
        const uint32_t k = k0 + tid;
        if (k < nl) {
          const uint32_t tv = live_list[k], lv = blk0 + tv;
          uint16_t ts1[S], trk[S];
          uint8_t st1[S], lab[S];
          float wv[S];
          rec_load<S>(rec_ptr(st, S, lv), wv, ts1, trk, lab, st1);  // the whole record (one or two lines at S = 8) in one go
          uint32_t rx, ry, rz;
          voxel_to_ring(d, d.v_begin + lv, rx, ry, rz);
          const uint32_t smax = stamp_max(st, rx, ry, rz);
          // everything of the record is requested before anything is looked at: left alone, the compiler (held to 64
          // registers) sank each load to its first use - five dependent round trips per listed voxel instead of one
          __builtin_amdgcn_sched_barrier(0);
          sdm_voxel_result out;
          // (the general version only: the plain one next to it costs registers and was measured to gain nothing here)
          occupancy_evaluate<S, false>(st, occ_threshold, remark, lv, smax, ts1, st1, wv, trk, lab, out);
          v2u o;
          __builtin_memcpy(&o, &out, 8);
          *stage_slot(tv) = o;
        }
      }
      __syncthreads();
    }
    // the stage leaves as lane-linear 16-byte stores: 1 KB contiguous per instruction
    {
      const unsigned long long whole_mask = __ballot(whole);
      v4u *dst = reinterpret_cast<v4u *>(st.res + blk0 + wave * 64 * OCC_VPT);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t row = k * 16 + (lane >> 2), piece = lane & 3u;  // row = the lane of this wave whose group it is
        if ((whole_mask >> row) & 1ull) {
          const uint32_t t = wave * 64 + row;
          __builtin_nontemporal_store(res_stage[t * 4 + (piece ^ ((t >> 1) & 3u))], dst + row * 4 + piece);
        }
      }
    }
  }
}

#define SCAN_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(S <= 8 ? 8 : 4, S <= 8 ? 8 : 4)))
// (Measured and not kept, round 6: four consecutive tiles per workgroup, so that a map whose groups are all hinted costs a
// quarter of the workgroups that leave after their hint bytes - 0.3 us of the dense case's 245, and the empty and the
// benchmark map went from 45 / 70 us to 125 / 155: a workgroup that walks four tiles holds its slot four times as long,
// and the launch lives on workgroups that come and go.)
template <int S, bool LISTS>
__device__ __forceinline__ void occupancy_scan_tiles(const Dims &d, float occ_threshold, const State &st, Counters *cnt,
                                                     unsigned long long *__restrict__ need, uint32_t n_tiles, uint32_t remark) {
  occupancy_scan_tile<S, LISTS>(d, occ_threshold, st, cnt, need, n_tiles, remark, blockIdx.x);
}
template <int S>
__global__ __launch_bounds__(TPB) SCAN_WAVES_ATTR void k_occupancy_scan(Dims d, float occ_threshold, State st, Counters *cnt,
                                                                       unsigned long long *__restrict__ need, uint32_t n_tiles, uint32_t remark) {
  occupancy_scan_tiles<S, false>(d, occ_threshold, st, cnt, need, n_tiles, remark);
}
template <int S>
__global__ __launch_bounds__(TPB) SCAN_WAVES_ATTR void k_occupancy_scan_lists(Dims d, float occ_threshold, State st, Counters *cnt,
                                                                             unsigned long long *__restrict__ need, uint32_t n_tiles, uint32_t remark) {
  occupancy_scan_tiles<S, true>(d, occ_threshold, st, cnt, need, n_tiles, remark);
}

#ifdef SDM_DENSE_WAVES
#define DENSE_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(SDM_DENSE_WAVES, SDM_DENSE_WAVES)))
#else
#define DENSE_WAVES_ATTR
#endif
// (Instruction issue bounds this kernel - the SIMDs issue for 86 % of its time on a dense map, profiles/README.md round 5 -
// so what is not the evaluation is kept off the vector unit: the wave index goes through readfirstlane, which makes a
// chunk's base address a scalar and its loads `global_load ..., v_lane_offset, s[base]`; the record array is allocated one
// chunk longer than the map, so no load clamps its address; the plain evaluation runs together with its own admission
// test, occupancy_evaluate_plain_checked.)
// A wave whose group of 512 voxels has State::grp_hint set (every chunk of it was dense in the last non-incremental
// sweep) finds nothing from k_occupancy_scan: it classifies its voxels itself - stamp and flag byte of the lane's voxel requested with the
// chunk's records, occupancy_classify in the step - takes every chunk as dense and writes all 64 results of a chunk, the
// constant ones included.  That is right whatever the group holds (only slow if it has emptied since: every record is
// fetched), and the hint is set anew from what this sweep finds.  On a dense map the first launch is then 8192 workgroups
// that leave after four bytes instead of 24 us of classification in front of a dependent launch.
// The units k_occupancy_scan_lists left (State::occ_unit): OCC_LIST_UNIT listed voxels of one tile per workgroup,
// one record per lane, fetched by the lane (80 bytes in five loads, all requested before the first is looked at), the
// plain evaluation with its admission test and the general one for a wave that holds a voxel the rare rules apply to.
// The result overwrites the zero the first launch stored, the flag byte is stored whatever it is.
#ifndef SDM_LISTED_GRID
#define SDM_LISTED_GRID 2048
#endif
constexpr int OCC_LISTED_GRID = SDM_LISTED_GRID;
template <int S>
__device__ __forceinline__ void occupancy_listed_units(const Dims &d, float occ_threshold, const State &st, uint32_t remark, uint32_t bid) {
  static_assert(OCC_LIST_UNIT == TPB, "one entry per thread");
  static_assert(OCC_LISTED_GRID % OCC_LIST_SHARDS == 0, "a workgroup stays with one shard's units");
  const uint32_t sh = bid & (OCC_LIST_SHARDS - 1);
  uint32_t n_units = (uint32_t)st.occ_shard[sh].word;
  n_units = n_units < st.occ_unit_cap ? n_units : st.occ_unit_cap;
#pragma unroll 1
  for (uint32_t h = bid / OCC_LIST_SHARDS; h < n_units; h += OCC_LISTED_GRID / OCC_LIST_SHARDS) {
    const uint2 un = st.occ_unit[(size_t)sh * st.occ_unit_cap + h];
    const uint32_t tile = un.x;
    uint32_t nl = st.occ_list_n[tile];
    nl = nl < OCC_LIST_CAP ? nl : OCC_LIST_CAP;
    const uint32_t k = un.y + threadIdx.x;
    if (k < nl) {
      const uint32_t tv = st.occ_list[(size_t)tile * OCC_LIST_CAP + k];
      const uint32_t lv = tile * OCC_TILE + tv;
      uint16_t ts1[S], trk[S];
      uint8_t st1[S], lab[S];
      float wv[S];
      rec_load<S>(rec_ptr(st, S, lv), wv, ts1, trk, lab, st1);  // the whole record (one or two lines at S = 8) in one go
      uint32_t rx, ry, rz;
      voxel_to_ring(d, d.v_begin + lv, rx, ry, rz);
      const uint32_t smax = stamp_max(st, rx, ry, rz);
      __builtin_amdgcn_sched_barrier(0);
      sdm_voxel_result out;
      uint8_t nf;
      const bool special = occupancy_evaluate_plain_checked<S>(occ_threshold, smax, ts1, st1, wv, trk, lab, out, nf);
      if (__ballot(special) != 0ull) occupancy_evaluate_core<S, false>(st, occ_threshold, remark, lv, smax, ts1, st1, wv, trk, lab, out, nf);
      store_result(st.res + lv, out);
      st.vflag[lv] = nf;
    }
  }
}

// the sweep's units are done: the counters go back to zero, and how many tiles listed anything becomes the hint for the
// next non-incremental sweep (first wave of one workgroup)
__device__ __forceinline__ void occupancy_listed_wrap(const State &st, uint32_t n_tiles, bool scan_ran) {
  static_assert(OCC_LIST_SHARDS == 64, "one shard per lane of the first wave");
  uint32_t t = (uint32_t)(st.occ_shard[threadIdx.x].word >> 32);
  st.occ_shard[threadIdx.x].word = 0;
  for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
  // 2: few tiles listed anything - surfaces - and the lists pay; 1: none did or most did, two launches do
  if (threadIdx.x == 0) {
    State::OccListShard &w = st.occ_shard[OCC_LIST_SHARDS];
    w.word = t > 0 && t <= n_tiles / 4 ? 2u : 1u;
    if (scan_ran) {  // did k_occupancy_scan enter any tile?  If not, the next sweep can do without it.  (When it did not
      w.aux[1] = w.aux[0] ? 1u : 2u;  // run, the waves of THIS launch say so when they find a group that is not dense.)
      w.aux[0] = 0u;
    }
  }
}
template <int S>
__global__ __launch_bounds__(TPB) void k_occupancy_listed(Dims d, float occ_threshold, State st, uint32_t remark) {
  occupancy_listed_units<S>(d, occ_threshold, st, remark, blockIdx.x);
}

// (Round 6, measured and not kept - the kernel is where the memory system and the vote's 320 vector instructions per chunk
// leave it, 0.242-0.255 ms over the round's boxes for 1.36 GB: its arguments pinned in scalar register pairs of their own -
// the compiler parks eight-word argument groups in the lanes of a vector register when the vote's comparison masks crowd
// the scalar file, and brings a whole group back, eight v_readlane, for one store's base address: 56 -> 38 v_readlane per
// step, no change in time; the step loop compiled twice, hinted and not, x rows and not: 138 registers; a wave's results
// kept in LDS and stored in one 4 KB burst when its eight chunks are through: 4 % faster where every voxel holds one track
// id - the cheaper vote -, 2 % slower on the eight-track case; tools/gpu_dense_ab.sh.)
template <int S>
__global__ __launch_bounds__(TPB) DENSE_WAVES_ATTR void k_occupancy_dense(Dims d, float occ_threshold, State st, Counters *cnt,
                                                         const unsigned long long *__restrict__ need, uint32_t remark, uint32_t n_tiles_all) {
  // (top bit of n_tiles_all: k_occupancy_scan did not run - launch_occupancy, OCC_SKIP_SCAN - and every group is taken as hinted)
  const bool all_groups = n_tiles_all >> 31;
  if (blockIdx.x == 0 && threadIdx.x < OCC_LIST_SHARDS) occupancy_listed_wrap(st, n_tiles_all & 0x7fffffffu, !all_groups);  // (the units are done: the launch before this one)
  constexpr int REC = 10 * (S - 1);              // bytes of one record
  static_assert((OCC_CHUNK * REC) % 128 == 0, "a chunk of records starts on a cache line");
  constexpr int PIECES = OCC_CHUNK * REC / 16;   // 16-byte pieces of one chunk of records
  constexpr int PPL = (PIECES + 63) / 64;
  __shared__ v4u rec_stage[OCC_WAVES][PPL * 64];  // one chunk of records per wave, for the lane <-> record transposition
  __shared__ uint32_t yz_stage[OCC_WAVES][OCC_CPW];      // max(y stamp, z stamp) of each chunk of the wave
  __shared__ v4u meta_t[OCC_WAVES][64];                  // fused: observation stamps of the wave's 512 voxels ...
  __shared__ v2u meta_f[OCC_WAVES][64];                  // ... and their flag bytes (two wide loads per wave, not two narrow ones per chunk)
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t cw = (blockIdx.x * OCC_WAVES + wave) * OCC_CPW;  // first chunk of this wave (a workgroup = a tile)
  const uint32_t lvw = cw * OCC_CHUNK;                            // ... and its first voxel
  const uint32_t hint4 = reinterpret_cast<const uint32_t *>(st.grp_hint)[blockIdx.x];  // the tile's four groups, as k_occupancy_scan saw them
  const bool fused = ((hint4 >> (8 * wave)) & 1u) || all_groups;                         // wave-uniform
  // the masks of the wave's chunks: lane k holds chunk k's (fused: every chunk of the map counts as dense)
  unsigned long long mk = 0;
  if (lane < (uint32_t)OCC_CPW && lvw + lane * OCC_CHUNK < d.v_count) mk = fused ? ~0ull : need[cw + lane];
  const uint32_t densebits = (uint32_t)(__ballot(mk != 0ull) & ((1ull << OCC_CPW) - 1ull));
  uint32_t hint_ok = 1;  // every chunk of this wave that lies in the map is dense
  {
    const uint32_t in_map = lvw >= d.v_count ? 0u : (d.v_count - lvw + OCC_CHUNK - 1) / OCC_CHUNK;
    const uint32_t mapbits = in_map >= (uint32_t)OCC_CPW ? (1u << OCC_CPW) - 1u : (1u << in_map) - 1u;
    if (!fused && densebits != mapbits) hint_ok = 0;
  }
  uint32_t n_eval = 0;
  if (!densebits) return;  // nothing of this wave's was left to this kernel (its hint is and stays 0)
  {
  if (fused) {  // (lanes beyond the end of the map read the arrays' padding)
    meta_t[wave][lane] = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(st.vts + lvw) + lane);
    meta_f[wave][lane] = __builtin_nontemporal_load(reinterpret_cast<const v2u *>(st.vflag + lvw) + lane);
  }
  uint32_t evalbits = 0;   // bit k: this lane's voxel of chunk k is to be evaluated (fused: decided in the step)
#pragma unroll
  for (int k = 0; k < OCC_CPW; ++k) {
    const unsigned long long m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mk >> 32), k) << 32) |
                                 (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mk, k);
    evalbits |= (uint32_t)((m >> lane) & 1ull) << k;
  }
  // Slab stamps.  The stamp of this lane's voxel of a chunk travels with the chunk's records (loaded where the
  // evaluation needs it, it was a dependent L2 round trip in every step) - and nothing may be computed from it where it is
  // requested: a max over the three axis stamps right behind their loads made the compiler wait for ALL outstanding
  // loads in every fetch.  With x_n >= 6 a chunk lies in one x row of the ring: its y and z stamps are one value per
  // chunk, fetched once per wave up front (yz_stage), and a fetch requests the x stamps only and leaves them raw; the max
  // is taken in the step.  Smaller maps: the per-lane max, as before.
  const bool rows = d.x_n >= 6;
  if (lane < (uint32_t)OCC_CPW) {
    uint32_t yzv = 0;  // (0 where the staged value is the max already)
    if (rows && lvw + lane * OCC_CHUNK < d.v_count) {
      uint32_t rx, ry, rz;
      voxel_to_ring(d, d.v_begin + lvw + lane * OCC_CHUNK, rx, ry, rz);
      const uint32_t b = st.stamps_y[ry], c = st.stamps_z[rz];
      yzv = b > c ? b : c;
    }
    yz_stage[wave][lane] = yzv;
  }
  // lane-linear loads of chunk k's records: 1 KB contiguous per instruction, all of them unconditional and in one basic
  // block (a load under a branch of its own is a basic block of its own, and the compiler then cannot count the loads
  // in flight).  Lanes beyond the end of the map read the arrays' padding.
  struct Landing {
    v4u buf[PPL];
    uint32_t smax;   // x stamp (rows) or slab stamp of the lane's voxel
  };
  auto fetch = [&](int k, Landing &l) {
    const uint32_t lvc = lvw + k * OCC_CHUNK;  // scalar
    if (rows) {
      l.smax = st.stamps_x[((d.v_begin + lvc) & (d.NX - 1)) + lane];
    } else {
      uint32_t rx, ry, rz;
      voxel_to_ring(d, d.v_begin + lvc + lane, rx, ry, rz);
      l.smax = stamp_max(st, rx, ry, rz);
    }
    const v4u *src = reinterpret_cast<const v4u *>(st.rec + (size_t)lvc * REC);
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      const uint32_t idx = j * 64 + lane;
      l.buf[j] = __builtin_nontemporal_load(src + (PIECES % 64 == 0 ? idx : (idx < (uint32_t)PIECES ? idx : (uint32_t)PIECES - 1u)));
    }
  };
  auto to_stage = [&](const Landing &l) {
#pragma unroll
    for (int j = 0; j < PPL; ++j) rec_stage[wave][j * 64 + lane] = l.buf[j];
  };
  auto next_dense = [&](int k) -> int {  // first dense chunk after k
    const uint32_t rest = densebits & ~((2u << k) - 1u);
    return rest ? __builtin_ctz(rest) : OCC_CPW;
  };
  // two chunks of the wave are under way at any time: one in the stage, one in registers (landing while the one in the
  // stage is evaluated).  A second register buffer was there until round 4 and bought nothing: a step takes the wave
  // several microseconds (its SIMD is shared by four), the next chunk has landed long before.
  int k = __builtin_ctz(densebits);
  int k1 = next_dense(k);
  Landing b0 = {};
  uint32_t s_cur = 0;  // stamps of the chunk in the stage
  {
    Landing first;
    fetch(k, first);
    if (k1 < OCC_CPW) fetch(k1, b0);
    to_stage(first);
    s_cur = first.smax;
  }
  // one step: evaluate chunk k out of the stage, move the next dense chunk (landed or landing) into the stage,
  // start the loads of the one after it
#pragma unroll 1
  for (int it = 0; it < OCC_CPW; ++it) {  // (a fixed trip count: with `while (k < OCC_CPW)` the compiler rotated the loop and left a wait for all loads on the back edge)
    if (k >= OCC_CPW) break;
    const uint32_t lv = lvw + k * OCC_CHUNK + lane;
    uint16_t ts1[S], trk[S];
    uint8_t st1[S], lab[S];
    float wv[S];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {  // the lane's record out of the stage (2-byte aligned: under-aligned ds_read_b128s)
      RecRaw<S> raw;
      rec_fetch<S, false>(raw, reinterpret_cast<const unsigned char *>(rec_stage[wave]) + lane * REC);
      rec_unpack<S>(raw, wv, ts1, trk, lab, st1);
    }
    uint32_t smk = s_cur;
    {
      const uint32_t yz = yz_stage[wave][k];
      smk = smk > yz ? smk : yz;
    }
    bool mine = (evalbits >> k) & 1u, store_res = mine;
    sdm_voxel_result out;
    out.wsum = 0.f;
    out.track = 0;
    out.label = 0;
    out.occ = 0;
    uint8_t nf = VF_CLEAN;
    uint32_t oldflag = VF_CLEAN;  // (not fused: k_occupancy_scan has left VF_CLEAN in the bytes of the voxels it handed over)
    if (fused) {  // wave-uniform
      oldflag = reinterpret_cast<const uint8_t *>(meta_f[wave])[k * OCC_CHUNK + lane];
      const uint32_t t0 = reinterpret_cast<const uint16_t *>(meta_t[wave])[k * OCC_CHUNK + lane];
      const bool valid = lv < d.v_count;
      const int cls = occupancy_classify(t0, oldflag, smk, occ_threshold, 1, out, nf);  // non-incremental: 1 or 2
      if (cls == 2) nf = (uint8_t)oldflag;
      mine = valid && cls == 2;
      store_res = valid;
      const uint32_t n = (uint32_t)__popcll(__ballot(mine));
      n_eval += n;
      if (n < OCC_DENSE_MIN) hint_ok = 0;
    }
    // the evaluation runs on registers only; the chunk behind this one is landing meanwhile
    {
      sdm_voxel_result oe;
      uint8_t ne;
      const bool special = occupancy_evaluate_plain_checked<S>(occ_threshold, smk, ts1, st1, wv, trk, lab, oe, ne) && mine;
      if (__ballot(special) != 0ull) {  // wave-uniform and rare: clamp, cull, guessed births
        if (mine) occupancy_evaluate_core<S, false>(st, occ_threshold, remark, lv, smk, ts1, st1, wv, trk, lab, oe, ne);
      }
      if (mine) {
        out = oe;
        nf = ne;
      }
    }
    if (store_res) {
      store_result(st.res + lv, out);
      if (nf != oldflag) st.vflag[lv] = nf;
    }
    // the next chunk moves into the stage (its loads have had this evaluation's time) and the loads of the one after it start
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    k = k1;
    k1 = k1 < OCC_CPW ? next_dense(k1) : OCC_CPW;
    if (k < OCC_CPW) {
      to_stage(b0);
      s_cur = b0.smax;
    }
    if (k1 < OCC_CPW) fetch(k1, b0);
  }
  }
  // the group's hint for the next non-incremental sweep, and (fused) the counters k_occupancy_scan keeps for what it takes
  if (lane == 0) {
    const bool whole = lvw + OCC_CPW * OCC_CHUNK <= d.v_count;  // (whole groups only)
    const uint8_t h = hint_ok && whole ? 1 : 0;
    if ((uint8_t)((hint4 >> (8 * wave)) & 1u) != h) st.grp_hint[blockIdx.x * OCC_WAVES + wave] = h;
    if (all_groups && !h) st.occ_shard[OCC_LIST_SHARDS].aux[1] = 1u;  // the next sweep needs its first launch again
    if (fused) {
      if (n_eval) atomicAdd(&cnt->shard[(blockIdx.x + wave) & (VIS_SHARDS - 1)].sweep, n_eval);
      if (wave == 0 && (hint4 == 0x01010101u || all_groups)) atomicAdd(&cnt->shard[blockIdx.x & (VIS_SHARDS - 1)].sweep_tiles, 1u);  // (the scan counts the tiles it enters)
    }
  }
}

// after sdm_load_state: the "something here" flag of every voxel from the status rows (the voxel stamps were taken from
// slot 0 of the imported stamp array by k_rec_pack: sdm_dump_state / sdm_load_state keep the reference's layout, the time
// particle is slot 0 of the voxel, buffer.h:57-79)
__global__ __launch_bounds__(TPB) void k_vflag_from_records(Dims d, State st) {
  uint32_t lv = blockIdx.x * blockDim.x + threadIdx.x;
  if (lv >= d.v_count) return;
  bool any = false;
  for (uint32_t i = 1; i < d.S; ++i) any = any || slot_ref(st, d.S, lv, i).status() != ST_INVALID;
  st.vflag[lv] = any ? VF_DIRTY : VF_EMPTY;
}

// ------------------------------------------------------------------------------------ A6
// The reference reaches the voxels it updates by a BFS over 6-connected in-frustum grid VERTICES starting at the
// vertex under the point 1 m in front of the camera (mc_ring/operations.h:1312-1456).  The set it reaches is the
// connected component of that vertex among the in-frustum vertices.  Here:
//   1. k_vertex_mask    in-frustum bit of every vertex of the conservative frustum box, 64 per wave (ballot);
//   2. k_line_info      per x-line: non-empty?, its bits one contiguous run ("simple")?, does it share an x with the
//                       next line in y / in z? -> three small bitmaps over (y,z);
//   3. k_flood2d        when every line is simple, a line is reached as a whole or not at all, so the component is a
//                       flood fill over LINES in the (y,z) plane: one workgroup, reach bitmap in LDS, word-parallel
//                       fills along y, carry sweeps along z, until nothing changes;
//   4. k_flood_generic  only if some line was not simple (float rounding exactly on a frustum plane): the plain 3-D
//                       bit flood, one workgroup, exact for any mask.
// The visibility pass reads "reached" as (line reached & in-frustum bit) in the simple case, the flooded bits otherwise.
// Both paths are compared with the oracle's literal BFS in the parity tests.

// Frustum vertex mask (isPointInFrustum, operations.h:1240-1258, 1338-1340): one wave per 64 vertices along x.
__global__ __launch_bounds__(TPB) void k_vertex_mask(Dims d, const FrameArgs *__restrict__ fa, uint64_t *__restrict__ M, int wpl,
                                                      Counters *cnt) {
  const Frame f = fa->f;  // a copy (uniform registers): stores of the kernel cannot alias it
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // first kernel of the frustum chain: its flags start clean
    cnt->flood_complex = 0;
    cnt->flood_rounds = 0;
    cnt->start_in_frustum = 0;
  }
  const int VY = d.NY + 1;
  const int ny = f.bb1[1] - f.bb0[1] + 1, nz = f.bb1[2] - f.bb0[2] + 1;
  const int lane = threadIdx.x & 63;
  // the grid does not depend on the frame (the box does): waves stride over the box's words
  for (uint32_t gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; gw < (uint32_t)ny * nz * wpl; gw += (gridDim.x * blockDim.x) >> 6) {
  int xw = gw % wpl;
  int l = gw / wpl;
  int y = f.bb0[1] + l % ny, z = f.bb0[2] + l / ny;
  int x = xw * 64 + lane;
  bool in = false;
  if (x >= f.bb0[0] && x <= f.bb1[0]) {
    // vertex position: idx * size + (map_center + map_min), operations.h:1304,1338
    float gx = (float)x * d.voxel_size + (f.center[0] + d.pmin[0]);
    float gy = (float)y * d.voxel_size + (f.center[1] + d.pmin[1]);
    float gz = (float)z * d.voxel_size + (f.center[2] + d.pmin[2]);
    in = point_in_frustum(d, f, gx, gy, gz);
  }
  uint64_t mask = __ballot(in);
  if (lane == 0) M[((size_t)z * VY + y) * wpl + xw] = mask;
  }
}

constexpr int MAX_WPL = 9;  // 513 vertices along an axis at most (x_n, y_n <= 9)

// per-line summary bitmaps over (y,z); one wave per 64 lines along y, ballot-packed.  wy = words per z row.
__global__ __launch_bounds__(TPB) void k_line_info(Dims d, const FrameArgs *__restrict__ fa, const uint64_t *__restrict__ M, int wpl,
                                                   int wy, uint64_t *__restrict__ NE, uint64_t *__restrict__ EY,
                                                   uint64_t *__restrict__ EZ, Counters *cnt) {
  const Frame f = fa->f;  // a copy (uniform registers): stores of the kernel cannot alias it
  const int VY = d.NY + 1;
  const int yw0 = f.bb0[1] >> 6, yw1 = f.bb1[1] >> 6;
  const int nyw = yw1 - yw0 + 1, nz = f.bb1[2] - f.bb0[2] + 1;
  const int lane = threadIdx.x & 63;
  for (uint32_t gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; gw < (uint32_t)nyw * nz; gw += (gridDim.x * blockDim.x) >> 6) {
  const int yw = yw0 + gw % nyw, z = f.bb0[2] + gw / nyw;
  const int y = yw * 64 + lane;
  bool ne = false, ey = false, ez = false, complex_line = false;
  if (y >= f.bb0[1] && y <= f.bb1[1]) {
    const uint64_t *m = M + ((size_t)z * VY + y) * wpl;
    const bool has_y = y + 1 <= f.bb1[1], has_z = z + 1 <= f.bb1[2];
    const uint64_t *my = m + wpl, *mz = m + (size_t)VY * wpl;
    uint32_t rises = 0;
    uint64_t prev_msb = 0;
    for (int i = 0; i < wpl; ++i) {
      uint64_t w = m[i];
      ne = ne || w != 0;
      rises += (uint32_t)__popcll(w & ~((w << 1) | prev_msb));
      prev_msb = w >> 63;
      if (has_y && (w & my[i])) ey = true;
      if (has_z && (w & mz[i])) ez = true;
    }
    complex_line = rises > 1;
  }
  uint64_t bne = __ballot(ne), bey = __ballot(ey), bez = __ballot(ez), bc = __ballot(complex_line);
  if (lane == 0) {
    size_t o = (size_t)z * wy + yw;
    NE[o] = bne;
    EY[o] = bey;
    EZ[o] = bez;
    if (bc) cnt->flood_complex = 1;
  }
  }
}

__device__ __forceinline__ uint64_t fill_up64(uint64_t g, uint64_t p) {  // bit y may be entered from y-1 iff p[y]
  g |= p & (g << 1);  p &= (p << 1);
  g |= p & (g << 2);  p &= (p << 2);
  g |= p & (g << 4);  p &= (p << 4);
  g |= p & (g << 8);  p &= (p << 8);
  g |= p & (g << 16); p &= (p << 16);
  g |= p & (g << 32);
  return g;
}
__device__ __forceinline__ uint64_t fill_down64(uint64_t g, uint64_t p) {  // bit y may be entered from y+1 iff p[y]
  g |= p & (g >> 1);  p &= (p >> 1);
  g |= p & (g >> 2);  p &= (p >> 2);
  g |= p & (g >> 4);  p &= (p >> 4);
  g |= p & (g >> 8);  p &= (p >> 8);
  g |= p & (g >> 16); p &= (p >> 16);
  g |= p & (g >> 32);
  return g;
}
// occluded (Kogge-Stone) fill of g through the set bits of p, both directions inside one word
__device__ __forceinline__ uint64_t fill64(uint64_t g, uint64_t p) {
  g &= p;
  return fill_up64(g, p) | fill_down64(g, p);
}

// Flood over lines in the (y,z) plane.  One workgroup; r (reached lines, bit y of word [z][yw]) lives in LDS.
__global__ __launch_bounds__(TPB) void k_flood2d(Dims d, const FrameArgs *__restrict__ fa, const uint64_t *__restrict__ M, int wpl,
                                                 int wy, const uint64_t *__restrict__ EY, const uint64_t *__restrict__ EZ,
                                                 uint64_t *__restrict__ R2D, Counters *cnt) {
  const Frame f = fa->f;  // a copy (uniform registers): stores of the kernel cannot alias it
  extern __shared__ uint64_t lds[];  // r, ey, ez, pp: [nz][nyw] each
  __shared__ uint32_t changed;
  const int VY = d.NY + 1;
  const int yw0 = f.bb0[1] >> 6, yw1 = f.bb1[1] >> 6;
  const int nyw = yw1 - yw0 + 1, nz = f.bb1[2] - f.bb0[2] + 1, z0 = f.bb0[2];
  uint64_t *r = lds, *ey = lds + nz * nyw, *ez = lds + 2 * nz * nyw, *pp = lds + 3 * nz * nyw;
  for (int i = threadIdx.x; i < nz * nyw; i += blockDim.x) {
    r[i] = 0;
    const size_t g = (size_t)(z0 + i / nyw) * wy + yw0 + i % nyw;
    ey[i] = EY[g];
    ez[i] = EZ[g];
  }
  __syncthreads();
  if (threadIdx.x == 0 && f.start_ok) {
    // seed: the start vertex is reached iff it lies inside the frustum (operations.h:1324-1340)
    const int sx = f.start_v[0], sy = f.start_v[1], sz = f.start_v[2];
    if (sy >= f.bb0[1] && sy <= f.bb1[1] && sz >= f.bb0[2] && sz <= f.bb1[2] && sx >= f.bb0[0] && sx <= f.bb1[0]) {
      uint64_t w = M[((size_t)sz * VY + sy) * wpl + (sx >> 6)];
      if ((w >> (sx & 63)) & 1ull) {
        r[(sz - z0) * nyw + ((sy >> 6) - yw0)] = 1ull << (sy & 63);
        cnt->start_in_frustum = 1;
      }
    }
  }
  __syncthreads();
  uint32_t rounds = 0;
  for (; rounds < 256; ++rounds) {
    if (threadIdx.x == 0) changed = 0;
    __syncthreads();
    // fill along y inside every z row; edge bit y joins lines y and y+1
    for (int row = threadIdx.x; row < nz; row += blockDim.x) {
      uint64_t *rr = r + row * nyw;
      const uint64_t *e = ey + row * nyw;
      uint64_t any = 0;
      for (int i = 0; i < nyw; ++i) any |= rr[i];
      if (!any) continue;
      bool ch = false;
      uint64_t carry = 0;
      for (int i = 0; i < nyw; ++i) {  // upward
        uint64_t ew = e[i];
        uint64_t g = fill_up64(rr[i] | carry, ew << 1);
        carry = (g >> 63) & (ew >> 63);
        if (g != rr[i]) { rr[i] = g; ch = true; }
      }
      carry = 0;
      for (int i = nyw - 1; i >= 0; --i) {  // downward
        uint64_t ew = e[i];
        uint64_t g = fill_down64(rr[i] | (carry << 63), ew);
        carry = (i > 0) ? ((g & 1ull) & (e[i - 1] >> 63)) : 0ull;
        if (g != rr[i]) { rr[i] = g; ch = true; }
      }
      if (ch) changed = 1;
    }
    __syncthreads();
    // propagation along z as a parallel prefix (generate = r, propagate = edge to the neighbour row), log2(nz) steps
    // in each direction; every thread owns the (row, word) entries i = tid, tid + 256, ...
    for (int dir = 0; dir < 2; ++dir) {
      for (int i = threadIdx.x; i < nz * nyw; i += blockDim.x) {
        const int z = i / nyw;
        // up: row z is entered from z-1 through edge row z-1; down: from z+1 through edge row z
        pp[i] = dir == 0 ? (z > 0 ? ez[i - nyw] : 0ull) : (z < nz - 1 ? ez[i] : 0ull);
      }
      __syncthreads();
      for (int dist = 1; dist < nz; dist <<= 1) {
        uint64_t gn[3], pn[3];  // neighbour values of this thread's (at most 3) entries; nz*nyw <= 513*9 > 3*256 -> loop
        for (int base = 0; base < nz * nyw; base += 3 * (int)blockDim.x) {
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            const int i = base + q * (int)blockDim.x + (int)threadIdx.x;
            gn[q] = 0;
            pn[q] = 0;
            if (i < nz * nyw) {
              const int z = i / nyw;
              const int zn = dir == 0 ? z - dist : z + dist;
              if (zn >= 0 && zn < nz) {
                const int in = i + (zn - z) * nyw;
                gn[q] = r[in];
                pn[q] = pp[in];
              }
            }
          }
          __syncthreads();
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            const int i = base + q * (int)blockDim.x + (int)threadIdx.x;
            if (i < nz * nyw) {
              const uint64_t p = pp[i], old = r[i];
              const uint64_t g = old | (p & gn[q]);
              if (g != old) {
                r[i] = g;
                changed = 1;
              }
              pp[i] = p & pn[q];
            }
          }
          __syncthreads();
        }
      }
    }
    __syncthreads();
    if (!changed) break;
    __syncthreads();
  }
  for (int i = threadIdx.x; i < nz * nyw; i += blockDim.x) R2D[(size_t)(z0 + i / nyw) * wy + yw0 + i % nyw] = r[i];
  if (threadIdx.x == 0) cnt->flood_rounds = rounds;
}

// Exact 3-D bit flood for arbitrary masks (fallback, one workgroup): line fills along x, carry sweeps along y and z,
// repeated until a whole round changes nothing.
__global__ __launch_bounds__(1024) void k_flood_generic(Dims d, const FrameArgs *__restrict__ fa, const uint64_t *__restrict__ M,
                                                        uint64_t *__restrict__ R, int wpl, Counters *cnt) {
  const Frame f = fa->f;  // a copy (uniform registers): stores of the kernel cannot alias it
  if (!fa->force_generic && !cnt->flood_complex) return;
  __shared__ uint32_t changed;
  const int VY = d.NY + 1;
  const int y0 = f.bb0[1], z0 = f.bb0[2];
  const int ny = f.bb1[1] - y0 + 1, nz = f.bb1[2] - z0 + 1;
  const int xw0 = f.bb0[0] >> 6, nxw = (f.bb1[0] >> 6) - xw0 + 1;
  for (int t = threadIdx.x; t < ny * nz * wpl; t += blockDim.x) {
    int l = t / wpl;
    R[((size_t)(z0 + l / ny) * VY + (y0 + l % ny)) * wpl + t % wpl] = 0ull;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    cnt->start_in_frustum = 0;
    const int sx = f.start_v[0], sy = f.start_v[1], sz = f.start_v[2];
    if (f.start_ok && sy >= y0 && sy <= f.bb1[1] && sz >= z0 && sz <= f.bb1[2] && sx >= f.bb0[0] && sx <= f.bb1[0]) {
      size_t word = ((size_t)sz * VY + sy) * wpl + (sx >> 6);
      uint64_t bit = 1ull << (sx & 63);
      if (M[word] & bit) {
        R[word] = bit;
        cnt->start_in_frustum = 1;
      }
    }
  }
  __syncthreads();
  uint32_t rounds = 0;
  for (; rounds < 1024; ++rounds) {
    if (threadIdx.x == 0) changed = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < ny * nz; t += blockDim.x) {  // x fills
      size_t base = ((size_t)(z0 + t / ny) * VY + (y0 + t % ny)) * wpl;
      uint64_t r[MAX_WPL], m[MAX_WPL];
      uint64_t any = 0;
      for (int i = 0; i < wpl; ++i) { r[i] = R[base + i]; any |= r[i]; }
      if (!any) continue;
      for (int i = 0; i < wpl; ++i) m[i] = M[base + i];
      bool ch = false;
      uint64_t carry = 0;
      for (int i = 0; i < wpl; ++i) {
        uint64_t g = fill64(r[i] | (carry ? 1ull : 0ull), m[i]);
        carry = g >> 63;
        if (g != r[i]) ch = true;
        r[i] = g;
      }
      carry = 0;
      for (int i = wpl - 1; i >= 0; --i) {
        uint64_t g = fill64(r[i] | (carry ? (1ull << 63) : 0ull), m[i]);
        carry = g & 1ull;
        if (g != r[i]) ch = true;
        r[i] = g;
      }
      if (ch) {
        for (int i = 0; i < wpl; ++i) R[base + i] = r[i];
        changed = 1;
      }
    }
    __syncthreads();
    for (int axis = 1; axis <= 2; ++axis) {  // carry sweeps along y, then z
      const int no = axis == 1 ? nz : ny, na = axis == 1 ? ny : nz;
      const int o0 = axis == 1 ? z0 : y0, a0 = axis == 1 ? y0 : z0;
      const size_t stride = axis == 1 ? (size_t)wpl : (size_t)VY * wpl;
      for (int t = threadIdx.x; t < nxw * no; t += blockDim.x) {
        const int xw = xw0 + t % nxw, o = o0 + t / nxw;
        const size_t base = (axis == 1 ? (size_t)o * VY * wpl : (size_t)o * wpl) + xw;
        bool ch = false;
        uint64_t carry = 0;
        for (int a = a0; a < a0 + na; ++a) {
          size_t idx = base + (size_t)a * stride;
          uint64_t rr = R[idx], nr = rr | (carry & M[idx]);
          if (nr != rr) { R[idx] = nr; ch = true; }
          carry = nr;
        }
        carry = 0;
        for (int a = a0 + na - 1; a >= a0; --a) {
          size_t idx = base + (size_t)a * stride;
          uint64_t rr = R[idx], nr = rr | (carry & M[idx]);
          if (nr != rr) { R[idx] = nr; ch = true; }
          carry = nr;
        }
        if (ch) changed = 1;
      }
      __syncthreads();
    }
    if (!changed) break;
    __syncthreads();
  }
  if (threadIdx.x == 0) cnt->flood_rounds = rounds;
}

__device__ __forceinline__ bool vbit(const uint64_t *__restrict__ R, size_t line_base, int x) {
  return (R[line_base + (x >> 6)] >> (x & 63)) & 1ull;
}

// Per-voxel body of getIdxOfVisibleParitlces (operations.h:1344-1436): every voxel that shares a reached
// in-frustum vertex is handled exactly once (order does not matter: all effects are voxel-local except the
// per-pixel bins, whose order is made canonical afterwards).
// Dependent memory steps are kept few: (1) status row, stamp row, slab stamps and the depth under the voxel's "imaginary
// particle"; (2) positions of all live slots; (3) the depth pixel under each; (4) all bin-counter atomics and the row
// list reservations.  (Nine reached voxels in ten are empty and never get here: k_visibility's phase 1.)
template <int S>
__device__ __forceinline__ void visibility_voxel(const Dims &d, const Frame &f, const State &st, const Scratch &sc,
                                                 gptr<float> depth_img, int ax, int ay, int az) {
  const uint32_t rx = axis_correct(ax + f.eq[0], d.NX);
  const uint32_t ry = axis_correct(ay + f.eq[1], d.NY);
  const uint32_t rz = axis_correct(az + f.eq[2], d.NZ);
  const uint32_t shard = blockIdx.x & (VIS_SHARDS - 1);
  const uint32_t v = ring_to_voxel(d, rx, ry, rz);
  const uint32_t lv = v - d.v_begin;
  const size_t base = (size_t)lv * S;
  // (the caller has seen the voxel's flag byte: it holds something - the empty ones are finished in k_visibility's phase 1)
  // imaginary particle at the voxel's min corner, mapXYZIdxToGlobalPose (operations.h:986-991, 1418-1431); its depth is
  // requested together with the record's rows and only looked at when no particle of the voxel was observed
  float im_depth = 0.f, im_z = 0.f;
  bool im_ok;
  {
    float ix = (float)(uint32_t)ax * d.voxel_size + d.pmin[0] + f.center[0];
    float iy = (float)(uint32_t)ay * d.voxel_size + d.pmin[1] + f.center[1];
    float iz = (float)(uint32_t)az * d.voxel_size + d.pmin[2] + f.center[2];
    int row, col;
    im_ok = project_to_image(d, f, ix, iy, iz, row, col, im_z);
    if (im_ok) im_depth = depth_img[(size_t)row * d.W + col];
  }
  const uint32_t smax = stamp_max(st, rx, ry, rz);
  uint8_t stv[S];
  uint16_t tsv[S];
  unsigned char *const rec = rec_ptr(st, S, lv);
  rec_load_st_ts<S>(rec, stv, tsv);
  // the positions of all slots ride with the record's rows - the voxel's S positions are one 16*S-byte block (one line
  // at S = 8) whatever is live in it; requested slot by slot where a slot turned out live, each load sat in a branch of
  // its own and was waited for before the next one was requested (S - 1 dependent round trips)
  float4 pos[S];
#pragma unroll
  for (int i = 1; i < S; ++i) pos[i] = st.pos4[base + i];
  __builtin_amdgcn_sched_barrier(0);
  bool dirty = false, observed = false, wrote_free = false;
  int valid_n = 0;
  bool live[S];
#pragma unroll
  for (int i = 1; i < S; ++i) {
    live[i] = false;
    if (stv[i] == ST_INVALID) continue;
    if ((uint32_t)tsv[i] < smax) {  // outdated: delete (operations.h:1374-1378)
      stv[i] = ST_INVALID;
      dirty = true;
      continue;
    }
    valid_n++;
    live[i] = true;
  }
  int pixv[S], rowv[S];
  float camz[S], dptv[S];
#pragma unroll
  for (int i = 1; i < S; ++i) {
    pixv[i] = -1;
    rowv[i] = 0;
    if (!live[i]) continue;
    int row, col;
    if (project_to_image(d, f, pos[i].x, pos[i].y, pos[i].z, row, col, camz[i])) {
      pixv[i] = row * d.W + col;
      rowv[i] = row;
      dptv[i] = depth_img[pixv[i]];
    }
  }
  bool vis[S];
  uint32_t pib[S];
#pragma unroll
  for (int i = 1; i < S; ++i) {
    vis[i] = false;
    if (!live[i] || pixv[i] < 0) continue;
    const float dpt = dptv[i];
    if (dpt > d.dmax) {  // nothing measurable along this ray: free (operations.h:1389-1395)
      SlotRef{rec, S - 1, (uint32_t)i - 1u}.set_w(SDM_OCC_INIT_WEIGHT);
      wrote_free = true;
      observed = true;
      continue;
    }
    if (camz[i] > dpt * d.occl_coeff) continue;  // occluded (operations.h:1397-1400)
    observed = true;
    vis[i] = true;
  }
  // A visible particle takes its place in its pixel's bin (pib, counted per pixel) and goes on the list of its IMAGE ROW
  // (one of ROW_SUBS sub-lists per row, so that the counters are spread over many cache lines): the
  // workgroup of k_bin_rows that lays out the row's bins finds the row's particles there.
  // (the sub-list by workgroup AND lane: picked by workgroup alone - round 4 - the particles of a crowded row that a few
  // workgroups see went to a few of its sub-lists, and one full sub-list voids the frame although the row and the map had
  // room; spread over all eight, a row holds ROW_SUBS * row_cap particles whoever finds them)
  const uint32_t sub = (shard + (threadIdx.x & 63u)) & (ROW_SUBS - 1);
  uint32_t rq[S];
#pragma unroll
  for (int i = 1; i < S; ++i)
    if (vis[i]) {
      pib[i] = atomicAdd(&sc.bin_count[pixv[i]], 1u);
      rq[i] = atomicAdd(&sc.row_cnt[(size_t)(rowv[i] * ROW_SUBS + sub) * ROW_CNT_STRIDE], 1u);
    }
#pragma unroll
  for (int i = 1; i < S; ++i)
    if (vis[i]) {
      if (rq[i] < sc.row_cap && pib[i] < (1u << ROW_PIB_BITS)) {
        const uint32_t col = (uint32_t)(pixv[i] - rowv[i] * d.W);
        sc.row_list[(size_t)(rowv[i] * ROW_SUBS + sub) * sc.row_cap + rq[i]] =
            make_uint2((uint32_t)(((size_t)v << d.p_n) + i), col | pib[i] << ROW_COL_BITS);
      } else {
        sc.cnt->overflow = 1;
      }
    }
  if (dirty) rec_store_status<S>(rec, stv);
  if (dirty || wrote_free) st.vflag[lv] = VF_DIRTY;
  bool stamped = observed;
  if (!observed && valid_n == 0) stamped = im_ok && im_z <= im_depth;
  if (stamped) st.vts[lv] = (uint16_t)f.gts;
  if (dirty || wrote_free || stamped) mark_tile(st, lv, f.epoch);
}

// Three phases per workgroup round.  Only about a third of the voxels of the frustum's index box were reached, and testing
// them one by one costs a dozen bit-test loads each.  So the candidates are taken 64 at a time: a "word" is the 64
// voxels along x whose lower corner vertices share one 64-bit word of the vertex bitmaps.  A voxel is handled iff one
// of its 8 corner vertices was reached by the flood: per vertex line (4 per voxel row) that is word | word >> 1 (the
// upper-x corner; bit 0 of the next word shifts in).  Simple masks: vertex reached = in-frustum bit & its x-line
// reached; complex masks: the generic flood wrote the reached bits to sc.reach.  VIS_WORDS words per workgroup round;
// their set bits are then dealt out to the lanes.
// Nine candidates in ten are empty voxels, which need two dependent loads (flag byte, depth under the corner); the
// kernel is a chain of such dependent loads, so every thread runs its (up to VIS_J) candidates side by side: all flag
// bytes are requested, then all depths, then the stamps go out.  Voxels that hold something are listed in LDS and
// handled afterwards by full waves (round 2 ran one candidate per thread at a time, empty or not: 39 us, waiting).
constexpr int VIS_WORDS = 32;
constexpr int VIS_J = VIS_WORDS * 64 / TPB;  // candidates per thread and round, at most

template <int S>
__global__ __launch_bounds__(TPB) void k_visibility(Dims d, State st, Scratch sc) {
  __shared__ unsigned long long wmask[VIS_WORDS];
  __shared__ uint32_t woff[VIS_WORDS + 1];
  __shared__ uint16_t full_list[VIS_WORDS * 64];  // candidates that hold something: word << 6 | bit
  __shared__ uint32_t n_full;
  DBG_LANE0(1, 0);
  const Frame f = sc.fa->f;  // a copy (uniform registers): stores of the kernel cannot alias it
  const gptr<float> depth_img = as_global(sc.fa->depth);
  // (requested with the frame's scalars: one dependent step less before the masks)
  const uint32_t fgen = (uint32_t)sc.fa->force_generic, fcx = (uint32_t)sc.cnt->flood_complex;
  const bool generic = (fgen | fcx) != 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // (this kernel is the flood's last consumer: Counters::vis_*)
    sc.cnt->vis_flood_complex = fcx;
    sc.cnt->vis_flood_rounds = sc.cnt->flood_rounds;
    sc.cnt->vis_start_in_frustum = sc.cnt->start_in_frustum;
  }
  const int bx = f.bb1[0] - f.bb0[0], by = f.bb1[1] - f.bb0[1], bz = f.bb1[2] - f.bb0[2];  // voxel box [bb0,bb1)
  if (bx <= 0 || by <= 0 || bz <= 0) return;
  const int wlo = f.bb0[0] >> 6, nwx = ((f.bb1[0] - 1) >> 6) - wlo + 1;
  const uint32_t n_words = (uint32_t)nwx * by * bz;
  // the grid does not depend on the frame (the box does): workgroups stride over the box's words
  for (uint32_t g0 = blockIdx.x * VIS_WORDS; g0 < n_words; g0 += gridDim.x * VIS_WORDS) {
  __syncthreads();  // the previous round's lists have been read
  if (threadIdx.x == 0) n_full = 0;
  if (threadIdx.x < VIS_WORDS) {
    const uint32_t g = g0 + threadIdx.x;
    unsigned long long m = 0;
    if (g < n_words) {
      const int wi = wlo + (int)(g % nwx);
      const int ay = f.bb0[1] + (int)((g / nwx) % by);
      const int az = f.bb0[2] + (int)(g / ((uint32_t)nwx * by));
      const uint32_t rz = axis_correct(az + f.eq[2], d.NZ);
      if (rz >= d.rz_begin && rz < d.rz_begin + d.rz_count) {  // else: another shard's slab
        const uint64_t *__restrict__ bits = generic ? sc.reach : sc.vmask;
        const int VY = d.NY + 1;
        // the twelve words of the four lines are requested before any of them is looked at, none under a condition of
        // its own (a load in a branch of its own is waited for before the next one is requested: this was a chain of
        // eight dependent round trips at the start of every workgroup)
        unsigned long long w0[4], w1[4], lr[4];
        const int wi1 = wi + 1 < (int)sc.wpl ? wi + 1 : wi;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int y = ay + (c & 1), z = az + (c >> 1);
          const size_t lb = ((size_t)z * VY + y) * sc.wpl;
          lr[c] = sc.line_reach[(size_t)z * sc.wy + (y >> 6)];  // (not looked at in generic mode)
          w0[c] = bits[lb + wi];
          w1[c] = bits[lb + wi1];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int y = ay + (c & 1);
          const bool line_ok = generic || ((lr[c] >> (y & 63)) & 1ull);
          const unsigned long long w1m = wi + 1 < (int)sc.wpl ? w1[c] : 0ull;
          if (line_ok) m |= w0[c] | (w0[c] >> 1) | (w1m << 63);
        }
        // voxels of the box only
        const int x0 = wi << 6;
        const int lo = f.bb0[0] > x0 ? f.bb0[0] - x0 : 0;
        const int hi = f.bb1[0] - x0 < 64 ? f.bb1[0] - x0 : 64;  // exclusive
        unsigned long long keep = hi >= 64 ? ~0ull : ((1ull << hi) - 1ull);
        keep &= ~((1ull << lo) - 1ull);
        m &= keep;
      }
    }
    wmask[threadIdx.x] = m;
    // exclusive prefix of the words' popcounts by the wave that holds them
    const uint32_t c = (uint32_t)__popcll(m);
    uint32_t inc = c;
#pragma unroll
    for (int off = 1; off < VIS_WORDS; off <<= 1) {
      const uint32_t t = __shfl_up(inc, off, 64);
      if ((int)threadIdx.x >= off) inc += t;
    }
    woff[threadIdx.x] = inc - c;
    if (threadIdx.x == VIS_WORDS - 1) woff[VIS_WORDS] = inc;
  }
  __syncthreads();
  const uint32_t nl = woff[VIS_WORDS];
  DBG_LANE0(1, 1);
  // voxels handled (statistics): one add per workgroup - an atomic per voxel on these 64 addresses cost 26 us
  if (threadIdx.x == 0 && nl) atomicAdd(&sc.cnt->shard[blockIdx.x & (VIS_SHARDS - 1)].fv, nl);
  if (nl == 0) continue;  // workgroup-uniform
  // ---- phase 1: every candidate's flag byte; empty voxels finished, the others listed
  uint32_t code[VIS_J], lvv[VIS_J], flg[VIS_J];
#pragma unroll
  for (int j = 0; j < VIS_J; ++j) {
    const uint32_t li = threadIdx.x + (uint32_t)j * TPB;
    // (no branch around this: the flag byte of every candidate of the thread is requested before the first one is
    // looked at - under `if (li < nl)` each load was waited for where its branch ended, VIS_J dependent round trips)
    code[j] = 0xffffffffu;
    lvv[j] = 0;
    flg[j] = 0;
    if ((uint32_t)j * TPB >= nl) continue;  // workgroup-uniform: nobody has a j-th candidate
    const bool have = li < nl;
    int w = 0;
#pragma unroll
    for (int k = 1; k < VIS_WORDS; ++k) w += woff[k] <= li ? 1 : 0;
    const int bit = nth_set_bit(wmask[w], li - woff[w]) & 63;
    const uint32_t g = g0 + (uint32_t)w;
    const int ax = ((wlo + (int)(g % nwx)) << 6) + bit;
    const int ay = f.bb0[1] + (int)((g / nwx) % by);
    const int az = f.bb0[2] + (int)(g / ((uint32_t)nwx * by));
    const uint32_t rx = axis_correct(ax + f.eq[0], d.NX), ry = axis_correct(ay + f.eq[1], d.NY), rz = axis_correct(az + f.eq[2], d.NZ);
    code[j] = have ? (uint32_t)(w << 6 | bit) : 0xffffffffu;
    lvv[j] = have ? ring_to_voxel(d, rx, ry, rz) - d.v_begin : 0u;
    flg[j] = st.vflag[lvv[j]];  // VF_EMPTY: every slot INVALID, the record is not touched
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < VIS_J; ++j) flg[j] = code[j] != 0xffffffffu ? flg[j] & VF_STATE : 0u;
  float imz[VIS_J], imd[VIS_J];
#pragma unroll
  for (int j = 0; j < VIS_J; ++j) {
    imz[j] = 1.f;
    imd[j] = 0.f;  // "not seen"
    if (code[j] == 0xffffffffu) continue;
    if (flg[j]) {
      full_list[atomicAdd(&n_full, 1u)] = (uint16_t)code[j];
      continue;
    }
    // imaginary particle at the voxel's min corner, mapXYZIdxToGlobalPose (operations.h:986-991, 1418-1431)
    const uint32_t g = g0 + (code[j] >> 6);
    const int ax = ((wlo + (int)(g % nwx)) << 6) + (int)(code[j] & 63u);
    const int ay = f.bb0[1] + (int)((g / nwx) % by);
    const int az = f.bb0[2] + (int)(g / ((uint32_t)nwx * by));
    const float ix = (float)(uint32_t)ax * d.voxel_size + d.pmin[0] + f.center[0];
    const float iy = (float)(uint32_t)ay * d.voxel_size + d.pmin[1] + f.center[1];
    const float iz = (float)(uint32_t)az * d.voxel_size + d.pmin[2] + f.center[2];
    int row, col;
    float z;
    if (project_to_image(d, f, ix, iy, iz, row, col, z)) {
      imz[j] = z;
      imd[j] = depth_img[(size_t)row * d.W + col];
    }
  }
#pragma unroll
  for (int j = 0; j < VIS_J; ++j) {
    if (code[j] == 0xffffffffu || flg[j]) continue;
    if (imz[j] <= imd[j]) {  // (imd = 0 where the corner does not project into the image: z >= depth_min > 0)
      st.vts[lvv[j]] = (uint16_t)f.gts;
      mark_tile(st, lvv[j], f.epoch);
    }
  }
  __syncthreads();
  DBG_LANE0(1, 2);
  // ---- phase 2: the voxels that hold something, full waves
  const uint32_t nf = n_full;
  for (uint32_t k = threadIdx.x; k < nf; k += TPB) {
    const uint32_t c = full_list[k];
    const uint32_t g = g0 + (c >> 6);
    const int ax = ((wlo + (int)(g % nwx)) << 6) + (int)(c & 63u);
    const int ay = f.bb0[1] + (int)((g / nwx) % by);
    const int az = f.bb0[2] + (int)(g / ((uint32_t)nwx * by));
    visibility_voxel<S>(d, f, st, sc, depth_img, ax, ay, az);
  }
  DBG_LANE0(1, 3);
  }
}

__device__ __forceinline__ void sift_down(uint32_t *a, uint32_t start, uint32_t end) {
  uint32_t root = start;
  while (2 * root + 1 <= end) {
    uint32_t child = 2 * root + 1, sw = root;
    if (a[sw] < a[child]) sw = child;
    if (child + 1 <= end && a[sw] < a[child + 1]) sw = child + 1;
    if (sw == root) return;
    uint32_t t = a[root];
    a[root] = a[sw];
    a[sw] = t;
    root = sw;
  }
}

// pass 1 of the weight update (A7 below) splits the pixels by the number of particles their window holds
constexpr int A7_ROWS = 16;   // >= 2*window_half+1
constexpr int A7_ITEMS = 16;  // pixels / particles per workgroup
constexpr uint32_t CK_LIGHT_MAX = 4;
enum : uint8_t { CK_DONE = 0, CK_LIGHT = 1, CK_HEAVY = 2 };

__device__ __forceinline__ void ck_store(const Filter &flt, const Scratch &sc, float *__restrict__ ck_out, int finish, int p,
                                         const sdm_labeled_point &o, float ck) {
  if (finish) {
    const float ckk = ck * flt.p_detect + flt.noise_number;
    sc.ck_kappa[p] = ckk;
    sc.pix4[p] = make_float4(o.x, o.y, o.z, ckk);
    sc.pixt[p] = (uint32_t)o.track_id | (1u << 16);
  } else {
    ck_out[p] = ck;
  }
}

// The per-pixel bins of the visible particles (buffer.h:90-93, operations.h:1405-1407), one workgroup per IMAGE ROW.
// A window row of the weight update is a stretch of consecutive pixels of one image row, so all it needs is that the bins
// of a row lie back to back in pixel order - where a row's block starts is free.  The workgroup scans its row's per-pixel
// counts, reserves the row's block with ONE atomic on the frame's particle counter (which ends up as n_vis), writes the
// row's bin offsets (W + 1 per row: the last one is the row's end), drops the row's particles - k_visibility listed them
// per row - into their bins, and then every pixel's thread brings its bin into the canonical order (ascending particle
// index; the reference's push order is its BFS order) and gathers the fields the weight update reads into
// arrays laid out in bin order.  (Round 3: a device-wide scan of the 466 K counts, a scatter kernel, a sort-and-gather
// kernel - three launches, 35 us with their gaps, for what one row-local launch does.)
// Ten waves per workgroup and at most 96 registers: TWO workgroups fit a CU, so the 375 rows of the benchmark image are all
// resident at once (with 1024 threads and 128 registers it was one per CU: 256 at a time, the other 119 rows started when
// the first ones were through - the launch took two rounds, 17 us for 7 us of work per row).  PPT consecutive pixels per
// thread: 2 up to 1280 columns, 4 beyond.
constexpr int BR_TPB = 640;
constexpr int BR_WAVES = BR_TPB / 64;
constexpr int BR_MAXW = 1 << ROW_COL_BITS;  // image width the row kernel's LDS holds and a row-list entry's column field takes (checked when the map is created)
template <int PPT>
__global__ __launch_bounds__(BR_TPB) __attribute__((amdgpu_waves_per_eu(PPT <= 4 ? 5 : 4, PPT <= 4 ? 5 : 4))) void k_bin_rows(Dims d, State st, Scratch sc) {
  static_assert(PPT == 2 || PPT == 4 || PPT * BR_TPB >= BR_MAXW, "eight pixels per thread cover the widest image");
  __shared__ uint32_t pre[BR_MAXW + 1];
  __shared__ uint32_t wave_tot[BR_WAVES];
  __shared__ uint32_t s_base;
  const int r = blockIdx.x;
  DBG_LANE0(2, 0);
  // (the one place of a frame where nobody reads or writes the table of older set memberships: its deleted entries go)
  if (r == (int)gridDim.x - 1 && threadIdx.x < 64) alias_compact_wave(st);
  const bool overflow = sc.cnt->overflow != 0;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int W = d.W;
  // ---- the row's counts, PPT pixels per thread (consecutive: thread t holds columns PPT t ...), exclusive scan
  uint32_t c[PPT];
  const int j0 = PPT * (int)threadIdx.x;
#pragma unroll
  for (int u = 0; u < PPT; ++u) c[u] = (!overflow && j0 + u < W) ? sc.bin_count[r * W + j0 + u] : 0u;
  // the row's sub-lists (counts requested with the bins' counts)
  uint32_t sub_n = 0;
  if (threadIdx.x < ROW_SUBS && !overflow) {
    sub_n = sc.row_cnt[(size_t)(r * ROW_SUBS + threadIdx.x) * ROW_CNT_STRIDE];
    if (sub_n > sc.row_cap) sub_n = sc.row_cap;
  }
  uint32_t inc = 0;
#pragma unroll
  for (int u = 0; u < PPT; ++u) inc += c[u];
  const uint32_t mine = inc;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) wave_tot[wid] = inc;
  __syncthreads();
  uint32_t before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < BR_WAVES; ++w) {
    if (w < wid) before += wave_tot[w];
    total += wave_tot[w];
  }
  {
    uint32_t run = before + inc - mine;
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      if (j0 + u < W) pre[j0 + u] = run;
      run += c[u];
    }
  }
  if (threadIdx.x == 0) {
    pre[W] = total;
    s_base = total ? atomicAdd(&sc.cnt->n_vis, total) : 0u;
    if (total && (s_base > sc.cap_vis || total > sc.cap_vis - s_base)) {  // the bin-order arrays hold cap_vis particles
      sc.cnt->overflow = 1;
      pre[W] = 0;  // (this row is left out; the frame's results are void anyway: SDM_ERR_CAPACITY)
    }
  }
  // exclusive prefix of the sub-list lengths (wave 0 holds them in its first lanes)
  uint32_t sub_ex[ROW_SUBS + 1];
  {
    uint32_t run = 0;
#pragma unroll
    for (int k = 0; k < ROW_SUBS; ++k) {
      sub_ex[k] = run;
      run += (uint32_t)__shfl(sub_n, k, 64);
    }
    sub_ex[ROW_SUBS] = run;
  }
  __shared__ uint32_t s_sub[ROW_SUBS + 1];
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k <= ROW_SUBS; ++k) s_sub[k] = sub_ex[k];
  }
  __syncthreads();
  const uint32_t base = s_base;
  // ---- the row's bin offsets
  uint32_t *__restrict__ bs = sc.bin_start + (size_t)r * (W + 1);
  const bool row_over = total != 0 && pre[W] == 0;
  // how many particles does this row put into the window of the pixel in column j of ANY image row near it (columns
  // j - h .. j + h)?  Saturated to a byte: k_ck_classify only asks whether a window holds none, up to four or more
  {
    const int h = d.window_half;
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      const int j = j0 + u;
      if (j < W) {
        const uint32_t n = row_over ? 0u : pre[j + h + 1 > W ? W : j + h + 1] - pre[j - h < 0 ? 0 : j - h];
        sc.row_win[r * W + j] = (uint8_t)(n > 255u ? 255u : n);
      }
    }
  }
  for (int j = threadIdx.x; j <= W; j += BR_TPB) bs[j] = row_over ? 0u : base + pre[j];
  if (total == 0 || row_over) {
    DBG_LANE0(2, 1);
    return;  // (workgroup-uniform)
  }
  // ---- the row's particles into their bins
  const uint32_t n_row = s_sub[ROW_SUBS];
  for (uint32_t g = threadIdx.x; g < n_row; g += BR_TPB) {
    int sub = 0;
#pragma unroll
    for (int k = 1; k < ROW_SUBS; ++k) sub += s_sub[k] <= g ? 1 : 0;
    const uint2 e = sc.row_list[(size_t)(r * ROW_SUBS + sub) * sc.row_cap + (g - s_sub[sub])];
    sc.bin_idx[base + pre[e.y & (uint32_t)(BR_MAXW - 1)] + (e.y >> ROW_COL_BITS)] = e.x;
  }
  __syncthreads();  // (the bins were written by this workgroup: its own stores are visible to it behind the barrier)
  // ---- every pixel's bin: canonical order, gather
  const size_t slot_base = (size_t)d.v_begin << d.p_n;
#pragma unroll
  for (int u = 0; u < PPT; ++u) {
    const uint32_t n = c[u];
    if (!n) continue;
    const uint32_t p = (uint32_t)(r * W + j0 + u);
    const uint32_t s = base + pre[j0 + u];
    uint32_t *a = sc.bin_idx + s;
    // the usual bin: its entries in registers (one round trip for all of them), ordered by a sorting network, then the
    // particles' fields - again all requested before the first is used.  (Sorted in place in memory and gathered entry by
    // entry, a bin of ten was a chain of sixty dependent memory accesses: the tail this kernel used to end with.)
    auto small_bin = [&](auto kk) {
      constexpr int K = decltype(kk)::value;
      uint32_t v[K];
#pragma unroll
      for (int i = 0; i < K; ++i) v[i] = (uint32_t)i < n ? a[i] : 0xffffffffu;
#pragma unroll
      for (int pass = 0; pass < K; ++pass)
#pragma unroll
        for (int i = pass & 1; i + 1 < K; i += 2) {  // odd-even transposition: K passes order K keys
          const uint32_t lo = v[i] < v[i + 1] ? v[i] : v[i + 1], hi = v[i] < v[i + 1] ? v[i + 1] : v[i];
          v[i] = lo;
          v[i + 1] = hi;
        }
      // (eight entries per round of loads: sixteen positions at once are 64 registers)
#pragma unroll
      for (int i0 = 0; i0 < K; i0 += 8) {
        if ((uint32_t)i0 >= n) break;
        float4 q[8];
        float w8[8];
        uint16_t t8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if ((uint32_t)(i0 + i) < n) {
            const size_t li = (size_t)v[i0 + i] - slot_base;
            q[i] = st.pos4[li];
            q[i].w = __uint_as_float((uint32_t)st.forget[li]);
            const SlotRef sr = slot_ref_li(st, d.p_n, li);
            w8[i] = sr.w();
            t8[i] = sr.track();
          }
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if ((uint32_t)(i0 + i) < n) {
            if (n > 1) a[i0 + i] = v[i0 + i];
            sc.vp4[s + i0 + i] = make_float4(q[i].x, q[i].y, q[i].z, w8[i]);
            sc.vtf[s + i0 + i] = (uint32_t)t8[i] | ((__float_as_uint(q[i].w) & 0xffu) << 16);
            sc.vpix[s + i0 + i] = p;
          }
      }
    };
    if (n <= 8) {
      small_bin(std::integral_constant<int, 8>{});
      continue;
    }
    if (n <= 16) {
      small_bin(std::integral_constant<int, 16>{});
      continue;
    }
    if (n > 1) {
      if (n <= 32) {
        for (uint32_t i = 1; i < n; ++i) {
          uint32_t x = a[i];
          uint32_t j = i;
          while (j > 0 && a[j - 1] > x) {
            a[j] = a[j - 1];
            --j;
          }
          a[j] = x;
        }
      } else {  // heap sort, in place
        for (int start = (int)(n - 2) / 2; start >= 0; --start) sift_down(a, (uint32_t)start, n - 1);
        for (uint32_t end = n - 1; end > 0; --end) {
          uint32_t t = a[end];
          a[end] = a[0];
          a[0] = t;
          sift_down(a, 0, end - 1);
        }
      }
    }
    for (uint32_t i = 0; i < n; ++i) {
      const size_t li = (size_t)a[i] - slot_base;
      const float4 q = st.pos4[li];
      const SlotRef sr = slot_ref_li(st, d.p_n, li);
      sc.vp4[s + i] = make_float4(q.x, q.y, q.z, sr.w());
      sc.vtf[s + i] = (uint32_t)sr.track() | ((uint32_t)st.forget[li] << 16);
      sc.vpix[s + i] = p;
    }
  }
  DBG_LANE0(2, 1);
}

// Pass 1 of the weight update (A7 below) splits the pixels by the number of particles their window holds: none (half of
// all windows): ck = 0, stored right away.  Up to CK_LIGHT_MAX: left to the one-thread-per-pixel part of k_ck.  More: onto
// the sharded list of the row-parallel part.  Both parts then run in ONE launch side by side.
// One thread per pixel; the window total is the sum of the per-row counts k_bin_rows left in row_win (2 h + 1 byte loads).
// A wave reserves the list places of its heavy pixels with ONE atomic: a quarter of all pixels are heavy, and an atomic per
// pixel on the 64 list counters - a cache line each, 12 ns per atomic and line - was 20 us of this step.
__global__ __launch_bounds__(TPB) void k_ck_classify(Dims d, Filter flt, Scratch sc, float *__restrict__ ck_out, int finish) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  const int W = d.W, H = d.H, h = d.window_half;
  const bool in_image = p < (uint32_t)(W * H);
  const bool overflow = sc.cnt->overflow != 0;
  sdm_labeled_point o;
  o.is_valid = 0;
  uint32_t total = 0;
  if (in_image) {
    o = load_point(sc.fa->cloud, p);
    const int i = (int)p / W;
    // the window's row counts, all requested before the first is added (a row outside the window or the image repeats
    // the pixel's own row: no branch per load - each one was waited for before the next was requested, 2h + 1 dependent
    // round trips in a kernel of 8 us)
    uint32_t rw[A7_ROWS];
#pragma unroll
    for (int r = 0; r < A7_ROWS; ++r) {
      const int ni = i + r - h;
      const bool ok = r <= 2 * h && ni >= 0 && ni < H;
      rw[r] = sc.row_win[(int)p + (ok ? (r - h) * W : 0)];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < A7_ROWS; ++r) {
      const int ni = i + r - h;
      total += (r <= 2 * h && ni >= 0 && ni < H) ? rw[r] : 0u;
    }
  }
  uint8_t cls = CK_DONE;
  if (in_image && o.is_valid && total != 0 && !overflow) cls = total <= CK_LIGHT_MAX ? CK_LIGHT : CK_HEAVY;
  const unsigned long long hm = __ballot(cls == CK_HEAVY);
  if (hm) {  // (wave-uniform)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t shard = blockIdx.x & (VIS_SHARDS - 1);
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&sc.cnt->shard[shard].heavy, (uint32_t)__popcll(hm));
    base = (uint32_t)__shfl((int)base, 0, 64);
    // cap_heavy covers every pixel a shard's blocks can hold
    if (cls == CK_HEAVY) sc.ck_heavy[shard * sc.cap_heavy + base + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull))] = p;
  }
  if (!in_image) return;
  if (!finish) {
    // Z-slab shards: pass 1 leaves partial sums that are exchanged and summed before pass 2.  The per-pixel operands of pass 2
    // that do not depend on the sums - position, track, validity - are written here, so that pass 2 can take the summed image
    // as it comes out of the exchange (k_weight's ck_raw) and no per-pixel launch has to sit between the exchange and it.
    sc.pix4[p] = make_float4(o.x, o.y, o.z, 0.f);
    sc.pixt[p] = o.is_valid ? ((uint32_t)o.track_id | (1u << 16)) : 0u;
  }
  if (!o.is_valid) {
    if (finish) sc.pixt[p] = 0;  // invalid pixel: skipped by pass 2
  } else if (cls == CK_DONE) {
    ck_store(flt, sc, ck_out, finish, (int)p, o, 0.f);
  }
  sc.ck_class[p] = cls;
}

// ------------------------------------------------------------------------------------ A7
// SemanticDSPMap::updateParticles (semantic_dsp_map.h:960-1121), the SMC-PHD weight update.  Not HBM-bound: every
// particle-pixel pair of a (2h+1)^2 window costs three IEEE divisions and three LUT reads.  The work is split per
// WINDOW ROW so that the long serial chains of the CPU loops become 2h+1 short ones: a 16x16 workgroup holds 16
// pixels (pass 1) or 16 particles (pass 2) times up to 16 window rows; every thread accumulates its row in the
// reference's order, then the row sums are added in row order.  (Canonical summation order, DESIGN.md: the value is
// identical on the oracle's canonical mode and differs from the reference's single running sum only in rounding.)
// pass 1 (semantic_dsp_map.h:973-1037): ck of every valid pixel.  finish != 0 also applies ck*P_d + kappa (:1035).
// Visible particles are far fewer than pixels (C3: ~27 K against 466 K) and clustered: half the windows are empty,
// the median window holds one particle, the 99th percentile 124.  So the pixels are split (by k_ck_classify, which
// also finishes the empty windows): the light part of k_ck (one thread per pixel) computes every pixel whose window holds
// at most CK_LIGHT_MAX particles, the heavy part spreads each listed pixel over its window rows.  Both add a row's terms
// in bin order and the row sums in row order (canonical order) - the value does not depend on which part
// produced it.  One launch: the first CK_HEAVY_BLOCKS workgroups take batches of the heavy list (ticketed), the others
// one block of 256 pixels each.

template <bool FAST>
__device__ __forceinline__ float ck_term(const Filter &flt, const float *__restrict__ pdf, const float4 pv, const uint32_t tf,
                                         const sdm_labeled_point &o, float rsig, bool &skip) {
  const uint16_t ptrack = (uint16_t)(tf & 0xffffu);
  skip = flt.independent && ptrack != o.track_id;
  // (the forgetting factor is a load too - a table in the kernel's argument block, indexed at run time - and is requested
  // with the three table values of the term: behind them, in the branch that uses it, it was a second dependent round
  // trip per term)
  const float fg = flt.forget[(tf >> 16) & 7];
  float gk = query_pdf_r<FAST>(pdf, pv.x, o.x, o.sigma, rsig) * query_pdf_r<FAST>(pdf, pv.y, o.y, o.sigma, rsig) *
             query_pdf_r<FAST>(pdf, pv.z, o.z, o.sigma, rsig);
  if (!flt.independent) {
    gk *= fg;
    if (ptrack != o.track_id) gk *= flt.id_transition;
  }
  return pv.w * gk;
}

__device__ __forceinline__ void ck_light_pixel(const Dims &d, const Filter &flt, const State &st, const Scratch &sc,
                                               float *__restrict__ ck_out, int finish, int p) {
  const bool mine = p < d.W * d.H && sc.ck_class[p] == CK_LIGHT;
  sdm_labeled_point o;
  uint32_t ss[A7_ROWS], ee[A7_ROWS];
  if (mine) {
    o = load_point(sc.fa->cloud, p);
    const int h = d.window_half;
    const int i = p / d.W, j = p - i * d.W;
    const int j0 = j - h < 0 ? 0 : j - h;
    const int j1 = j + h >= d.W ? d.W - 1 : j + h;
#pragma unroll
    for (int r = 0; r < A7_ROWS; ++r) {
      const int ni = i + r - h;
      ss[r] = ee[r] = 0;
      if (r <= 2 * h && ni >= 0 && ni < d.H) {
        ss[r] = sc.bin_start[ni * (d.W + 1) + j0];
        ee[r] = sc.bin_start[ni * (d.W + 1) + j1 + 1];
      }
    }
  }
  float ck = 0.f;
  const float *__restrict__ pdf = st.pdf;
  const float rsig = mine ? div_recip(o.sigma) : 1.f;
  auto rows = [&](auto fast) {
#pragma unroll
    for (int r = 0; r < A7_ROWS; ++r) {
      if (ss[r] == ee[r]) continue;  // an empty row adds +0
      float acc = 0.f;
      for (uint32_t k = ss[r]; k < ee[r]; ++k) {
        bool skip;
        const float t = ck_term<decltype(fast)::value>(flt, pdf, sc.vp4[k], sc.vtf[k], o, rsig, skip);
        if (!skip) acc += t;
      }
      ck += acc;
    }
  };
  if (__ballot(rsig == 0.f) == 0ull) {  // wave-uniform: every sigma of the wave inside the range div_by is verified for
    if (mine) rows(std::true_type{});
  } else {
    if (mine) rows(std::false_type{});
  }
  if (mine) ck_store(flt, sc, ck_out, finish, p, o, ck);
}

// heavy part: 16 listed pixels per workgroup round, shard of the list = blockIdx.x & 63.  Window sizes are very uneven (9 .. ~300
// particles), so the particle-pixel terms of the 16 pixels are flattened (pixel, row, bin order) and dealt out evenly
// to the 256 lanes - each computes a contiguous chunk of terms into LDS - and then lane (pixel, row) adds its row's
// terms in bin order; the row sums are added in row order.  More terms than the LDS buffer holds: several passes,
// the running row sums stay in registers.
// (Round 6: 3584 terms per pass instead of 4096 and the kernel held to 72 registers (it took 79) make it seven resident
// workgroups per CU instead of six, by LDS - 21.5 KB - and by registers: weight stage 57 -> 55 us on the benchmark frames,
// 77 -> 76 us on `driven`, in two A/B calls.  Eight (64 registers, 2048 terms) spills and is slower: 61 / 83 us.)
#ifndef SDM_CK_TERM_CAP
#define SDM_CK_TERM_CAP 3584
#endif
constexpr uint32_t CK_TERM_CAP = SDM_CK_TERM_CAP;

// (workgroups beyond what the chip holds at once only draw a ticket past the end of their list and leave: 28 per list
// - 1792, the resident number at this kernel's register and LDS use; round 4, at 24 resident per list: 37.4 us against
// 39.4 us with 64 per list)
#ifndef SDM_CK_HEAVY_PER_LIST
#define SDM_CK_HEAVY_PER_LIST 28
#endif
constexpr uint32_t CK_HEAVY_BLOCKS = SDM_CK_HEAVY_PER_LIST * VIS_SHARDS;
#ifndef SDM_CK_WAVES
#define SDM_CK_WAVES 7
#endif
#define CK_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(SDM_CK_WAVES, SDM_CK_WAVES)))

__global__ __launch_bounds__(A7_ROWS *A7_ITEMS) CK_WAVES_ATTR void k_ck(Dims d, Filter flt, State st, Scratch sc,
                                                          float *__restrict__ ck_out, int finish) {
  DBG_LANE0(3, 0);
  if (blockIdx.x >= CK_HEAVY_BLOCKS) {  // light part: one thread per pixel
    ck_light_pixel(d, flt, st, sc, ck_out, finish,
                   (int)((blockIdx.x - CK_HEAVY_BLOCKS) * (A7_ROWS * A7_ITEMS) + threadIdx.y * A7_ROWS + threadIdx.x));
    DBG_LANE0(3, 1);
    return;
  }
  __shared__ float rowsum[A7_ITEMS][A7_ROWS];
  __shared__ uint32_t rowoff[A7_ITEMS * A7_ROWS + 1];  // exclusive prefix of the row lengths, flattened (pixel, row)
  __shared__ uint32_t rowbeg[A7_ITEMS * A7_ROWS];      // first bin entry of the row
  __shared__ float opx[A7_ITEMS][5];                   // x, y, z, sigma of the pixel's point, div_recip(sigma)
  __shared__ uint32_t otrk[A7_ITEMS];
  __shared__ float term[CK_TERM_CAP];
  __shared__ uint8_t rowof[CK_TERM_CAP];              // which (pixel, row) a term of the pass belongs to
  const int r = threadIdx.x, it = threadIdx.y;
  const int lane = it * A7_ROWS + r;
  const int h = d.window_half;
  const uint32_t shard = blockIdx.x & (VIS_SHARDS - 1);
  const uint32_t n = sc.cnt->shard[shard].heavy;
  const float *__restrict__ pdf = st.pdf;
  const gptr<sdm_labeled_point> cloud_img = as_global(sc.fa->cloud);
  // batches of 16 listed pixels are handed out by a ticket counter (per shard): window sizes are very uneven, a fixed
  // assignment leaves the kernel waiting for the workgroup that drew the long batches.  Which workgroup computes a pixel
  // does not change its value.
  // (Measured and not kept, round 5: the NEXT batch's inputs - ticket, listed pixel, its point and row ends: three dependent
  // round trips - fetched while this batch is computed.  Bit-exact, and the frame went 0.2186-0.2198 -> 0.2230-0.2243 ms in
  // one run with both builds: the prefetched values live across the batch's nine barriers and the kernel loses resident
  // workgroups for them.)
  // (Round 6, the lean version of the same: only the ticket, drawn two batches ahead, and the listed pixel, requested one
  // batch ahead - one word of LDS, one register, 72 registers as before.  Weight stage of `driven` 77 -> 85 us, the benchmark
  // frames +- 0: a workgroup that holds the tickets of its next two batches keeps them while others run dry - the list's
  // tail is three batches long instead of one.)
  __shared__ uint32_t s_q0, s_slow;
  for (;;) {
    __syncthreads();
    if (lane == 0) {
      s_q0 = atomicAdd(&sc.cnt->shard[shard].heavy_ticket, (uint32_t)A7_ITEMS);
      s_slow = 0;  // set by a pixel whose sigma is outside the range div_by is verified for
    }
    __syncthreads();
    const uint32_t q0 = s_q0;
    if (q0 >= n) break;
#ifdef SDM_AB_TIMERS
    if (lane == 0) g_dbg[3][blockIdx.x * 4 + 2] += 1;  // batches of this workgroup
#endif
    const uint32_t q = q0 + it;
    int p = 0;
    uint32_t s = 0, e = 0;
    sdm_labeled_point o;
    if (q < n) {
      p = (int)sc.ck_heavy[shard * sc.cap_heavy + q];
      o = load_point(sc.fa->cloud, p);
      const int i = p / d.W, j = p - i * d.W;
      const int ni = i + r - h;
      if (r <= 2 * h && ni >= 0 && ni < d.H) {
        const int j0 = j - h < 0 ? 0 : j - h;
        const int j1 = j + h >= d.W ? d.W - 1 : j + h;
        s = sc.bin_start[ni * (d.W + 1) + j0];
        e = sc.bin_start[ni * (d.W + 1) + j1 + 1];
      }
      if (r == 0) {
        opx[it][0] = o.x;
        opx[it][1] = o.y;
        opx[it][2] = o.z;
        opx[it][3] = o.sigma;
        opx[it][4] = div_recip(o.sigma);
        if (opx[it][4] == 0.f) s_slow = 1;
        otrk[it] = o.track_id;
      }
    }
    rowbeg[lane] = s;
    rowoff[lane] = e - s;
    __syncthreads();
    if (lane < 64) {  // exclusive prefix over the 256 row lengths by one wave: 4 per lane + wave scan
      uint32_t v[4], sum = 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v[u] = rowoff[lane * 4 + u];
        sum += v[u];
      }
      uint32_t inc = sum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
      }
      uint32_t run = inc - sum;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        rowoff[lane * 4 + u] = run;
        run += v[u];
      }
      if (lane == 63) rowoff[A7_ITEMS * A7_ROWS] = inc;
    }
    __syncthreads();
    const uint32_t total = rowoff[A7_ITEMS * A7_ROWS];
    const uint32_t my_a = rowoff[lane], my_b = rowoff[lane + 1];
    float acc = 0.f;
    for (uint32_t base = 0; base < total; base += CK_TERM_CAP) {
      const uint32_t cnt = total - base < CK_TERM_CAP ? total - base : CK_TERM_CAP;
      // every row marks its slice of the pass (a search for the row of each term cost 8 LDS reads and 40 instructions per
      // term)
      {
        const uint32_t a = my_a > base ? my_a : base;
        const uint32_t b = my_b < base + cnt ? my_b : base + cnt;
        for (uint32_t t = a; t < b; ++t) rowof[t - base] = (uint8_t)lane;
      }
      __syncthreads();
      // lane l computes terms base + l, base + l + 256, ...; four at a time, so that the loads of four
      // independent terms are in flight together (a lane's terms are otherwise a chain of dependent loads)
      auto terms = [&](auto fast) {
      for (uint32_t g0 = base + (uint32_t)lane; g0 < base + cnt; g0 += 4u * 256u) {
        float4 pv[4];
        uint32_t tf[4];
        int px[4];
        bool on[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t g = g0 + (uint32_t)u * 256u;
          on[u] = g < base + cnt;
          px[u] = 0;
          if (on[u]) {
            const int lo = rowof[g - base];
            const uint32_t k = rowbeg[lo] + (g - rowoff[lo]);
            px[u] = lo / A7_ROWS;
            pv[u] = sc.vp4[k];
            tf[u] = sc.vtf[k];
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (!on[u]) continue;
          sdm_labeled_point oo;
          oo.x = opx[px[u]][0];
          oo.y = opx[px[u]][1];
          oo.z = opx[px[u]][2];
          oo.sigma = opx[px[u]][3];
          oo.track_id = (uint16_t)otrk[px[u]];
          bool skip;
          const float t = ck_term<decltype(fast)::value>(flt, pdf, pv[u], tf[u], oo, opx[px[u]][4], skip);
          term[g0 + (uint32_t)u * 256u - base] = skip ? -0.f : t;  // x + (-0) == x: a skipped term leaves the sum untouched
        }
      }
      };
      if (!s_slow) terms(std::true_type{}); else terms(std::false_type{});  // workgroup-uniform
      __syncthreads();
      {
        const uint32_t a = my_a > base ? my_a : base;
        const uint32_t b = my_b < base + cnt ? my_b : base + cnt;
        for (uint32_t t = a; t < b; ++t) acc += term[t - base];
      }
      __syncthreads();
    }
    rowsum[it][r] = acc;
    __syncthreads();
    if (r == 0 && q < n) {
      float ck = 0.f;
      for (int m = 0; m <= 2 * h; ++m) ck += rowsum[it][m];
      ck_store(flt, sc, ck_out, finish, p, o, ck);
    }
    __syncthreads();
  }
  DBG_LANE0(3, 1);
}

// ck_kappa from the per-slab partial images, summed in slab order (multi-GPU path)
__global__ __launch_bounds__(TPB) void k_ck_finish(Dims d, Filter flt, Scratch sc, const float *__restrict__ parts,
                                                   int n_parts, size_t part_stride) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= d.W * d.H) return;
  const sdm_labeled_point o = load_point(sc.fa->cloud, p);
  if (!o.is_valid) {
    sc.pixt[p] = 0;
    return;
  }
  float ck = 0.f;
  for (int g = 0; g < n_parts; ++g) ck += parts[(size_t)g * part_stride + p];
  const float ckk = ck * flt.p_detect + flt.noise_number;
  sc.ck_kappa[p] = ckk;
  sc.pix4[p] = make_float4(o.x, o.y, o.z, ckk);
  sc.pixt[p] = (uint32_t)o.track_id | (1u << 16);
}

// Chunk-owner exchange of the partial ck images (multi-GPU path): the image is cut into `world` chunks of `chunk` pixels,
// shard r owns chunk r.  stage holds, for this shard's chunk, the `world` partial sums of all shards (part s = what shard
// s computed for these pixels, received by an all-to-all); they are added in slab order - the same float sums as
// k_ck_finish forms from whole images - into this shard's chunk of the full image, which is then all-gathered.
// own_part: this shard's partial image (its own part of its own chunk is read from there: the exchange copies nothing from
// a shard to itself); nullptr: part `rank` of the stage holds it (the split entry points, whose caller fills the stage).
__global__ __launch_bounds__(TPB) void k_ck_reduce_chunk(const float *__restrict__ stage, const float *__restrict__ own_part,
                                                         float *__restrict__ full, uint32_t chunk, int world, int rank) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= chunk) return;
  float ck = 0.f;
  for (int g = 0; g < world; ++g) ck += (g == rank && own_part) ? own_part[(size_t)rank * chunk + i] : stage[(size_t)g * chunk + i];
  full[(size_t)rank * chunk + i] = ck;
}

// pass 2 (semantic_dsp_map.h:1041-1119): one wave per U binned particles, one lane per pixel of a particle's window
// (ROUNDS rounds of 64 lanes: 49 pixels at window_half 3, 121 at 5, at most 225).  A dependent fetch costs 1-2 us here
// whatever its size (in-kernel clocks, profiles/r03_in_kernel_timers.txt), and eight waves per SIMD hold fewer particles
// than a frame shows: so a wave issues the fetches of all its U particles level by level - own entries (wave-uniform,
// scalar loads), then sigma beside every window pixel, then the table - and is through after three levels.
// The terms go through LDS and are added in the order of the reference's loops: along each window row from 0.f, then
// the rows; a skipped pixel adds +0.f, which leaves a sum that started at +0.f (and so never is -0.f) as it is.
// lanes of ONE wave hand values to each other through LDS: the scheduling barrier keeps the compiler from moving the
// accesses across it, the wavefront-scope release / acquire pair orders them in the memory model (the barrier alone
// rests on the hardware's in-order LDS path)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
constexpr int WT_WAVES = 4;
constexpr int WT_GRID = 2048;  // x 4 waves = the 8192 waves the chip holds at once
template <int U, int ROUNDS, bool RAW>
__global__ __launch_bounds__(64 * WT_WAVES) void k_weight(Dims d, Filter flt, State st, Scratch sc, const float *__restrict__ ck_raw) {
  static_assert(U * A7_ROWS <= 64, "lanes (u, row) of the row sums");
  DBG_LANE0(4, 0);
  const Frame f = sc.fa->f;  // a copy (uniform registers): stores of the kernel cannot alias it
  // (+ 16 floats per particle: the row sums read term[.][lu][lr * side + c] in lanes (lu, lr); with whole multiples of 32
  // banks between the particles the lanes of two particles met in one bank - 0.29 of the kernel's LDS cycles were conflicts)
  __shared__ float term[WT_WAVES][U][ROUNDS * 64 + 16];
  __shared__ float rowsum[WT_WAVES][U][A7_ROWS];
  if (sc.cnt->overflow) return;
  const uint32_t n = sc.cnt->n_vis;
  const gptr<sdm_labeled_point> cloud_img = as_global(sc.fa->cloud);
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = d.window_half, side = 2 * h + 1, pairs = side * side;
  const float *__restrict__ pdf = st.pdf;
  const size_t slot_base = (size_t)d.v_begin << d.p_n;
  // the lane's window pixel in each round
  int wr[ROUNDS], wc[ROUNDS];
#pragma unroll
  for (int rd = 0; rd < ROUNDS; ++rd) {
    const int q = rd * 64 + lane;
    wr[rd] = q / side;
    wc[rd] = q - wr[rd] * side;
  }
  const uint32_t stride = gridDim.x * WT_WAVES * U;
  for (uint32_t k0 = (blockIdx.x * WT_WAVES + wv) * U; k0 < n; k0 += stride) {
    uint32_t p[U], tf[U], bi[U];
    float4 pv[U];
    float sigma[U], rsig[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      p[u] = tf[u] = bi[u] = 0;
      pv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k0 + u < n) p[u] = sc.vpix[k0 + u], pv[u] = sc.vp4[k0 + u], tf[u] = sc.vtf[k0 + u], bi[u] = sc.bin_idx[k0 + u];
    }
    uint32_t ot[U][ROUNDS];
    float4 o[U][ROUNDS];
    float cr[U][ROUNDS];  // ck_raw != nullptr (Z-slab shards): the pixel's summed ck as the exchange left it; ck + kappa is formed here
    bool all_fast = true;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // sigma of the particle's own pixel (semantic_dsp_map.h:1047); div_by is verified against the division for a
      // range of sigma, div_recip returns 0 outside it
      sigma[u] = k0 + u < n ? cloud_img[p[u]].sigma : 1.f;
      const int i = p[u] / d.W, j = p[u] % d.W;
#pragma unroll
      for (int rd = 0; rd < ROUNDS; ++rd) {
        const int ni = i + wr[rd] - h, nj = j + wc[rd] - h;
        ot[u][rd] = 0;
        o[u][rd] = make_float4(0.f, 0.f, 0.f, 1.f);
        cr[u][rd] = 0.f;
        if (k0 + u < n && rd * 64 + lane < pairs && ni >= 0 && ni < d.H && nj >= 0 && nj < d.W) {
          ot[u][rd] = sc.pixt[ni * d.W + nj];
          o[u][rd] = sc.pix4[ni * d.W + nj];  // x, y, z, ck+kappa (beside the validity word, not behind it)
          if constexpr (RAW) cr[u][rd] = ck_raw[ni * d.W + nj];  // (the same round of loads)
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      rsig[u] = div_recip(sigma[u]);
      all_fast = all_fast && rsig[u] != 0.f;
    }
    if constexpr (RAW) {  // ck * P_d + kappa (semantic_dsp_map.h:1035), the very expression ck_store / k_ck_finish evaluate
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) o[u][rd].w = cr[u][rd] * flt.p_detect + flt.noise_number;
    }
    int right[U];
    auto terms = [&](auto fast) {
      constexpr bool F = decltype(fast)::value;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint16_t ptrack = (uint16_t)(tf[u] & 0xffffu);
        const float ff = flt.forget[(tf[u] >> 16) & 7];
        right[u] = 0;
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
          float t = 0.f;
          const uint16_t otrack = (uint16_t)(ot[u][rd] & 0xffffu);
          if ((ot[u][rd] >> 16) && !(flt.independent && otrack != ptrack)) {  // a valid pixel
            float gk = query_pdf_r<F>(pdf, pv[u].x, o[u][rd].x, sigma[u], rsig[u]) *
                       query_pdf_r<F>(pdf, pv[u].y, o[u][rd].y, sigma[u], rsig[u]) *
                       query_pdf_r<F>(pdf, pv[u].z, o[u][rd].z, sigma[u], rsig[u]);
            if (!flt.independent) {
              if (ptrack != otrack) {
                gk *= flt.id_transition;
              } else {
                if (gk > SDM_MIN_RIGHT_PDF) right[u] = 1;
              }
              gk *= ff;
            }
            t = gk / o[u][rd].w;
          }
          term[wv][u][rd * 64 + lane] = t;
        }
      }
    };
    if (all_fast) terms(std::true_type{}); else terms(std::false_type{});
    unsigned long long rb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) rb[u] = __ballot(right[u]);
    wave_lds_sync();
    // lane (u, r): row r of particle u
    const int lu = lane / A7_ROWS, lr = lane % A7_ROWS;
    if (lu < U && lr < side) {
      float acc = 0.f;
      for (int c = 0; c < side; ++c) acc += term[wv][lu][lr * side + c];
      rowsum[wv][lu][lr] = acc;
    }
    wave_lds_sync();
    if (lu < U && lr == 0 && k0 + lu < n) {
      float a = 0.f;
      for (int m = 0; m < side; ++m) a += rowsum[wv][lu][m];
      uint32_t my_bi = bi[0], my_tf = tf[0];
      float my_w = pv[0].w;
      bool right_id = rb[0] != 0ull;
#pragma unroll
      for (int u = 1; u < U; ++u)
        if (lu == u) my_bi = bi[u], my_tf = tf[u], my_w = pv[u].w, right_id = rb[u] != 0ull;
      const size_t li = (size_t)my_bi - slot_base;
      const uint32_t fc = (my_tf >> 16) & 0xffu;
      const SlotRef sr = slot_ref_li(st, d.p_n, li);
      sr.set_w(my_w * (a * flt.p_detect + 1.f - flt.p_detect));
      sr.set_status(ST_UPDATED);
      st.vflag[li >> d.p_n] = VF_DIRTY;
      mark_tile(st, li >> d.p_n, f.epoch);
      sr.set_ts((uint16_t)f.gts);
      if (!flt.independent) {
        uint32_t nf = right_id ? 0u : (fc < 5u ? fc + 1u : fc);
        if (nf != fc) st.forget[li] = (uint8_t)nf;
      }
    }
    wave_lds_sync();
  }
  DBG_LANE0(4, 1);
}

// ------------------------------------------------------------------------------------ A8 / A9
// Birth raster order (semantic_dsp_map.h:778-800): 9 interleaved stride-3 passes.
__device__ __forceinline__ void birth_seq_to_pixel(const BirthOrder &bo, int q, int &i, int &j) {
  int p = 0;
#pragma unroll
  for (int k = 1; k < 9; ++k)
    if (q >= bo.off[k]) p = k;
  int r = q - bo.off[p];
  int cols = bo.cols[p];
  i = p / 3 + 3 * (r / cols);
  j = p % 3 + 3 * (r % cols);
}

__global__ __launch_bounds__(TPB) void k_birth_flags(Dims d, BirthOrder bo, Scratch sc) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= d.W * d.H) return;
  int i, j;
  birth_seq_to_pixel(bo, q, i, j);
  sc.b_valid[q] = as_global(sc.fa->cloud)[i * d.W + j].is_valid ? 1u : 0u;
}

// One thread per birth candidate b = q * nb + n (q = position in the raster order, n = copy).
// The table cursor of the reference advances by 3 per copy of every valid pixel in raster order
// (semantic_dsp_map.h:1180-1188, basic_algorithms.h:426-440), so the draw index is a function of the
// exclusive rank of q among valid pixels.
__global__ __launch_bounds__(TPB) void k_birth_candidates(Dims d, Filter flt, BirthOrder bo, State st, Scratch sc) {
  const Frame f = sc.fa->f;  // a copy (uniform registers): stores of the kernel cannot alias it
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t total = (uint32_t)(d.W * d.H) * (uint32_t)flt.nb;
  if (b >= total) return;
  int q = (int)(b / (uint32_t)flt.nb), n = (int)(b % (uint32_t)flt.nb);
  int i, j;
  birth_seq_to_pixel(bo, q, i, j);
  const sdm_labeled_point pt = load_point(sc.fa->cloud, (size_t)(i * d.W + j));
  uint32_t key = d.V;  // sorts behind every real voxel
  float x = pt.x, y = pt.y, z = pt.z;
  if (pt.is_valid) {
    if (flt.use_rng) {
      long long draw = (long long)sc.cur->birth_cursor + 3ll * ((long long)flt.nb * sc.b_rank[q] + n);
      float nx = pt.sigma * st.noise[(draw + 1) % flt.noise_n];
      float ny = pt.sigma * st.noise[(draw + 2) % flt.noise_n];
      float nz = pt.sigma * st.noise[(draw + 3) % flt.noise_n];
      x = pt.x + nx;
      y = pt.y + ny;
      z = pt.z + nz;
    }
    uint32_t rx, ry, rz;
    uint32_t v = global_pos_to_voxel(d, f, x, y, z, rx, ry, rz);
    if (v != INVALID_INDEX && rz >= d.rz_begin && rz < d.rz_begin + d.rz_count) key = v;
  }
  sc.bkey_a[b] = key;
  sc.bval_a[b] = b;
  sc.bpos[b] = make_float4(x, y, z, __uint_as_float((uint32_t)pt.track_id | ((uint32_t)pt.label_id << 16)));
}

__global__ void k_birth_cursor(Dims d, Filter flt, Scratch sc) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int last = d.W * d.H - 1;
  uint32_t n_valid = sc.b_rank[last] + sc.b_valid[last];
  sc.cnt->n_valid_px = n_valid;
  sc.cnt->n_birth_attempts = n_valid * (uint32_t)flt.nb;
  if (flt.use_rng) {
    long long c = (long long)sc.cur->birth_cursor + 3ll * flt.nb * (long long)n_valid;
    sc.cur->birth_cursor = (int32_t)(c % flt.noise_n);
  }
}

// `while (run > thr) thr += wpp;` of resampleParticlesInVoxel (semantic_dsp_map.h:1448-1519).  wpp is the voxel's weight sum
// over S/2, capped at 1 - and a weight the update has just multiplied up can be in the hundreds before the next sweep clamps
// it: the loop then runs once per unit of weight, in ONE lane of a sparse wave, and that lane is the workgroup's, and a
// handful of such workgroups the kernel's, last 15 us: k_birth_replay's median workgroup lived 7 us, its slowest 22 - for four
// rounds (tools/probes/timers_frame_gaps.py; tools/probes/timers_birth.py had put the time between "rows arrived" and
// "candidates chosen", where there is nothing but registers - and this loop).  Closed form: the kernel 23 -> 9 us, the
// benchmark frame 0.208 -> 0.1965 ms, `driven` 0.2925 -> 0.279 ms, two rounds alternating in one call.
// With wpp capped, thr is a whole number (1 + 1 + ...) and the loop's result is the
// smallest whole number that is not below run: ceilf, exact for every float (below 2^24 the additions of 1.f are exact too;
// beyond it the reference's loop does not end).  Uncapped, run is at most the weight sum = S/2 * wpp: a few rounds.
__device__ __forceinline__ float resample_next_threshold(float run, float thr, float wpp) {
  if (!(run > thr)) return thr;
  if (wpp == 1.f) return ceilf(run);
  while (run > thr) thr += wpp;
  return thr;
}

// resampleParticlesInVoxel (semantic_dsp_map.h:1448-1519) on the register copy of one voxel (status row, owner row).
// wv / trk: the voxel's weight and track rows as they were when the replay started - the slots the resampling looks at
// (UPDATED ones) are not written by births, so the rows are still theirs.  (Loaded here, behind the stores of the
// insertions before it, they cost a store drain and a round trip in the middle of the replay.)
template <int S>
__device__ __forceinline__ bool resample_voxel(const Dims &d, State &st, size_t base, uint8_t (&stv)[S], uint16_t (&own)[S],
                                               uint32_t n_alias, bool touched, uint32_t fbits, float (&wv)[S], const uint16_t (&trk)[S]) {
  // (on the register copy of the voxel: the caller stores the rows)
  float weight_sum = 0.f;
  uint32_t updated = 0;
#pragma unroll
  for (int i = 1; i < S; ++i)
    if (stv[i] == ST_UPDATED) {
      weight_sum += wv[i];
      ++updated;
    }
  const uint32_t trigger = S >> 1;
  if (updated <= trigger) return false;
  if (weight_sum < 0.01f) {
#pragma unroll
    for (int i = 1; i < S; ++i)
      if (stv[i] == ST_UPDATED) {
        stv[i] = ST_INVALID;
        owner_erase_local(st, base + i, trk[i], own[i], n_alias, touched, fbits, i);  // removeParticleFromObj
      }
    return true;
  }
  float wpp = weight_sum / (float)trigger;
  if (wpp > 1.f) wpp = 1.f;
  float run = 0.f, thr = wpp;
#pragma unroll
  for (int i = 1; i < S; ++i)
    if (stv[i] == ST_UPDATED) {
      run += wv[i];
      if (run < thr) {
        stv[i] = ST_INVALID;
        owner_erase_local(st, base + i, trk[i], own[i], n_alias, touched, fbits, i);
      } else {
        wv[i] = wpp;
        thr += wpp;
        thr = resample_next_threshold(run, thr, wpp);
      }
    }
  return true;
}

// The same with every row read where it is needed (owner_erase / owner_insert of sdm_internal.h): used by the sequential
// replay below only.
template <int S>
__device__ __forceinline__ bool resample_voxel_seq(const Dims &d, State &st, size_t base, uint8_t (&stv)[S]) {
  float weight_sum = 0.f;
  uint32_t updated = 0;
  unsigned char *const rec = st.rec + base / S * rec_bytes(S);  // (base = lv * S)
  float wv[S];
  {
    uint16_t t1[S], t2[S];
    uint8_t l1[S], s1[S];
    rec_load<S>(rec, wv, t1, t2, l1, s1);
  }
#pragma unroll
  for (int i = 1; i < S; ++i)
    if (stv[i] == ST_UPDATED) {
      weight_sum += wv[i];
      ++updated;
    }
  const uint32_t trigger = S >> 1;
  if (updated <= trigger) return false;
  if (weight_sum < 0.01f) {
#pragma unroll
    for (int i = 1; i < S; ++i)
      if (stv[i] == ST_UPDATED) {
        stv[i] = ST_INVALID;
        SlotRef{rec, S - 1, (uint32_t)i - 1u}.set_status(ST_INVALID);
        owner_erase(st, base + i, SlotRef{rec, S - 1, (uint32_t)i - 1u}.track());  // removeParticleFromObj
      }
    return true;
  }
  float wpp = weight_sum / (float)trigger;
  if (wpp > 1.f) wpp = 1.f;
  float run = 0.f, thr = wpp;
#pragma unroll
  for (int i = 1; i < S; ++i)
    if (stv[i] == ST_UPDATED) {
      run += wv[i];
      if (run < thr) {
        stv[i] = ST_INVALID;
        SlotRef{rec, S - 1, (uint32_t)i - 1u}.set_status(ST_INVALID);
        owner_erase(st, base + i, SlotRef{rec, S - 1, (uint32_t)i - 1u}.track());
      } else {
        SlotRef{rec, S - 1, (uint32_t)i - 1u}.set_w(wpp);
        thr += wpp;
        thr = resample_next_threshold(run, thr, wpp);
      }
    }
  return true;
}

// The literal candidate-by-candidate walk of one voxel's segment (the reference's loop): k_birth_replay<S, true>.  The
// frames of a map replay in closed form (k_birth_replay<S, false>, below) - except where the closed form's premise fails:
// once global_time_stamp has passed 65535 the 16-bit time stamp of a fresh particle can be older than the voxel's 32-bit
// slab stamp; the reference's comparison (operations.h:810-816) then finds the fresh particle vacant again, every
// candidate lands in the same slot and the voxel never fills.  The host picks the kernel by the time stamp.
template <int S>
__device__ __forceinline__ void birth_replay_sequential(const Dims &d, const Frame &f, const Filter &flt, State &st, const Scratch &sc,
                                                     const uint32_t *__restrict__ skey, const uint32_t *__restrict__ sval, uint32_t total,
                                                     uint32_t t, uint32_t v, uint32_t smax, uint32_t &n_success_out, uint32_t &n_resamp_out) {
  const size_t base = (size_t)(v - d.v_begin) * S;
  uint8_t stv[S];
  uint16_t tsv[S];
  unsigned char *const rec = rec_ptr(st, S, v - d.v_begin);
  rec_load_st_ts<S>(rec, stv, tsv);
  bool resampled = false, checked = false;
  uint32_t n_success = 0, n_resamp = 0;
  // The candidates of a voxel are consecutive in the sorted list; eight at a time are fetched before the first is
  // replayed (key, index, then the eight positions: two dependent loads per batch instead of per candidate).
  bool done = false;
  for (uint32_t u0 = t; u0 < total && !done; u0 += 8) {
    uint32_t vs[8];
    bool ok[8];
    float4 bps[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t u = u0 + j;
      ok[j] = u < total && skey[u] == v;
      vs[j] = ok[j] ? sval[u] : 0u;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (ok[j]) bps[j] = sc.bpos[vs[j]];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (done) break;
      if (!ok[j]) {  // end of this voxel's segment
        done = true;
        break;
      }
      const float4 bp = bps[j];
      const uint32_t tl = __float_as_uint(bp.w);
      const uint16_t track = (uint16_t)(tl & 0xffffu);
      const uint8_t label = (uint8_t)((tl >> 16) & 0xffu);
      bool changed = false;  // did this birth change the voxel (insert or triggered resample)?
#pragma unroll
      for (int attempt = 0; attempt < 2; ++attempt) {
        int slot = -1;
#pragma unroll
        for (int i = S - 1; i >= 1; --i)
          if (stv[i] == ST_INVALID || (uint32_t)tsv[i] < smax) slot = i;  // lowest vacant slot
        if (slot > 0) {
          // addNewParticleWithSemantics (operations.h:171-184)
          st.pos4[base + slot] = make_float4(bp.x, bp.y, bp.z, 0.f);
          st.forget[base + slot] = 0;
          const SlotRef sr{rec, S - 1, (uint32_t)slot - 1u};
          sr.set_w(SDM_OCC_INIT_WEIGHT);
          sr.set_ts((uint16_t)f.gts);
          sr.set_track(track);
          sr.set_label(label);
          sr.set_status(ST_REGULAR_BORN);
          if ((int)track <= d.max_movable) {  // addParticleToObj
            if (!owner_insert(st, base + slot, track)) sc.cnt->overflow = 1;
            flag_owner_chunk(st, base + slot);
          }
#pragma unroll
          for (int i = 1; i < S; ++i)
            if (i == slot) {
              stv[i] = ST_REGULAR_BORN;
              tsv[i] = (uint16_t)f.gts;
            }
          changed = true;
          ++n_success;
          break;
        }
        // voxel full
        if (!flt.consider_depth_noise) break;     // no retry in the no-noise flavour
        if (attempt == 1 || resampled || checked) break;
        if (resample_voxel_seq<S>(d, st, base, stv)) {
          resampled = true;
          changed = true;
          n_resamp++;
        } else {
          checked = true;
          break;
        }
      }
      if (!flt.consider_depth_noise && !resampled && !checked) {
        // semantic_dsp_map.h:1165-1170: after every add, resample until it has triggered once
        if (resample_voxel_seq<S>(d, st, base, stv)) {
          resampled = true;
          changed = true;
          n_resamp++;
        } else {
          checked = true;
        }
      }
      // fixed point: the voxel is full, its one resample per frame is used up (or cannot trigger, since births
      // never add UPDATED particles), and this birth changed nothing -> no later birth of this voxel can either
      if (!changed && (resampled || checked)) done = true;
    }
  }
  n_success_out = n_success;
  n_resamp_out = n_resamp;
}

// Ordered per-voxel replay of the births (addNewbornParticleAndResample / ...WithNoiseAndResample,
// semantic_dsp_map.h:1148-1230; addParticleByGlobalPos, operations.h:782-803).  The sorted list keeps
// raster order inside each voxel segment; the segment head thread replays it and stops at the fixed point
// (voxel full and its one resample per frame used up or impossible).
template <int S, bool LITERAL>
__global__ __launch_bounds__(TPB) void k_birth_replay(Dims d, Filter flt, State st, Scratch sc,
                                                      const uint32_t *__restrict__ skey, const uint32_t *__restrict__ sval,
                                                      uint32_t total) {
  const Frame f = sc.fa->f;  // a copy (uniform registers): stores of the kernel cannot alias it
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
#ifndef SDM_TIMERS_BIRTH
  if (threadIdx.x == 0) DBG_PUT(0, DBG_T());
#endif
  // The kernel is a chain of dependent loads in the few lanes that are segment heads (one candidate in fifty): left where
  // they are, nearly every wave of the launch carries one or two of them through the whole chain.  So the heads of a
  // workgroup's 256 candidates are compacted first (LDS) and replayed by its first lanes: a quarter of the waves do all
  // the work, the others leave.
  __shared__ uint32_t heads[TPB], head_key[TPB];
  __shared__ uint32_t n_heads;
  if (threadIdx.x == 0) n_heads = 0;
  __syncthreads();
  if (t < total) {
    const uint32_t key = skey[t];
    const uint32_t prev = t > 0 ? skey[t - 1] : 0xffffffffu;
    if (key < d.V && prev != key) {
      const uint32_t q = atomicAdd(&n_heads, 1u);
      heads[q] = t;
      head_key[q] = key;
    }
  }
  __syncthreads();
  if (threadIdx.x >= n_heads) return;
#ifndef SDM_TIMERS_BIRTH
  if (threadIdx.x == 0) DBG_PUT(1, DBG_T());
#endif
  t = heads[threadIdx.x];
  const uint32_t v = head_key[threadIdx.x];
  // How many candidates does the segment hold?  At most 2 (S-1) of them can ever be inserted (the voxel's vacant slots, and
  // after its one resampling the slots that freed), so the count is needed up to LMAX only.  The keys that give the count,
  // the candidates' indices and (below) the voxel's record are all requested in one round: the voxel is known from the
  // compaction, and which of the candidates make it is a selection among values that are already here.
  constexpr int LMAX = 2 * (S - 1) + 1;
  uint32_t kb[LMAX], sv[LMAX];
  kb[0] = v;
#pragma unroll
  for (int j = 1; j < LMAX; ++j) kb[j] = t + j < total ? skey[t + j] : 0xffffffffu;
  if constexpr (!LITERAL) {
#pragma unroll
    for (int j = 0; j < LMAX; ++j) sv[j] = t + j < total ? sval[t + j] : 0u;
  }
  uint32_t rx, ry, rz;
  voxel_to_ring(d, v, rx, ry, rz);
  const uint32_t smax = stamp_max(st, rx, ry, rz);
  if constexpr (LITERAL) {
    uint32_t ns = 0, nr = 0;
    birth_replay_sequential<S>(d, f, flt, st, sc, skey, sval, total, t, v, smax, ns, nr);
    if (ns || nr) {
      st.vflag[v - d.v_begin] = VF_DIRTY;
      mark_tile(st, v - d.v_begin, f.epoch);
    }
    if (ns) atomicAdd(&sc.cnt->shard[blockIdx.x & (VIS_SHARDS - 1)].birth, ns);
    if (nr) atomicAdd(&sc.cnt->shard[blockIdx.x & (VIS_SHARDS - 1)].resample, nr);
    return;
  } else {
  const size_t base = (size_t)(v - d.v_begin) * S;
  uint8_t stv[S];
  uint16_t tsv[S];
  unsigned char *const rec = rec_ptr(st, S, v - d.v_begin);
  // The voxel's whole record, its forget counts and (below) its owner entries in one round.  The replay works on these
  // register copies and stores every row ONCE at the end: stored field by field per insertion - seven stores and up to
  // three for the owner set, up to fourteen insertions - a head ran into the 64 memory operations a wave may have in
  // flight and waited for its own stores to retire, a microsecond per insertion (tools/probes/timers_moves.py measured it on the
  // replay of the moved copies, which had the same shape).
  float wv0[S];
  uint16_t trk0[S];
  uint8_t lab0[S], fg[S];
  rec_load<S>(rec, wv0, tsv, trk0, lab0, stv);
  __builtin_memcpy(fg, st.forget + base, S);
  // the voxel's owner entries and the length of the table of older memberships (addParticleToObj / removeParticleFromObj),
  // and the rows the one resampling of the voxel reads: everything the replay needs of the voxel, in one round
  uint16_t own[S];
  load_vec<(2 * S < 16 ? 2 * S : 16)>(own, st.owner + base);
  const uint32_t n_alias = st.alias[0];
  bool alias_touched = false;
  uint32_t fbits = n_alias ? alias_filter_bits<S>(st, base) : 0u;  // (with the voxel's rows: one round)
  uint32_t L = 0;  // candidates of the segment, capped at LMAX
  {
    bool run = true;
#pragma unroll
    for (int j = 0; j < LMAX; ++j) {
      run = run && kb[j] == v;
      L += run ? 1u : 0u;
    }
  }
  // ---- The replay in closed form.  The reference walks the segment candidate by candidate (addNewbornParticleAndResample
  // / ...WithNoiseAndResample, semantic_dsp_map.h:1148-1230); what that walk does to a voxel follows from three numbers:
  //   * an insertion takes the LOWEST vacant slot (operations.h:790-796), so the first candidates fill the voxel's vacant
  //     slots in ascending order;
  //   * the voxel is resampled at most once per frame (:1166-1169, 1212), and births never add UPDATED particles, so
  //     whether the resampling triggers does not depend on how many candidates came before it.  Noise flavour (:1205-1228):
  //     it is tried when a candidate finds the voxel full, and that candidate retries once; plain flavour (:1160-1170): it
  //     is tried after the first candidate, inserted or not;
  //   * afterwards the remaining candidates fill what is vacant then, ascending, and the first one that finds the voxel
  //     full with the resampling used up (or impossible) ends the walk: nothing later can change the voxel.
  // Phase A = insertions before the resampling, phase B = after it.  One pass over the slots per phase instead of a loop
  // over candidates whose every iteration was a few hundred dependent instructions for the slowest lane of the wave.
  // (Round 5, measured and not kept - none of them moved the kernel's 23 us or the frame, three rounds alternating in one
  // call each: one s_waitcnt vmcnt(0) here, in front of the divergent code, which takes the compiler's own vmcnt(0) out of
  // every branch of the resampling - those also wait for the stores before them, the counter being in order; the same
  // behind the requests for the candidates' positions, so that no insertion waits for the stores of the one before it;
  // the resampling as a chain of selects with the voxel's rows stored once each, whole, no branch per slot; the alias
  // paths compiled out.  The instruction cache is not it either: 850 misses per launch over 128 caches,
  // tools/pmc_icache.sh.  In-kernel clocks say where the time is NOT - rows arrive 2 us after the start in every
  // workgroup, tools/probes/timers_birth.py - but their checkpoints inside straight-line code are not to be trusted:
  // s_memrealtime orders with memory operations only, the compiler places it anywhere between two of them.)
#ifdef SDM_TIMERS_BIRTH  // (the clock read is made to depend on the rows: it cannot be taken before they have arrived)
  if (threadIdx.x == 0) DBG_PUT(0, DBG_T() + (stv[1] == 0xEE ? 1 : 0) + (own[1] == 0xEEEE ? 1 : 0) + (wv0[1] == 1234.5f ? 1 : 0) + (trk0[1] == 0xEEEE ? 1 : 0) + (kb[LMAX - 1] == 0xEEEEEEEEu ? 1 : 0) + (sv[LMAX - 1] == 0xEEEEEEEEu ? 1 : 0));
#endif
  const bool noise_flavour = flt.consider_depth_noise != 0;
  uint32_t vac0 = 0;
#pragma unroll
  for (int i = 1; i < S; ++i)
    if (stv[i] == ST_INVALID || (uint32_t)tsv[i] < smax) vac0 |= 1u << i;
  const uint32_t V0 = (uint32_t)__popc(vac0);
  const uint32_t nA = noise_flavour ? (L < V0 ? L : V0) : ((L >= 1 && V0 >= 1) ? 1u : 0u);
  int cand[S];  // candidate (position in the segment) that goes into slot i, -1: none
#pragma unroll
  for (int i = 1; i < S; ++i) {
    const uint32_t rank = (uint32_t)__popc(vac0 & ((1u << i) - 1u));
    const bool take = ((vac0 >> i) & 1u) && rank < nA;
    cand[i] = take ? (int)rank : -1;
    if (take) {
      stv[i] = ST_REGULAR_BORN;
      tsv[i] = (uint16_t)f.gts;
    }
  }
  const bool try_resample = noise_flavour ? L > V0 : L >= 1;
  uint32_t n_resamp = 0;
  if (try_resample && resample_voxel<S>(d, st, base, stv, own, n_alias, alias_touched, fbits, wv0, trk0)) n_resamp = 1;
  uint32_t nB = 0;
  if (try_resample) {
    const uint32_t consumed = noise_flavour ? V0 : 1u;  // (noise flavour: the candidate that found the voxel full retries)
    uint32_t vac1 = 0;
#pragma unroll
    for (int i = 1; i < S; ++i)
      if (stv[i] == ST_INVALID || (uint32_t)tsv[i] < smax) vac1 |= 1u << i;
    const uint32_t V1 = (uint32_t)__popc(vac1), left = L - consumed;
    nB = left < V1 ? left : V1;
#pragma unroll
    for (int i = 1; i < S; ++i) {
      const uint32_t rank = (uint32_t)__popc(vac1 & ((1u << i) - 1u));
      if (((vac1 >> i) & 1u) && rank < nB) cand[i] = (int)(consumed + rank);
    }
  }
  const uint32_t n_success = nA + nB;
  bool any_owner = false;
#ifdef SDM_TIMERS_BIRTH
  if (threadIdx.x == 0) DBG_PUT(1, DBG_T() + (n_success == 77u ? 1 : 0) + (n_resamp == 77u ? 1 : 0) + (stv[S - 1] == 0xEE ? 1 : 0));
#endif
  // the candidates that made it: index, then position / track / label - two rounds for all of them
  uint32_t cidx[S];
  float4 bps[S];
#pragma unroll
  for (int i = 1; i < S; ++i) {
    cidx[i] = 0;
    if (cand[i] >= 0) {
#pragma unroll
      for (int j = 0; j < LMAX; ++j) cidx[i] = cand[i] == j ? sv[j] : cidx[i];
    }
  }
#pragma unroll
  for (int i = 1; i < S; ++i) bps[i] = sc.bpos[cidx[i]];  // (unconditional - entry 0 where the slot takes nobody: all S - 1 requests in one round)
  __builtin_amdgcn_sched_barrier(0);
#ifdef SDM_TIMERS_BIRTH
  if (threadIdx.x == 0) DBG_PUT(2, DBG_T() + (bps[1].x == 1234.5f ? 1 : 0) + (bps[S - 1].x == 1234.5f ? 1 : 0));
#endif
#pragma unroll
  for (int i = 1; i < S; ++i) {
    if (cand[i] < 0) continue;
    const float4 bp = bps[i];
    const uint32_t tl = __float_as_uint(bp.w);
    const uint16_t track = (uint16_t)(tl & 0xffffu);
    const uint8_t label = (uint8_t)((tl >> 16) & 0xffu);
    // addNewParticleWithSemantics (operations.h:171-184)
    st.pos4[base + i] = make_float4(bp.x, bp.y, bp.z, 0.f);
    fg[i] = 0;
    wv0[i] = SDM_OCC_INIT_WEIGHT;
    tsv[i] = (uint16_t)f.gts;
    trk0[i] = track;
    lab0[i] = label;
    stv[i] = ST_REGULAR_BORN;
    if ((int)track <= d.max_movable) {  // addParticleToObj
      if (!owner_insert_local(st, base + i, track, own[i], n_alias, alias_touched, fbits, i)) sc.cnt->overflow = 1;
      any_owner = true;
    }
  }
  if (n_success || n_resamp) {  // the voxel's rows, once
    rec_store_all<S>(rec, wv0, tsv, trk0, lab0, stv);
    __builtin_memcpy(st.owner + base, own, 2 * S);
    if (n_success) __builtin_memcpy(st.forget + base, fg, S);
    if (any_owner) flag_owner_chunk(st, base + 1);  // (a voxel's slots lie in one chunk)
  }
  if (threadIdx.x == 0) {
#ifdef SDM_TIMERS_BIRTH
    DBG_PUT(3, DBG_T());
#else
    DBG_PUT(2, DBG_T());
    DBG_PUT(3, n_success);
#endif
  }
  // same-address atomics retire one at a time: counters every wave bumps are sharded by block
  if (n_success || n_resamp) {
    st.vflag[v - d.v_begin] = VF_DIRTY;
    mark_tile(st, v - d.v_begin, f.epoch);
  }
  if (n_success) atomicAdd(&sc.cnt->shard[blockIdx.x & (VIS_SHARDS - 1)].birth, n_success);
  if (n_resamp) atomicAdd(&sc.cnt->shard[blockIdx.x & (VIS_SHARDS - 1)].resample, n_resamp);
  }
}

// ------------------------------------------------------------------------------------ N1
// PointCloudTools::generateLabeledPointCloud (utils/pointcloud_tools.h:88-310), general path: merge the masks into a
// track-id image, back-project every valid depth pixel in double, cast to float, attach label and sigma.
// PINNED: K^-1 is (1/fx, -cx/fx, 1/fy, -cy/fy) in double; K^-1*(j,i,1) = (ifx*j + icx, ify*i + icy, 1);
// camera-to-global = ((r0*x + r1*y) + r2*z) + t with Eigen's toRotationMatrix formula in double.
struct CloudArgs {
  double R[9], t[3];
  double ifx, icx, ify, icy;
  double dmin, dmax;
  float sigma0, sigma1;
  int consider_depth_noise, consider_instance, n_objects, has_static;
  int sky_instance, has_bbox;
  int track[MAX_CLOUD_OBJECTS], label[MAX_CLOUD_OBJECTS];
};

// manualResize (pointcloud_tools.h:1104-1133): dst(i, j) = src(min(int(i / scale), rows - 1), min(int(j / scale), cols - 1))
template <typename T>
__global__ __launch_bounds__(TPB) void k_manual_resize(const T *__restrict__ src, T *__restrict__ dst, int src_w, int src_h,
                                                       int dst_w, int dst_h, float scale_inv) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= dst_w * dst_h) return;
  const int i = p / dst_w, j = p - i * dst_w;
  int si = (int)((float)i * scale_inv), sj = (int)((float)j * scale_inv);
  si = si < src_h - 1 ? si : src_h - 1;
  sj = sj < src_w - 1 ? sj : src_w - 1;
  dst[p] = src[(size_t)si * src_w + sj];
}

__global__ __launch_bounds__(TPB) void k_labeled_cloud(Dims d, CloudArgs a, const float *__restrict__ depth,
                                                       const uint8_t *__restrict__ static_mask,
                                                       const uint16_t *__restrict__ label_to_inst,
                                                       const uint8_t *__restrict__ obj_masks,
                                                       const double *__restrict__ bbox,
                                                       sdm_labeled_point *__restrict__ cloud) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int hw = d.W * d.H;
  if (p >= hw) return;
  const int i = p / d.W, j = p % d.W;
  const float dv = depth[p];
  sdm_labeled_point o;
  if (isnan(dv) || (double)dv < a.dmin || (double)dv > a.dmax) {  // :228
    // PINNED: the reference leaves the fields of an invalid point uninitialised
    o.x = o.y = o.z = 0.f;
    o.sigma = a.consider_depth_noise ? a.sigma0 : 0.1f;
    o.track_id = 0;
    o.label_id = 0;
    o.is_valid = 0;
    cloud[p] = o;
    return;
  }
  // track id image: static mask first (:121-156), then every movable object's mask in order (:163-213)
  uint32_t inst = 65535u;
  int label = 0;
  bool from_object = false;
  if (a.has_static) inst = label_to_inst[(uint32_t)static_mask[p] + 1u > 255u ? 255u : (uint32_t)static_mask[p] + 1u];
  if (a.consider_instance)
    for (int k = 0; k < a.n_objects; ++k)
      if (obj_masks[(size_t)k * hw + p] > 0) {
        inst = (uint32_t)a.track[k];
        label = a.label[k];
        from_object = true;
      }
  if (a.sky_instance >= 0 && inst == (uint32_t)a.sky_instance) {  // ZED2: sky pixels are invalid (:236-242)
    o.x = o.y = o.z = 0.f;
    o.sigma = a.consider_depth_noise ? a.sigma0 : 0.1f;
    o.track_id = 0;
    o.label_id = 0;
    o.is_valid = 0;
    cloud[p] = o;
    return;
  }
  const double x = (a.ifx * (double)j + a.icx) * (double)dv;  // :243
  const double y = (a.ify * (double)i + a.icy) * (double)dv;
  const double z = (double)dv;
  const double gx = ((a.R[0] * x + a.R[1] * y) + a.R[2] * z) + a.t[0];  // :247
  const double gy = ((a.R[3] * x + a.R[4] * y) + a.R[5] * z) + a.t[1];
  const double gz = ((a.R[6] * x + a.R[7] * y) + a.R[8] * z) + a.t[2];
  if (a.has_bbox && a.consider_instance && (int)inst < d.max_movable) {  // ZED2 segmentation-noise filter (:254-272)
    // track_id_*_map[instance]: the box of the last object with that track id, zeros when there is none
    double b[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < a.n_objects; ++k)
      if ((uint32_t)a.track[k] == inst)
        for (int c = 0; c < 6; ++c) b[c] = bbox[k * 6 + c];
    if (gx < b[0] || gx > b[1] || gy < b[2] || gy > b[3] || gz < b[4] || gz > b[5]) {
      o.x = (float)gx;
      o.y = (float)gy;
      o.z = (float)gz;
      o.sigma = a.consider_depth_noise ? a.sigma0 + a.sigma1 * dv : 0.1f;  // PINNED: left unset by the reference
      o.track_id = 65535;
      o.label_id = 0;
      o.is_valid = 1;
      cloud[p] = o;
      return;
    }
  }
  if ((int)inst > d.max_movable) {  // :277-283: static instance -> its label
    label = 0;
    if (a.has_static)
      for (int l = 0; l < 256; ++l)
        if (label_to_inst[l] == inst) {
          label = l;
          break;
        }
  } else if (!from_object) {
    label = 0;  // movable id that came out of the static mask table: track_to_label_id_map default (:282)
  }
  o.x = (float)gx;
  o.y = (float)gy;
  o.z = (float)gz;
  o.sigma = a.consider_depth_noise ? a.sigma0 + a.sigma1 * dv : 0.1f;  // :284-289
  o.track_id = (uint16_t)inst;
  o.label_id = (uint8_t)label;
  o.is_valid = 1;
  cloud[p] = o;
}

// ------------------------------------------------------------------------------------ utilities
__global__ __launch_bounds__(TPB) void k_count_live(Dims d, State st, unsigned long long *out) {
  uint32_t lv = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t c = 0, cv = 0;
  if (lv < d.v_count) {
    uint32_t v = d.v_begin + lv, rx, ry, rz;
    voxel_to_ring(d, v, rx, ry, rz);
    uint32_t smax = stamp_max(st, rx, ry, rz);
    for (uint32_t i = 1; i < d.S; ++i) {
      const SlotRef sr = slot_ref(st, d.S, lv, i);
      if (sr.status() != ST_INVALID && (uint32_t)sr.ts() >= smax) c++;
    }
    // bits 36..: voxels that pass isVoxelValid and hold a live slot (the ones the sweep fetches in full)
    const uint32_t t0 = st.vts[lv];
    if (c && t0 != 0 && t0 >= smax) cv = 1;
  }
  for (int off = 32; off > 0; off >>= 1) {
    c += __shfl_down(c, off, 64);
    cv += __shfl_down(cv, off, 64);
  }
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long)c | ((unsigned long long)cv << 36));
}

__global__ __launch_bounds__(TPB) void k_count_owner(Dims d, State st, uint16_t track, unsigned long long *out) {
  size_t n = (size_t)d.v_count * d.S;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t c = 0;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (st.owner[i] == track) c++;
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // older memberships the reference's set still holds
    uint32_t na = st.alias[0];
    if (na > st.alias_cap) na = st.alias_cap;
    for (uint32_t k = 0; k < na; ++k)
      if (st.alias[3 + 2 * k] == track) c++;
  }
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long)c);
}

// pack / unpack between the C-ABI's SoA dump format and pos4
__global__ __launch_bounds__(TPB) void k_pack_pos4(float4 *pos4, uint8_t *forget_plane, const float *px, const float *py, const float *pz,
                                                   const uint8_t *forget, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    pos4[i] = make_float4(px[i], py[i], pz[i], 0.f);
    forget_plane[i] = forget[i];
  }
}
__global__ __launch_bounds__(TPB) void k_unpack_pos4(const float4 *pos4, const uint8_t *forget_plane, float *px, float *py, float *pz, uint8_t *forget,
                                                     size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    float4 q = pos4[i];
    px[i] = q.x;
    py[i] = q.y;
    pz[i] = q.z;
    forget[i] = forget_plane[i];
  }
}

// stable compaction of voxels by result code (getOccupancyResult's emission order = storage order,
// semantic_dsp_map.h:1244,1353): flag pass, scan, scatter.
// The result lists (getOccupancyResult's output side, semantic_dsp_map.h:1258-1376): the voxels whose result says
// "occupied" (or "free") in ascending voxel order.  Two launches over one byte per voxel:
//   k_emit_mark   a thread takes EM_VPT consecutive voxels: the flag bytes first - a voxel whose flag says that its result
//                 entry holds the "unobserved" constant (VR_UNOBSERVED: most of a map) is on neither list and its entry is
//                 not read - then the entries of the others; the thread's selection as a bit mask, the workgroup's count;
//   k_emit_write  workgroups that selected something find their place in the list (sum of the counts before them), rank
//                 their voxels (mask popcounts) and write the points.
// (Rounds 1-3: a flag word per voxel, a device-wide scan of 16.7 M words and a third pass over all of them - 0.3 ms of
// the 1.2 ms a SemanticDSPMap::update call took.)
constexpr int EM_VPT = 8;
constexpr uint32_t EM_CHUNK = TPB * EM_VPT;
__host__ __device__ inline uint32_t emit_blocks(uint32_t v_count) { return (v_count + EM_CHUNK - 1) / EM_CHUNK; }

__global__ __launch_bounds__(TPB) void k_emit_mark(Dims d, State st, uint8_t *__restrict__ mask, uint32_t *__restrict__ blk_cnt,
                                                   int want_free) {
  __shared__ uint32_t wsum[TPB / 64];
  const uint32_t t = blockIdx.x * TPB + threadIdx.x;
  const uint32_t lv0 = t * EM_VPT;
  uint32_t m = 0;
  if (lv0 < d.v_count) {
    const uint32_t n = d.v_count - lv0 < (uint32_t)EM_VPT ? d.v_count - lv0 : (uint32_t)EM_VPT;
    uint32_t cand = 0;
    if (n == EM_VPT) {
      const uint2 fw = *reinterpret_cast<const uint2 *>(st.vflag + lv0);  // (the array is 256-byte aligned, lv0 a multiple of 8)
#pragma unroll
      for (int u = 0; u < EM_VPT; ++u) {
        const uint32_t fl = ((u < 4 ? fw.x : fw.y) >> (8 * (u & 3))) & 0xffu;
        if ((fl & VR_MASK) != VR_UNOBSERVED) cand |= 1u << u;
      }
    } else {
      for (uint32_t u = 0; u < n; ++u)
        if ((st.vflag[lv0 + u] & VR_MASK) != VR_UNOBSERVED) cand |= 1u << u;
    }
    if (cand) {
      if (n == EM_VPT) {  // the thread's 64 bytes of result entries, occ = the top byte of an entry's second word
        const uint4 *rp = reinterpret_cast<const uint4 *>(st.res + lv0);
#pragma unroll
        for (int q = 0; q < EM_VPT / 2; ++q) {
          const uint4 r = rp[q];
          const int o0 = (int)r.y >> 24, o1 = (int)r.w >> 24;
          if (want_free ? o0 == 0 : o0 > 0) m |= 1u << (2 * q);
          if (want_free ? o1 == 0 : o1 > 0) m |= 1u << (2 * q + 1);
        }
        m &= cand;
      } else {
        for (uint32_t u = 0; u < n; ++u) {
          const int o = st.res[lv0 + u].occ;
          if (((cand >> u) & 1u) && (want_free ? o == 0 : o > 0)) m |= 1u << u;
        }
      }
    }
  }
  mask[t] = (uint8_t)m;
  uint32_t c = (uint32_t)__popc(m);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
#pragma unroll
    for (int w = 0; w < TPB / 64; ++w) tot += wsum[w];
    blk_cnt[blockIdx.x] = tot;
  }
}

// the list's length alone (statistics): the sum of the workgroup counts
constexpr int EM_SUM_TPB = 1024;
__global__ __launch_bounds__(EM_SUM_TPB) void k_emit_total(const uint32_t *__restrict__ blk_cnt, uint32_t *__restrict__ total, uint32_t n) {
  __shared__ uint32_t wtot[EM_SUM_TPB / 64];
  uint32_t acc = 0;
#pragma unroll 8
  for (uint32_t i = threadIdx.x; i < n; i += EM_SUM_TPB) acc += blk_cnt[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) wtot[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
#pragma unroll
    for (int w = 0; w < EM_SUM_TPB / 64; ++w) t += wtot[w];
    *total = t;
  }
}

// voxelIdxToGlobalFramePos: ring -> map index -> min corner (operations.h:940-983, 1022-1033)
__device__ __forceinline__ void emit_voxel_corner(const Dims &d, const Frame &f, uint32_t lv, float &x, float &y, float &z) {
  uint32_t v = d.v_begin + lv, rx, ry, rz;
  voxel_to_ring(d, v, rx, ry, rz);
  uint32_t mx = axis_correct((int)rx - f.eq[0], d.NX);
  uint32_t my = axis_correct((int)ry - f.eq[1], d.NY);
  uint32_t mz = axis_correct((int)rz - f.eq[2], d.NZ);
  x = (float)mx * d.voxel_size + d.pmin[0];
  y = (float)my * d.voxel_size + d.pmin[1];
  z = (float)mz * d.voxel_size + d.pmin[2];
  x += f.center[0];
  y += f.center[1];
  z += f.center[2];
}

struct EmitPlain {
  sdm_point *out;
  int mark_fov;
  __device__ __forceinline__ void operator()(const Dims &d, const Frame &f, const State &st, uint32_t lv, uint32_t o, float sub_x,
                                             float sub_y, float sub_z) const {
    float x, y, z;
    emit_voxel_corner(d, f, lv, x, y, z);
    sdm_voxel_result r = st.res[lv];
    sdm_point pt;
    pt.x = x - sub_x;
    pt.y = y - sub_y;
    pt.z = z - sub_z;
    pt.track = r.track;
    pt.label = r.label;
    pt.occ = r.occ;
    // semantic_dsp_map.h:1339-1342: the uncentred voxel position against the frame's frustum
    if (mark_fov && !point_in_frustum(d, f, x, y, z)) pt.occ = (int8_t)(pt.occ | SDM_OCC_OUT_OF_FOV);
    out[o] = pt;
  }
};

// The selected voxels of a workgroup's chunk go out in ascending order: the chunk's place in the list is the sum of the
// counts of the chunks before it - every workgroup that has something to write adds those up itself (a few thousand
// words that sit in L2, one round of loads; a scan kernel in between cost 8 us and a launch) - and the voxels of the chunk
// are ranked by the masks' popcounts.  The last workgroup also leaves the list's length.
template <typename Emit>
__global__ __launch_bounds__(TPB) void k_emit_write(Dims d, Frame f, State st, const uint8_t *__restrict__ mask,
                                                    const uint32_t *__restrict__ blk_cnt, uint32_t *__restrict__ total, uint32_t cap,
                                                    float sub_x, float sub_y, float sub_z, Emit emit) {
  __shared__ uint32_t wsum[TPB / 64], wbase[TPB / 64];
  const uint32_t my = blk_cnt[blockIdx.x];
  const bool last = blockIdx.x == gridDim.x - 1;
  if (my == 0 && !last) return;  // (workgroup-uniform)
  const uint32_t t = blockIdx.x * TPB + threadIdx.x;
  uint32_t m = my ? mask[t] : 0u;
  uint32_t acc = 0;
#pragma unroll 8
  for (uint32_t i = threadIdx.x; i < blockIdx.x; i += TPB) acc += blk_cnt[i];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  const uint32_t c = (uint32_t)__popc(m);
  uint32_t inc = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t v = __shfl_up(inc, off, 64);
    if (lane >= off) inc += v;
  }
  if (lane == 0) wbase[wid] = acc;
  if (lane == 63) wsum[wid] = inc;
  __syncthreads();
  uint32_t o = inc - c;
#pragma unroll
  for (int w = 0; w < TPB / 64; ++w) {
    o += wbase[w];
    if (w < wid) o += wsum[w];
  }
  if (last && threadIdx.x == TPB - 1) *total = o + c;
  while (m) {
    const uint32_t u = (uint32_t)__ffs(m) - 1u;
    m &= m - 1u;
    if (o < cap) emit(d, f, st, t * EM_VPT + u, o, sub_x, sub_y, sub_z);
    ++o;
  }
}

// ---- N2: colour rules + OpenCV's 8-bit RGB <-> HSV (published algorithm of OpenCV 4.x imgproc color_hsv: RGB2HSV_b with hsv_shift 12,
// HSV2RGB_b in float; the test-side restatement performs the same operations in the same order)
// RGB2HSV_b, hrange 180, hsv_shift 12
__device__ __forceinline__ void rgb2hsv_8u(const ColourTables &ct, int r, int g, int b, int &h, int &s, int &v) {
  v = max(max(b, g), r);
  const int vmin = min(min(b, g), r);
  const int diff = v - vmin;
  s = (diff * ct.sdiv[v] + (1 << 11)) >> 12;
  int hh = v == r ? g - b : (v == g ? b - r + 2 * diff : r - g + 4 * diff);
  hh = (hh * ct.hdiv180[diff] + (1 << 11)) >> 12;  // arithmetic shift
  hh += hh < 0 ? 180 : 0;
  h = min(max(hh, 0), 255);
}
// HSV2RGB_b -> HSV2RGB_native, float32
__device__ __forceinline__ void hsv2rgb_8u(int h, int s, int v, int &r, int &g, int &b) {
  const float sf = (float)s * (1.f / 255.f), vf = (float)v * (1.f / 255.f);
  float bf, gf, rf;
  if (s == 0) {
    bf = gf = rf = vf;
  } else {
    float hh = fmodf((float)h * (6.f / 180.f), 6.f);
    int sector = (int)floorf(hh);
    hh -= (float)sector;
    if ((unsigned)sector >= 6u) {
      sector = 0;
      hh = 0.f;
    }
    const float t0 = vf, t1 = vf * (1.f - sf), t2 = vf * (1.f - sf * hh), t3 = vf * (1.f - sf * (1.f - hh));
    const float tab[4] = {t0, t1, t2, t3};
    const int sec[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
    bf = tab[sec[sector][0]];
    gf = tab[sec[sector][1]];
    rf = tab[sec[sector][2]];
  }
  b = min(max(__float2int_rn(bf * 255.f), 0), 255);  // saturate_cast<uchar>: round half to even
  g = min(max(__float2int_rn(gf * 255.f), 0), 255);
  r = min(max(__float2int_rn(rf * 255.f), 0), 255);
}

struct EmitRgb {
  sdm_point_xyzrgb *out;
  const ColourTables *ctp;
  int want_free;
  __device__ __forceinline__ void operator()(const Dims &d, const Frame &f, const State &st, uint32_t lv, uint32_t o, float sub_x,
                                             float sub_y, float sub_z) const {
  const ColourTables &ct = *ctp;
  float x, y, z;
  emit_voxel_corner(d, f, lv, x, y, z);
  const sdm_voxel_result res = st.res[lv];
  sdm_point_xyzrgb pt;
  pt.x = x - sub_x;
  pt.y = y - sub_y;
  pt.z = z - sub_z;
  pt.one = 1.f;
  pt.a = 255;
  pt.pad[0] = pt.pad[1] = pt.pad[2] = 0;
  int r, g, b;
  if (want_free) {  // semantic_dsp_map.h:1371-1373
    r = 0;
    g = 255;
    b = 0;
  } else {
    const int label = res.label, track = res.track;
    if (res.occ != 1) {  // guessed occupied (:1325-1331)
      r = g = b = 255;
    } else if (label == ct.cfg.background_label) {  // :1277-1294
      const float src = ct.cfg.jet_axis == 0 ? -pt.z + 2.f : pt.y + 2.f;
      const int ci = min(max((int)(src * 51.2f), 0), 255);
      if (ci < 64) { r = 0; g = 0; b = ci * 4; }
      else if (ci < 128) { r = 0; g = (ci - 64) * 4; b = 255; }
      else if (ci < 192) { r = (ci - 128) * 4; g = 255; b = 255 - (ci - 128) * 4; }
      else { r = 255; g = 255 - (ci - 192) * 4; b = 0; }
      if (ct.cfg.evaluation_format) r = g = b = 0;
    } else if (track > d.max_movable || ct.cfg.colour_by_label) {  // :1297-1309
      b = ct.cfg.label_bgr[label][0];
      g = ct.cfg.label_bgr[label][1];
      r = ct.cfg.label_bgr[label][2];
    } else if (ct.cfg.evaluation_format) {  // :1311-1316
      r = label;
      g = track >> 8;
      b = track & 0xFF;
    } else {  // :1317-1319 (PINNED: the reference indexes its 256-entry table with the 16-bit track id)
      r = 160;
      g = ct.cfg.perm[track & 0xFF];
      b = ct.cfg.perm[label];
    }
    if (!ct.cfg.evaluation_format) {  // :1333-1351
      int h, s, vv;
      rgb2hsv_8u(ct, r, g, b, h, s, vv);
      if (!point_in_frustum(d, f, x, y, z)) vv = (int)((float)vv * 0.7f);  // uchar *= 0.7f
      hsv2rgb_8u(h, s, vv, r, g, b);
    }
  }
  pt.r = (uint8_t)r;
  pt.g = (uint8_t)g;
  pt.b = (uint8_t)b;
  out[o] = pt;
  }
};

inline unsigned blocks_for(size_t n, int tpb = TPB) { return (unsigned)((n + tpb - 1) / tpb); }

}  // namespace

// ============================================================================ launchers
// fresh = the buffers have never been written (sdm_create): everything to zero.  Otherwise the reference's clear()
// (operations.h:697-722) resets status, position, weight and time stamp of every slot and leaves track id, label and
// forget count of the dead slots as they were.
void launch_clear(const Dims &d, const State &st, hipStream_t s, bool fresh) {
  size_t n = (size_t)d.v_count * d.S;
  if (fresh) {
    hipMemsetAsync(st.pos4, 0, n * sizeof(float4), s);
    hipMemsetAsync(st.forget, 0, n, s);
    hipMemsetAsync(st.rec, 0, (size_t)d.v_count * rec_bytes(d.S), s);  // INVALID = 0 (the time particles' status is not stored)
    hipMemsetAsync(st.vts, 0, (size_t)d.v_count * sizeof(uint16_t), s);
    hipMemsetAsync(st.vflag, 0, (size_t)d.v_count, s);
    hipMemsetAsync(st.owner, 0xFF, n * sizeof(uint16_t), s);
    hipMemsetAsync(st.res, 0, (size_t)d.v_count * sizeof(sdm_voxel_result), s);
  } else {
    if (d.S == 8) hipLaunchKernelGGL(k_clear_map<8>, dim3(blocks_for(d.v_count, CLR_VOX)), dim3(TPB), 0, s, d, st);
    else if (d.S == 16) hipLaunchKernelGGL(k_clear_map<16>, dim3(blocks_for(d.v_count, CLR_VOX)), dim3(TPB), 0, s, d, st);
    else
      hipLaunchKernelGGL(k_clear_slots, dim3(blocks_for(n)), dim3(TPB), 0, s, d, st, n);
  }
  hipMemsetAsync(st.grp_hint, 0, grp_hint_bytes(d.v_count), s);  // nothing is dense any more
  hipMemsetAsync(st.alias, 0, 8, s);  // no older memberships (count and the sticky overflow word)
  hipMemsetAsync(st.alias_filter, 0, ALIAS_FILTER_WORDS * 4, s);
  hipMemsetAsync(st.owner_flag, 0, owner_flag_bytes(n), s);
  hipMemsetAsync(st.owner_flag2, 0, owner_flag2_bytes(n), s);
}

#define SDM_DISPATCH_S(kernel, grid, s, ...)                                                      \
  switch (d.p_n) {                                                                                 \
    case 1: hipLaunchKernelGGL(kernel<2>, grid, dim3(TPB), 0, s, __VA_ARGS__); break;              \
    case 2: hipLaunchKernelGGL(kernel<4>, grid, dim3(TPB), 0, s, __VA_ARGS__); break;              \
    case 3: hipLaunchKernelGGL(kernel<8>, grid, dim3(TPB), 0, s, __VA_ARGS__); break;              \
    default: hipLaunchKernelGGL(kernel<16>, grid, dim3(TPB), 0, s, __VA_ARGS__); break;            \
  }

// remark = the epoch a sweep marks the tiles with that it has to see again (non-incremental sweeps: given by the host,
// they also run outside frames; the in-frame sweep derives it from the frame block, which also works inside a graph)
size_t tile_mark_bytes(const Dims &d) {
  const size_t n_tiles = blocks_for(d.v_count, OCC_TILE);
  return (std::max<size_t>(n_tiles, (size_t)TPB * OCC_SEG_MAX) + 16 + 15) / 16 * 16;  // one of the two arrays
}
void launch_occupancy(const Dims &d, const Filter &flt, const State &st, Counters *cnt, int all_dirty, const FrameArgs *fa, uint32_t remark,
                      hipStream_t s, int mode) {
  dim3 grid(blocks_for(d.v_count, OCC_TILE));
  const int lists = mode & OCC_LISTS;
  if (all_dirty && (mode & OCC_SKIP_SCAN)) {
    SDM_DISPATCH_S(k_occupancy_dense, grid, s, d, flt.occ_threshold, st, cnt, st.occ_need, remark, grid.x | 0x80000000u);
  } else if (all_dirty) {
    // (Measured and not kept, round 5: the map cut into 2 / 4 / 8 slices of tiles, the scans of slices 1.. on a second
    // stream next to the dense launches of the slices before them - every cross-stream event costs more than the slice
    // of classification it hides: 0.285 -> 0.297 / 0.318 / 0.354 ms on the dense case.)
    // lists: the tiles' sparse voxels go through a launch of their own (k_occupancy_scan says when that pays).  (Measured
    // and not kept: the listed units as extra workgroups of the last launch, the last of them to finish wrapping up - one
    // launch less, and 55 us against 47 us on an empty map: that kernel then holds 126 registers and its workgroups' LDS
    // whatever a workgroup does; the lists worked off by the last launch's own workgroup of the tile - 67 us against 56
    // on the `driven` map, four resident workgroups per CU.)
    const dim3 sgrid = grid;
    if (lists) {
      SDM_DISPATCH_S(k_occupancy_scan_lists, sgrid, s, d, flt.occ_threshold, st, cnt, st.occ_need, grid.x, remark);
      SDM_DISPATCH_S(k_occupancy_listed, dim3(OCC_LISTED_GRID), s, d, flt.occ_threshold, st, remark);
    } else {
      SDM_DISPATCH_S(k_occupancy_scan, sgrid, s, d, flt.occ_threshold, st, cnt, st.occ_need, grid.x, remark);
    }
    SDM_DISPATCH_S(k_occupancy_dense, grid, s, d, flt.occ_threshold, st, cnt, st.occ_need, remark, grid.x);
  } else {
    const uint32_t n_tiles = grid.x;
    // marks per thread of the tile scan, in whole 16-byte loads; 0: too many tiles for one workgroup to scan
    const uint32_t seg = n_tiles <= TPB * OCC_SEG_MAX ? ((n_tiles + TPB - 1) / TPB + 15u) / 16u * 16u : 0u;
    if (seg) grid = dim3(std::min<uint32_t>(OCC_GRID, n_tiles));
    SDM_DISPATCH_S(k_occupancy, grid, s, d, flt.occ_threshold, st, cnt, fa, n_tiles, seg);
  }
}


void launch_set_frame(FrameArgs *fa_dev, const FrameArgs &fa, hipStream_t s) {
  hipLaunchKernelGGL(k_set_frame, dim3(1), dim3(TPB), 0, s, fa_dev, fa);
}
const void *set_frame_kernel() { return reinterpret_cast<const void *>(k_set_frame); }

// The frustum reach set depends on the camera pose only, not on the map: it runs on a side stream next to the
// object moves.  Grids and the flood's LDS size are functions of the map dimensions only (the kernels stride over the
// frame's box), so that the launch sequence of a frame is the same every frame (hipGraph).
void launch_frustum(const Dims &d, const Scratch &sc, hipStream_t s) {
  static bool lds_attr_set = false;
  if (!lds_attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_flood2d), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
    lds_attr_set = true;
  }
  const size_t max_words = (size_t)(d.NY + 1) * (d.NZ + 1) * sc.wpl;  // 64-vertex words of the whole vertex grid
  const unsigned g_mask = (unsigned)std::min<size_t>(blocks_for(max_words * 64), 4096);
  const unsigned g_line = (unsigned)std::min<size_t>(blocks_for((size_t)sc.wy * (d.NZ + 1) * 64), 256);
  hipLaunchKernelGGL(k_vertex_mask, dim3(g_mask), dim3(TPB), 0, s, d, sc.fa_side, sc.vmask, sc.wpl, sc.cnt);
  hipLaunchKernelGGL(k_line_info, dim3(g_line), dim3(TPB), 0, s, d, sc.fa_side, sc.vmask, sc.wpl, sc.wy, sc.line_ne, sc.line_ey, sc.line_ez,
                     sc.cnt);
  hipLaunchKernelGGL(k_flood2d, dim3(1), dim3(TPB), (size_t)(d.NZ + 1) * sc.wy * 8 * 4, s, d, sc.fa_side, sc.vmask, sc.wpl, sc.wy, sc.line_ey,
                     sc.line_ez, sc.line_reach, sc.cnt);
  hipLaunchKernelGGL(k_flood_generic, dim3(1), dim3(1024), 0, s, d, sc.fa_side, sc.vmask, sc.reach, sc.wpl, sc.cnt);
}

void launch_visibility(const Dims &d, const Filter &flt, const State &st, const Scratch &sc, float *ck_out, int finish, hipStream_t s,
                       hipEvent_t vis_done) {
  {
    const size_t max_words = (size_t)((d.NX + 63) / 64 + 1) * d.NY * d.NZ;
    dim3 grid((unsigned)std::min<size_t>(blocks_for(max_words, VIS_WORDS), 2048));
    if (vis_done) {
      switch (d.p_n) {
        case 1: hipExtLaunchKernelGGL(k_visibility<2>, grid, dim3(TPB), 0, s, nullptr, vis_done, 0, d, st, sc); break;
        case 2: hipExtLaunchKernelGGL(k_visibility<4>, grid, dim3(TPB), 0, s, nullptr, vis_done, 0, d, st, sc); break;
        case 3: hipExtLaunchKernelGGL(k_visibility<8>, grid, dim3(TPB), 0, s, nullptr, vis_done, 0, d, st, sc); break;
        default: hipExtLaunchKernelGGL(k_visibility<16>, grid, dim3(TPB), 0, s, nullptr, vis_done, 0, d, st, sc); break;
      }
    } else {
      SDM_DISPATCH_S(k_visibility, grid, s, d, st, sc);
    }
  }
  // bins: one workgroup per image row lays the row's bins out, fills and orders them; then the pixels are classified for pass 1
  if (d.W <= 2 * BR_TPB) hipLaunchKernelGGL(k_bin_rows<2>, dim3((unsigned)d.H), dim3(BR_TPB), 0, s, d, st, sc);
  else if (d.W <= 4 * BR_TPB) hipLaunchKernelGGL(k_bin_rows<4>, dim3((unsigned)d.H), dim3(BR_TPB), 0, s, d, st, sc);
  else hipLaunchKernelGGL(k_bin_rows<8>, dim3((unsigned)d.H), dim3(BR_TPB), 0, s, d, st, sc);  // (up to 4095 columns: round 6)
  hipLaunchKernelGGL(k_ck_classify, dim3(blocks_for((size_t)d.W * d.H)), dim3(TPB), 0, s, d, flt, sc, ck_out, finish);
}

void launch_ck(const Dims &d, const Filter &flt, const State &st, const Scratch &sc, float *ck_out, int finish, hipStream_t s) {
  hipLaunchKernelGGL(k_ck, dim3(CK_HEAVY_BLOCKS + blocks_for((size_t)d.W * d.H, A7_ROWS * A7_ITEMS)), dim3(A7_ROWS, A7_ITEMS), 0, s, d, flt, st, sc,
                     ck_out, finish);
}
void launch_ck_finish(const Dims &d, const Filter &flt, const Scratch &sc, const float *parts, int n_parts, size_t part_stride, hipStream_t s) {
  hipLaunchKernelGGL(k_ck_finish, dim3(blocks_for((size_t)d.W * d.H)), dim3(TPB), 0, s, d, flt, sc, parts, n_parts,
                     part_stride ? part_stride : (size_t)d.W * d.H);
}
void launch_ck_reduce_chunk(const float *stage, const float *own_part, float *full, uint32_t chunk, int world, int rank, hipStream_t s) {
  hipLaunchKernelGGL(k_ck_reduce_chunk, dim3((chunk + TPB - 1) / TPB), dim3(TPB), 0, s, stage, own_part, full, chunk, world, rank);
}
void launch_weight(const Dims &d, const Filter &flt, const State &st, const Scratch &sc, hipStream_t s, const float *ck_raw) {
  // four particles per wave while a window fits two rounds of 64 lanes (window_half <= 5), two beyond
  const int side = 2 * d.window_half + 1;
#define SDM_WEIGHT(UU, RR)                                                                                              \
  if (ck_raw) hipLaunchKernelGGL((k_weight<UU, RR, true>), dim3(WT_GRID), dim3(64 * WT_WAVES), 0, s, d, flt, st, sc, ck_raw); \
  else hipLaunchKernelGGL((k_weight<UU, RR, false>), dim3(WT_GRID), dim3(64 * WT_WAVES), 0, s, d, flt, st, sc, ck_raw);
  if (side * side <= 64) {
    SDM_WEIGHT(4, 1)
  } else if (side * side <= 128) {
    SDM_WEIGHT(4, 2)
  } else {
    SDM_WEIGHT(2, 4)
  }
#undef SDM_WEIGHT
}

// Birth candidates and their stable sort by target voxel depend on the input cloud only: side stream.
// Returns which double buffer holds the sorted list.
int launch_birth_prepare(const Dims &d, const Filter &flt, const BirthOrder &bo, const State &st,
                         const Scratch &sc, hipStream_t s) {
  const size_t hw = (size_t)d.W * d.H;
  const size_t total = hw * flt.nb;
  if (flt.use_rng) {  // the exclusive rank among valid pixels only feeds the noise-table cursor
    hipLaunchKernelGGL(k_birth_flags, dim3(blocks_for(hw)), dim3(TPB), 0, s, d, bo, sc);
    exclusive_scan_u32(sc.b_valid, sc.b_rank, hw, sc.scan_scratch_b, s);
  }
  hipLaunchKernelGGL(k_birth_candidates, dim3(blocks_for(total)), dim3(TPB), 0, s, d, flt, bo, st, sc);
  if (flt.use_rng) hipLaunchKernelGGL(k_birth_cursor, dim3(1), dim3(64), 0, s, d, flt, sc);
  int nbits = d.x_n + d.y_n + d.z_n + 1;
  return radix_sort_pairs(sc.bkey_a, sc.bval_a, sc.bkey_b, sc.bval_b, total, nbits, sc.sort_scratch, s);
}

// literal: walk every segment candidate by candidate (frames whose 16-bit time stamp has wrapped, see above)
void launch_birth_replay(const Dims &d, const Filter &flt, const State &st, const Scratch &sc, int which, bool literal,
                         hipStream_t s) {
  const size_t total = (size_t)d.W * d.H * flt.nb;
  const uint32_t *skey = which ? sc.bkey_b : sc.bkey_a;
  const uint32_t *sval = which ? sc.bval_b : sc.bval_a;
  dim3 grid(blocks_for(total));
#define SDM_BIRTH(SS)                                                                                                  \
  if (literal) hipLaunchKernelGGL((k_birth_replay<SS, true>), grid, dim3(TPB), 0, s, d, flt, st, sc, skey, sval, (uint32_t)total); \
  else hipLaunchKernelGGL((k_birth_replay<SS, false>), grid, dim3(TPB), 0, s, d, flt, st, sc, skey, sval, (uint32_t)total);
  switch (d.p_n) {
    case 1: SDM_BIRTH(2) break;
    case 2: SDM_BIRTH(4) break;
    case 3: SDM_BIRTH(8) break;
    default: SDM_BIRTH(16) break;
  }
#undef SDM_BIRTH
}

void launch_labeled_cloud(const Dims &d, const CloudArgsHost &h, const float *depth, const uint8_t *static_mask,
                          const uint16_t *label_to_inst, const uint8_t *obj_masks, const double *bbox,
                          sdm_labeled_point *cloud, hipStream_t s) {
  CloudArgs a;
  static_assert(sizeof(CloudArgs) == sizeof(CloudArgsHost), "CloudArgs layout");
  __builtin_memcpy(&a, &h, sizeof(a));
  hipLaunchKernelGGL(k_labeled_cloud, dim3(blocks_for((size_t)d.W * d.H)), dim3(TPB), 0, s, d, a, depth, static_mask, label_to_inst,
                     obj_masks, bbox, cloud);
}

void launch_manual_resize(const Dims &d, const void *src, void *dst, int src_w, int src_h, float scale, int elem_bytes,
                          hipStream_t s) {
  const float scale_inv = 1.f / scale;  // pointcloud_tools.h:1119
  dim3 grid(blocks_for((size_t)d.W * d.H));
  if (elem_bytes == 4)
    hipLaunchKernelGGL(k_manual_resize<float>, grid, dim3(TPB), 0, s, (const float *)src, (float *)dst, src_w, src_h, d.W, d.H, scale_inv);
  else
    hipLaunchKernelGGL(k_manual_resize<uint8_t>, grid, dim3(TPB), 0, s, (const uint8_t *)src, (uint8_t *)dst, src_w, src_h, d.W, d.H,
                       scale_inv);
}

// dense slot-order arrays <-> per-voxel records (state export / import)
__global__ __launch_bounds__(TPB) void k_rec_pack(Dims d, State st, const float *__restrict__ w, const uint16_t *__restrict__ ts,
                                                  const uint16_t *__restrict__ track, const uint8_t *__restrict__ label,
                                                  const uint8_t *__restrict__ status, size_t n) {
  size_t li = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (li >= n) return;
  if ((li & (d.S - 1)) == 0) {  // the time particle (buffer.h:57-79): its stamp is the voxel's observation stamp, nothing else of it is kept
    st.vts[li >> d.p_n] = ts[li];
    return;
  }
  const SlotRef sr = slot_ref_li(st, d.p_n, li);
  sr.set_status(status[li]);
  sr.set_w(w[li]);
  sr.set_ts(ts[li]);
  sr.set_track(track[li]);
  sr.set_label(label[li]);
}
__global__ __launch_bounds__(TPB) void k_rec_unpack(Dims d, State st, float *__restrict__ w, uint16_t *__restrict__ ts,
                                                    uint16_t *__restrict__ track, uint8_t *__restrict__ label,
                                                    uint8_t *__restrict__ status, size_t n) {
  size_t li = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (li >= n) return;
  if ((li & (d.S - 1)) == 0) {  // slot 0 as the reference holds it: TIMEPTC, the voxel's observation stamp, zeros
    status[li] = ST_TIMEPTC;
    w[li] = 0.f;
    ts[li] = st.vts[li >> d.p_n];
    track[li] = 0;
    label[li] = 0;
    return;
  }
  const SlotRef sr = slot_ref_li(st, d.p_n, li);
  status[li] = sr.status();
  w[li] = sr.w();
  ts[li] = sr.ts();
  track[li] = sr.track();
  label[li] = sr.label();
}

void launch_rec_pack(const Dims &d, const State &st, const float *w, const uint16_t *ts, const uint16_t *track,
                     const uint8_t *label, const uint8_t *status, hipStream_t s) {
  const size_t n = (size_t)d.v_count * d.S;
  hipLaunchKernelGGL(k_rec_pack, dim3(blocks_for(n)), dim3(TPB), 0, s, d, st, w, ts, track, label, status, n);
}
void launch_rec_unpack(const Dims &d, const State &st, float *w, uint16_t *ts, uint16_t *track, uint8_t *label, uint8_t *status,
                       hipStream_t s) {
  const size_t n = (size_t)d.v_count * d.S;
  hipLaunchKernelGGL(k_rec_unpack, dim3(blocks_for(n)), dim3(TPB), 0, s, d, st, w, ts, track, label, status, n);
}

// bench hook: every slot of every voxel holds a live particle with pseudo-random weight / track / label, every voxel is
// observed - the dense case of SURVEY.md 8(d) for the occupancy sweep.  mode 0: every SLOT draws one of eight track ids
// (four to five different ones per voxel: the worst case for the vote); mode 1 ("surface"): every VOXEL draws one, as in
// a real map, where a voxel holds particles of one surface - except one voxel in 16, whose slots are split between two
// (object borders).
__global__ __launch_bounds__(TPB) void k_fill_dense(Dims d, State st, uint32_t stamp, int mode) {
  size_t li = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)d.v_count * d.S;
  if (li >= n) return;
  const uint32_t i = (uint32_t)(li & (d.S - 1));
  const size_t lv = li >> d.p_n;
  uint32_t h = (uint32_t)li * 2654435761u;
  h ^= h >> 15;
  h *= 2246822519u;
  h ^= h >> 13;
  if (i) {
    const SlotRef sr = slot_ref_li(st, d.p_n, li);
    sr.set_status((uint8_t)((h & 7u) == 0 ? ST_REGULAR_BORN : ST_UPDATED));
    sr.set_w(0.06f + (float)(h >> 20) * (0.3f / 4096.f));
    sr.set_ts((uint16_t)stamp);
  }
  uint32_t pick = (h >> 8) & 7u;
  if (mode == 1) {
    uint32_t hv = (uint32_t)lv * 2654435761u;
    hv ^= hv >> 15;
    hv *= 2246822519u;
    hv ^= hv >> 13;
    pick = (hv >> 8) & 7u;
    if (((hv >> 4) & 15u) == 0 && (i & 1u)) pick = (pick + 1u) & 7u;  // a border voxel: two tracks
  }
  if (i) {
    const SlotRef sr = slot_ref_li(st, d.p_n, li);
    sr.set_track((uint16_t)(65524u + pick));
    sr.set_label((uint8_t)(5u + pick));
  }
  st.owner[li] = OWNER_NONE;
  if (i == 0) {
    st.vts[lv] = (uint16_t)stamp;
    st.vflag[lv] = VF_DIRTY;
  }
}
#ifdef SDM_AB_TIMERS
void debug_timers(unsigned long long *out32, int reset) {
  (void)hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_dbg), sizeof(g_dbg));
  if (reset) {
    static unsigned long long z[6][8192 * 4];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), z, sizeof(z));
  }
}
#endif
void launch_fill_dense(const Dims &d, const State &st, uint32_t stamp, int mode, hipStream_t s) {
  const size_t n = (size_t)d.v_count * d.S;
  hipLaunchKernelGGL(k_fill_dense, dim3(blocks_for(n)), dim3(TPB), 0, s, d, st, stamp, mode);
}

void launch_vflag_from_records(const Dims &d, const State &st, hipStream_t s) {
  hipLaunchKernelGGL(k_vflag_from_records, dim3(blocks_for(d.v_count)), dim3(TPB), 0, s, d, st);
}

void launch_count_live(const Dims &d, const State &st, unsigned long long *out, hipStream_t s) {
  hipMemsetAsync(out, 0, 8, s);
  hipLaunchKernelGGL(k_count_live, dim3(blocks_for(d.v_count)), dim3(TPB), 0, s, d, st, out);
}
void launch_count_owner(const Dims &d, const State &st, uint16_t track, unsigned long long *out, hipStream_t s) {
  hipMemsetAsync(out, 0, 8, s);
  hipLaunchKernelGGL(k_count_owner, dim3(2048), dim3(TPB), 0, s, d, st, track, out);
}
void launch_pack_pos4(float4 *pos4, uint8_t *forget_plane, const float *px, const float *py, const float *pz, const uint8_t *forget, size_t n,
                      hipStream_t s) {
  hipLaunchKernelGGL(k_pack_pos4, dim3(blocks_for(n)), dim3(TPB), 0, s, pos4, forget_plane, px, py, pz, forget, n);
}
void launch_unpack_pos4(const float4 *pos4, const uint8_t *forget_plane, float *px, float *py, float *pz, uint8_t *forget, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_unpack_pos4, dim3(blocks_for(n)), dim3(TPB), 0, s, pos4, forget_plane, px, py, pz, forget, n);
}
// the list's length only
void launch_emit_count(const Dims &d, const State &st, const EmitScratch &e, int want_free, hipStream_t s) {
  const uint32_t nb = emit_blocks(d.v_count);
  hipLaunchKernelGGL(k_emit_mark, dim3(nb), dim3(TPB), 0, s, d, st, e.mask, e.blk_cnt, want_free);
  hipLaunchKernelGGL(k_emit_total, dim3(1), dim3(EM_SUM_TPB), 0, s, e.blk_cnt, e.total, nb);
}
size_t emit_mask_bytes(const Dims &d) { return (size_t)emit_blocks(d.v_count) * TPB; }
size_t emit_block_elems(const Dims &d) { return (size_t)emit_blocks(d.v_count); }
void launch_emit_points_rgb(const Dims &d, const Frame &f, const State &st, const ColourTables *ct, const EmitScratch &e,
                            sdm_point_xyzrgb *out, uint32_t cap, int want_free, const float sub[3], hipStream_t s) {
  const uint32_t nb = emit_blocks(d.v_count);
  hipLaunchKernelGGL(k_emit_mark, dim3(nb), dim3(TPB), 0, s, d, st, e.mask, e.blk_cnt, want_free);
  hipLaunchKernelGGL(k_emit_write<EmitRgb>, dim3(nb), dim3(TPB), 0, s, d, f, st, e.mask, e.blk_cnt, e.total, cap, sub[0], sub[1], sub[2],
                     EmitRgb{out, ct, want_free});
}
void launch_emit_points(const Dims &d, const Frame &f, const State &st, const EmitScratch &e, sdm_point *out, uint32_t cap, int want_free,
                        const float sub[3], int mark_fov, hipStream_t s) {
  const uint32_t nb = emit_blocks(d.v_count);
  hipLaunchKernelGGL(k_emit_mark, dim3(nb), dim3(TPB), 0, s, d, st, e.mask, e.blk_cnt, want_free);
  hipLaunchKernelGGL(k_emit_write<EmitPlain>, dim3(nb), dim3(TPB), 0, s, d, f, st, e.mask, e.blk_cnt, e.total, cap, sub[0], sub[1], sub[2],
                     EmitPlain{out, mark_fov});
}

}  // namespace sdm
