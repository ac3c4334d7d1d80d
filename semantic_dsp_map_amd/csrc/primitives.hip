// primitives.hip — hand-written gfx950 building blocks: exclusive scan and stable LSD radix sort.
//
// Both are written for 64-wide wavefronts (ballot masks are 64 bit, wave reductions
// use 64-lane shuffles) and are deterministic: no atomics decide an output position.
// The sort is what turns the reference's strictly sequential "first vacant slot"
// insertion order (mc_ring/operations.h:790-796, semantic_dsp_map.h:778-800) into
// per-voxel segments that keep their original order (stability) and can be replayed
// voxel-parallel.
#include "sdm_internal.h"

namespace sdm {

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 2048

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t n = __shfl_up(v, off, 64);
    if (lane >= off) v += n;
  }
  return v;
}

// exclusive scan across the 256 threads of a block of one value per thread; returns the block total in *total
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *total) {
  __shared__ uint32_t wave_sums[SCAN_THREADS / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t inc = wave_inclusive_scan(v);
  if (lane == 63) wave_sums[wid] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < SCAN_THREADS / 64; ++i) {
    uint32_t s = wave_sums[i];
    if (i < wid) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// pass 1: per-tile totals
__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_sums(const uint32_t *__restrict__ in, uint32_t *__restrict__ sums,
                                                               size_t n, const uint32_t *__restrict__ n_dev) {
  if (n_dev) n = *n_dev < n ? (size_t)*n_dev : n;  // tiles beyond the count sum to zero
  size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    size_t k = base + i;
    if (k < n) s += in[k];
  }
  uint32_t total;
  block_exclusive_scan(s, &total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// pass 2: every block first reduces the totals of all preceding tiles (a few thousand values at most, read
// by 256 threads), then scans its own tile.  Two launches per scan instead of three.
__global__ __launch_bounds__(SCAN_THREADS) void scan_tiles(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                           const uint32_t *__restrict__ sums, size_t n,
                                                           const uint32_t *__restrict__ n_dev) {
  if (n_dev) n = *n_dev < n ? (size_t)*n_dev : n;
  if ((size_t)blockIdx.x * SCAN_TILE >= n) return;
  uint32_t pre = 0;
  for (uint32_t t = threadIdx.x; t < blockIdx.x; t += SCAN_THREADS) pre += sums[t];
  uint32_t tile_offset;
  block_exclusive_scan(pre, &tile_offset);
  size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    size_t k = base + i;
    v[i] = k < n ? in[k] : 0u;
    s += v[i];
  }
  uint32_t total;
  uint32_t ex = block_exclusive_scan(s, &total) + tile_offset;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    size_t k = base + i;
    if (k < n) out[k] = ex;
    ex += v[i];
  }
}

// One-launch scan for the sizes a frame needs (a few hundred tiles at most): every tile publishes its total as one
// 8-byte word {flag, total} (one agent-scope atomic store: flag and value cannot be seen apart), then wave 0 of the tile
// reads the words of ALL its predecessors, 64 at a time, waiting for each to appear, and adds them up - no chain from tile
// to tile, so a tile's latency is its own reduction plus one round trip.  Tile ids are handed out by a ticket counter in
// start order: a tile only ever waits for tiles that started before it.  The last tile to finish clears the words and
// the counters for the next scan on this scratch buffer.  Integer adds: the result does not depend on the order.
constexpr int SCAN_ONEPASS_MAX_TILES = 512;

__global__ __launch_bounds__(SCAN_THREADS) void scan_onepass(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, size_t n,
                                                             const uint32_t *__restrict__ n_dev,
                                                             unsigned long long *__restrict__ desc, uint32_t *__restrict__ counters) {
  __shared__ uint32_t s_tile, s_prefix, s_last;
  if (n_dev) n = *n_dev < n ? (size_t)*n_dev : n;
  if (threadIdx.x == 0) s_tile = atomicAdd(&counters[0], 1u);
  __syncthreads();
  const uint32_t tile = s_tile;
  const size_t base = (size_t)tile * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    const size_t k = base + i;
    v[i] = k < n ? in[k] : 0u;
    s += v[i];
  }
  uint32_t total;
  uint32_t ex = block_exclusive_scan(s, &total);
  if (threadIdx.x == 0) __hip_atomic_store(&desc[tile], (1ull << 32) | (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x < 64) {
    uint32_t run = 0;
    for (uint32_t b = 0; b < tile; b += 64) {
      const uint32_t idx = b + threadIdx.x;
      if (idx < tile) {
        unsigned long long d = __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while ((d >> 32) == 0ull) {
          __builtin_amdgcn_s_sleep(1);
          d = __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        run += (uint32_t)d;
      }
    }
    for (int off = 32; off > 0; off >>= 1) run += __shfl_down(run, off, 64);
    if (threadIdx.x == 0) s_prefix = run;
  }
  __syncthreads();
  ex += s_prefix;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    const size_t k = base + i;
    if (k < n) out[k] = ex;
    ex += v[i];
  }
  // the last tile through clears the scratch for the next scan (every tile has finished reading by then)
  if (threadIdx.x == 0) s_last = atomicAdd(&counters[1], 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (s_last) {
    for (uint32_t t = threadIdx.x; t < gridDim.x; t += SCAN_THREADS) __hip_atomic_store(&desc[t], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) {
      __hip_atomic_store(&counters[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&counters[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---------------------------------------------------------------- radix sort
constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;  // 2048 keys per block
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_BITS = 9;                      // 25-bit voxel keys sort in 3 passes
constexpr int RS_RADIX = 1 << RS_BITS;          // 512
constexpr int RS_DPT = RS_RADIX / RS_THREADS;   // digits per thread when a block walks the digit table

// histogram of one digit per tile; layout hist[digit * n_tiles + tile] so that one exclusive scan over the
// whole array yields the global scatter offsets.
__global__ __launch_bounds__(RS_THREADS) void rs_histogram(const uint32_t *__restrict__ keys, uint32_t *__restrict__ hist,
                                                           size_t n, int shift, uint32_t n_tiles,
                                                           const uint32_t *__restrict__ n_dev) {
  __shared__ uint32_t h[RS_RADIX];
  if (n_dev) n = *n_dev < n ? (size_t)*n_dev : n;
#pragma unroll
  for (int q = 0; q < RS_DPT; ++q) h[threadIdx.x + q * RS_THREADS] = 0;
  __syncthreads();
  size_t base = (size_t)blockIdx.x * RS_TILE;
#pragma unroll
  for (int r = 0; r < RS_ITEMS; ++r) {
    size_t k = base + (size_t)r * RS_THREADS + threadIdx.x;
    if (k < n) atomicAdd(&h[(keys[k] >> shift) & (RS_RADIX - 1)], 1u);
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < RS_DPT; ++q) {
    int dg = threadIdx.x + q * RS_THREADS;
    hist[(size_t)dg * n_tiles + blockIdx.x] = h[dg];
  }
}

// stable scatter: elements of a tile are visited in index order (round-major, then thread);
// the rank of an element among equal digits is (earlier rounds) + (earlier waves of this round)
// + (lower lanes of this wave), the last from a ballot-built peer mask.
__global__ __launch_bounds__(RS_THREADS) void rs_scatter(const uint32_t *__restrict__ keys_in,
                                                         const uint32_t *__restrict__ vals_in,
                                                         uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                                                         const uint32_t *__restrict__ offsets, size_t n, int shift,
                                                         uint32_t n_tiles, const uint32_t *__restrict__ n_dev) {
  __shared__ uint32_t digit_base[RS_RADIX];
  __shared__ uint32_t wave_cnt[RS_WAVES][RS_RADIX];
  if (n_dev) n = *n_dev < n ? (size_t)*n_dev : n;
  if ((size_t)blockIdx.x * RS_TILE >= n) return;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < RS_DPT; ++q) {
    int dg = threadIdx.x + q * RS_THREADS;
    digit_base[dg] = offsets[(size_t)dg * n_tiles + blockIdx.x];
  }
  size_t base = (size_t)blockIdx.x * RS_TILE;
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int r = 0; r < RS_ITEMS; ++r) {
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w)
#pragma unroll
      for (int q = 0; q < RS_DPT; ++q) wave_cnt[w][threadIdx.x + q * RS_THREADS] = 0;
    __syncthreads();
    size_t k = base + (size_t)r * RS_THREADS + threadIdx.x;
    bool valid = k < n;
    uint32_t key = valid ? keys_in[k] : 0u;
    uint32_t val = valid ? vals_in[k] : 0u;
    uint32_t digit = (key >> shift) & (RS_RADIX - 1);
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < RS_BITS; ++b) {
      bool bit = (digit >> b) & 1u;
      uint64_t m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    uint32_t rank_in_wave = (uint32_t)__popcll(peers & lt_mask);
    if (valid && rank_in_wave == 0) wave_cnt[wid][digit] = (uint32_t)__popcll(peers);
    __syncthreads();
    if (valid) {
      uint32_t off = digit_base[digit] + rank_in_wave;
#pragma unroll
      for (int w = 0; w < RS_WAVES; ++w)
        if (w < wid) off += wave_cnt[w][digit];
      keys_out[off] = key;
      vals_out[off] = val;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RS_DPT; ++q) {
      int dg = threadIdx.x + q * RS_THREADS;
      uint32_t add = 0;
#pragma unroll
      for (int w = 0; w < RS_WAVES; ++w) add += wave_cnt[w][dg];
      digit_base[dg] += add;
    }
    __syncthreads();
  }
}

}  // namespace

// the scratch holds either the tile totals of the two-launch scan or, for at most SCAN_ONEPASS_MAX_TILES tiles, the
// 8-byte words + two counters of the one-launch scan, which must be ZERO before the first use (they clean up after
// themselves)
size_t scan_scratch_elems(size_t n) {
  const size_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  return tiles <= (size_t)SCAN_ONEPASS_MAX_TILES ? 2 * tiles + 8 : tiles + 1;
}

void exclusive_scan_u32(const uint32_t *in, uint32_t *out, size_t n, uint32_t *scratch, hipStream_t s, const uint32_t *n_dev) {
  if (n == 0) return;
  size_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (tiles <= (size_t)SCAN_ONEPASS_MAX_TILES) {
    // two counters first, the words behind them (8-byte aligned): the layout does not depend on n, so scans of different
    // sizes may share a scratch buffer - each leaves the part it used zeroed
    uint32_t *counters = scratch;
    unsigned long long *desc = reinterpret_cast<unsigned long long *>(scratch + 2);
    hipLaunchKernelGGL(scan_onepass, dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, s, in, out, n, n_dev, desc, counters);
    return;
  }
  hipLaunchKernelGGL(scan_tile_sums, dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, s, in, scratch, n, n_dev);
  hipLaunchKernelGGL(scan_tiles, dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, s, in, out, scratch, n, n_dev);
}

// sort scratch: [one-launch scan region, fixed size | digit histogram | tile totals of a two-launch scan, if the
// histogram is that long].  The scan region sits at a fixed place so that it stays zero whatever n is sorted next.
constexpr size_t RS_SCAN_REGION = 2 * (size_t)SCAN_ONEPASS_MAX_TILES + 8;
size_t sort_scratch_elems(size_t n) {
  size_t tiles = (n + RS_TILE - 1) / RS_TILE;
  size_t hist = (size_t)RS_RADIX * tiles;
  size_t scan_tiles = (hist + SCAN_TILE - 1) / SCAN_TILE;
  return RS_SCAN_REGION + hist + (scan_tiles > (size_t)SCAN_ONEPASS_MAX_TILES ? scan_tiles + 1 : 0);
}

int radix_sort_pairs(uint32_t *keys_a, uint32_t *vals_a, uint32_t *keys_b, uint32_t *vals_b, size_t n, int nbits,
                     uint32_t *scratch, hipStream_t s, const uint32_t *n_dev) {
  if (n == 0) return 0;
  size_t tiles = (n + RS_TILE - 1) / RS_TILE;
  size_t hist_n = (size_t)RS_RADIX * tiles;
  uint32_t *hist = scratch + RS_SCAN_REGION;
  const size_t scan_tiles = (hist_n + SCAN_TILE - 1) / SCAN_TILE;
  uint32_t *scan_scratch = scan_tiles > (size_t)SCAN_ONEPASS_MAX_TILES ? hist + hist_n : scratch;
  int which = 0;
  for (int shift = 0; shift < nbits; shift += RS_BITS) {
    uint32_t *kin = which ? keys_b : keys_a, *vin = which ? vals_b : vals_a;
    uint32_t *kout = which ? keys_a : keys_b, *vout = which ? vals_a : vals_b;
    hipLaunchKernelGGL(rs_histogram, dim3((unsigned)tiles), dim3(RS_THREADS), 0, s, kin, hist, n, shift, (uint32_t)tiles, n_dev);
    exclusive_scan_u32(hist, hist, hist_n, scan_scratch, s, nullptr);
    hipLaunchKernelGGL(rs_scatter, dim3((unsigned)tiles), dim3(RS_THREADS), 0, s, kin, vin, kout, vout, hist, n, shift,
                       (uint32_t)tiles, n_dev);
    which ^= 1;
  }
  return which;
}

}  // namespace sdm
