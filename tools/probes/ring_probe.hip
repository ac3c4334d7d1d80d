// Measurement probe (not part of the library): what a streaming SKELETON for the dense case of the occupancy sweep can
// reach on this machine - 80 B of records per voxel read, 8 B of result + 1 B of flag written, a chunk = 64 voxels =
// 5 KB of records - before any evaluation is put into it.  Variants:
//   copy      plain 16-byte copy of the same byte count (the guide's reference point, MI355X_MICROARCH.md:35)
//   regs      today's shape: one workgroup per 2048 voxels, a wave walks 8 chunks, records land in registers
//             (lane-linear 16-byte non-temporal loads), pass through LDS, every lane reads its 80-byte record
//   ring<R>   persistent waves; records land in a per-wave LDS ring by LDS-DMA (global_load_lds_dwordx4), R slots,
//             R-1 chunks in flight, counted s_waitcnt vmcnt(N); optional filler VALU work per chunk
//   hipcc --offload-arch=gfx950 -O3 -o ring_probe ring_probe.hip && ./ring_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));

constexpr int REC = 80, CHUNK = 64, PIECES = CHUNK * REC / 16, PPL = PIECES / 64;  // 320 pieces, 5 per lane

__device__ __forceinline__ uint32_t lds_addr(const void *p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}
template <bool NT>
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst) {
  unsigned keep;
  if (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__global__ __launch_bounds__(256) void k_copy(const v4u *__restrict__ src, v4u *__restrict__ dst, size_t n16) {
  size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t < n16) __builtin_nontemporal_store(__builtin_nontemporal_load(src + t), dst + t);
}

// filler: F dependent VALU operations per lane on the record's words
template <int F>
__device__ __forceinline__ uint32_t chew(const v4u (&r)[PPL]) {
  uint32_t a = r[0].x ^ r[1].y ^ r[2].z ^ r[3].w ^ r[4].x;
  uint32_t b = r[0].y + r[1].z + r[2].w + r[3].x + r[4].y;
#pragma unroll
  for (int i = 0; i < F / 2; ++i) {
    a = a * 0x9e3779b1u + b;
    b = (b >> 3) ^ a;
  }
  return a + b;
}

// MODE bits: 1 = result stores, 2 = flag byte stores, 4 = flag stores merged (one lane stores 8 flag bytes of 8 voxels:
// a stand-in for "whole lines"), 8 = plain instead of non-temporal result stores, 16 = no LDS pass
// CPW chunks per wave; PERSIST: the grid is what is resident, a wave strides over its groups of CPW chunks
template <int F, int MODE, int CPW, bool PERSIST>
__global__ __launch_bounds__(256) void k_regs(const unsigned char *__restrict__ rec, unsigned long long *__restrict__ res,
                                              uint8_t *__restrict__ flag, uint32_t n_chunks) {
  __shared__ v4u stage[4][PIECES];
  const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t n_groups = n_chunks / CPW;
  for (uint32_t g = blockIdx.x * 4 + wave; g < n_groups; g += gridDim.x * 4) {
    const uint32_t c0 = g * CPW;
    auto fetch = [&](uint32_t c, v4u (&buf)[PPL]) {
      const v4u *src = reinterpret_cast<const v4u *>(rec + (size_t)c * CHUNK * REC);
#pragma unroll
      for (int j = 0; j < PPL; ++j) buf[j] = __builtin_nontemporal_load(src + j * 64 + lane);
    };
    v4u b0[PPL], b1[PPL];
    fetch(c0, b0);
#pragma unroll 1
    for (uint32_t k = 0; k < CPW; ++k) {
      if (k + 1 < CPW) fetch(c0 + k + 1, b1);
      v4u r[PPL];
      if (MODE & 16) {
#pragma unroll
        for (int j = 0; j < PPL; ++j) r[j] = b0[j];
      } else {
#pragma unroll
        for (int j = 0; j < PPL; ++j) stage[wave][j * 64 + lane] = b0[j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const v4u *mine = reinterpret_cast<const v4u *>(reinterpret_cast<const unsigned char *>(stage[wave]) + lane * REC);
#pragma unroll
        for (int j = 0; j < PPL; ++j) r[j] = mine[j];
      }
      const uint32_t h = chew<F>(r);
      const size_t lv = (size_t)(c0 + k) * CHUNK + lane;
      if (MODE & 32) {  // results of two chunks leave together: 16 bytes per lane, 1 KB contiguous per instruction
        if (k & 1) __builtin_nontemporal_store(v4u{h, lane, h + 1, lane}, reinterpret_cast<v4u *>(res + (size_t)(c0 + k - 1) * CHUNK) + lane);
      } else if (MODE & 64) {  // ... of four chunks: two such stores back to back
        if ((k & 3) == 3) {
          __builtin_nontemporal_store(v4u{h, lane, h + 1, lane}, reinterpret_cast<v4u *>(res + (size_t)(c0 + k - 3) * CHUNK) + lane);
          __builtin_nontemporal_store(v4u{h, lane, h + 2, lane}, reinterpret_cast<v4u *>(res + (size_t)(c0 + k - 1) * CHUNK) + lane);
        }
      } else if (MODE & 1) {
        if (MODE & 8) res[lv] = ((unsigned long long)h << 32) | lane;
        else __builtin_nontemporal_store(((unsigned long long)h << 32) | lane, res + lv);
      }
      if (MODE & 2) {
        if (MODE & 4) {
          if (lane < 8) reinterpret_cast<unsigned long long *>(flag + (size_t)(c0 + k) * CHUNK)[lane] = h * 0x0101010101010101ull;
        } else {
          flag[lv] = (uint8_t)h;
        }
      }
      if (!(MODE & 3) && h == 0x12345u) res[0] = h;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < PPL; ++j) b0[j] = b1[j];
    }
    if (!PERSIST) break;
  }
}

// persistent ring.  WAVES waves per workgroup, R slots per wave; a wave takes chunks w, w + NW, w + 2 NW, ...
// (INTERLEAVE) or a contiguous range.
template <int R, int WAVES, int F, bool NT, bool INTERLEAVE, bool STORES>
__global__ __launch_bounds__(64 * WAVES) void k_ring(const unsigned char *__restrict__ rec, unsigned long long *__restrict__ res,
                                                     uint8_t *__restrict__ flag, uint32_t n_chunks) {
  __shared__ v4u ring[WAVES][R][PIECES];
  const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t nw = gridDim.x * WAVES, w = blockIdx.x * WAVES + wave;
  uint32_t first, stride, count;
  if (INTERLEAVE) {
    first = w;
    stride = nw;
    count = w < n_chunks ? (n_chunks - w + nw - 1) / nw : 0;
  } else {
    const uint32_t per = (n_chunks + nw - 1) / nw;
    first = w * per;
    stride = 1;
    count = first < n_chunks ? (n_chunks - first < per ? n_chunks - first : per) : 0;
  }
  const uint32_t ring_base = __builtin_amdgcn_readfirstlane(lds_addr(&ring[wave][0][0]));
  auto issue = [&](uint32_t i) {  // chunk number i of this wave into slot i % R
    const uint32_t c = first + i * stride;
    const unsigned char *src = rec + (size_t)c * CHUNK * REC + lane * 16;
    const uint32_t dst = ring_base + (i % R) * (PIECES * 16);
#pragma unroll
    for (int j = 0; j < PPL; ++j) glds16<NT>(src + j * 1024, dst + j * 1024);
  };
  constexpr int D = R - 1;  // chunks in flight
#pragma unroll
  for (int i = 0; i < D; ++i)
    if ((uint32_t)i < count) issue(i);
#pragma unroll 1
  for (uint32_t i = 0; i < count; ++i) {
    // ops issued after chunk i's loads: D - 1 younger chunks (PPL each) and, with STORES, 2 per step since
    const bool more = i + D < count;
    if (more) issue(i + D);  // into the slot read in the step before this one
    if (more) {
      wait_vm<D * (PPL + (STORES ? 2 : 0))>();
    } else {
      wait_vm<0>();
    }
    v4u r[PPL];
    const v4u *mine = reinterpret_cast<const v4u *>(reinterpret_cast<const unsigned char *>(ring[wave][i % R]) + lane * REC);
#pragma unroll
    for (int j = 0; j < PPL; ++j) r[j] = mine[j];
    const uint32_t h = chew<F>(r);
    if (STORES) {
      const size_t lv = (size_t)(first + i * stride) * CHUNK + lane;
      __builtin_nontemporal_store(((unsigned long long)h << 32) | lane, res + lv);
      flag[lv] = (uint8_t)h;
    } else if (h == 0x12345u) {
      res[0] = h;
    }
  }
}

static hipEvent_t e0, e1;
template <typename L>
static double timeit(L launch, int reps = 20) {
  for (int i = 0; i < 3; ++i) launch();
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / reps;
}

static unsigned char *rec;
static unsigned long long *res;
static uint8_t *flag;
static const uint32_t V = 1u << 24, NC = V / CHUNK;
static const double BYTES = (double)V * 89.0;  // 80 read + 8 + 1 written

template <int R, int WAVES, int F, bool NT, bool IL, bool ST>
static void run_ring(int wg_per_cu) {
  const unsigned grid = 256 * wg_per_cu;
  const double us = timeit([&] { hipLaunchKernelGGL((k_ring<R, WAVES, F, NT, IL, ST>), dim3(grid), dim3(64 * WAVES), 0, 0, rec, res, flag, NC); });
  const double b = ST ? BYTES : (double)V * 80.0;
  printf("ring R=%d waves/wg=%d wg/cu=%d F=%3d nt=%d interleave=%d stores=%d : %7.2f us  %5.2f TB/s  (lds/wg %d KB)\n", R, WAVES, wg_per_cu, F,
         (int)NT, (int)IL, (int)ST, us, b / us / 1e6, R * WAVES * PIECES * 16 / 1024);
  fflush(stdout);
}

int main() {
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const size_t rec_bytes = (size_t)V * REC;
  if (hipMalloc(&rec, rec_bytes) != hipSuccess || hipMalloc(&res, (size_t)V * 8) != hipSuccess || hipMalloc(&flag, V) != hipSuccess) return 1;
  hipMemset(rec, 1, rec_bytes);
  {
    unsigned char *dst;
    const size_t half = (size_t)V * 89 / 2 / 4096 * 4096;
    if (hipMalloc(&dst, half) != hipSuccess) return 1;
    const size_t n16 = half / 16;
    const double us = timeit([&] { hipLaunchKernelGGL(k_copy, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, (const v4u *)rec, (v4u *)dst, n16); });
    printf("copy %.1f MB read + as much written: %7.2f us  %5.2f TB/s\n", half / 1e6, us, 2.0 * half / us / 1e6);
    hipFree(dst);
  }
#define RUN_REGS(F, MODE, CPW, PERSIST, GRID)                                                                                  \
  {                                                                                                                             \
    const double us = timeit([&] { hipLaunchKernelGGL((k_regs<F, MODE, CPW, PERSIST>), dim3(GRID), dim3(256), 0, 0, rec, res, flag, NC); }); \
    const double by = (double)V * (80.0 + ((MODE) & 1 ? 8 : 0) + ((MODE) & 2 ? 1 : 0));                                          \
    printf("regs F=%3d mode %2d cpw %2d persist %d grid %5u : %7.2f us  %5.2f TB/s\n", F, MODE, CPW, (int)PERSIST, (unsigned)(GRID), us, by / us / 1e6); \
    fflush(stdout);                                                                                                             \
  }
  RUN_REGS(16, 3, 8, false, NC / 32)
  RUN_REGS(16, 0, 8, false, NC / 32)
  RUN_REGS(16, 1, 8, false, NC / 32)
  RUN_REGS(16, 2, 8, false, NC / 32)
  RUN_REGS(16, 7, 8, false, NC / 32)
  RUN_REGS(16, 11, 8, false, NC / 32)
  RUN_REGS(16, 16, 8, false, NC / 32)
  RUN_REGS(16, 19, 8, false, NC / 32)
  RUN_REGS(16, 33, 8, false, NC / 32)
  RUN_REGS(16, 35, 8, false, NC / 32)
  RUN_REGS(16, 65, 8, false, NC / 32)
  RUN_REGS(16, 67, 8, false, NC / 32)
  RUN_REGS(16, 39, 8, false, NC / 32)
  RUN_REGS(200, 33, 8, false, NC / 32)
  RUN_REGS(200, 39, 8, false, NC / 32)
  RUN_REGS(16, 3, 16, false, NC / 64)
  RUN_REGS(16, 3, 8, true, 1024)
  RUN_REGS(16, 3, 8, true, 1280)
  RUN_REGS(16, 3, 8, true, 2048)
  RUN_REGS(16, 3, 32, true, 1024)
  RUN_REGS(200, 3, 8, false, NC / 32)
  RUN_REGS(200, 3, 8, true, 1024)
  RUN_REGS(200, 3, 8, true, 1280)
  run_ring<3, 2, 16, true, true, true>(4);
  run_ring<2, 2, 16, true, true, true>(8);
  return 0;
}
