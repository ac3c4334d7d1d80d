// Measurement probe (not part of the library): where phase 1 of the occupancy sweep spends its time.  Variants add the
// pieces of the real kernel one at a time over the same 16.7 M-voxel stamp + flag arrays.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o sweep_probe sweep_probe.hip && ./sweep_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));

struct Args {
  const uint16_t *vts;
  uint8_t *vflag;
  const uint32_t *sx, *sy, *sz;
  uint64_t *res;
  uint32_t *out;
  uint32_t v_count, x_n, y_n, z_n;
};

// LEVEL 0: loads only; 1: + ring stamps; 2: + per-voxel decisions (stores compiled in, never taken with this data);
// 3: + LDS list, two barriers (the real phase 1)
template <int LEVEL, int VPT>
__global__ __launch_bounds__(256) void k_phase1(Args a) {
  __shared__ uint16_t live_list[256 * VPT];
  __shared__ uint32_t n_live;
  if (LEVEL >= 3) {
    if (threadIdx.x == 0) n_live = 0;
    __syncthreads();
  }
  const uint32_t lv0 = (blockIdx.x * 256 + threadIdx.x) * VPT;
  uint32_t acc = 0;
  if (lv0 < a.v_count) {
    uint16_t t0v[VPT];
    uint8_t flag[VPT];
    if constexpr (VPT == 8) {
      v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(a.vts + lv0));
      v2u f = __builtin_nontemporal_load(reinterpret_cast<const v2u *>(a.vflag + lv0));
      __builtin_memcpy(t0v, &t, 16);
      __builtin_memcpy(flag, &f, 8);
    } else {
      v4u t[2];
      t[0] = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(a.vts + lv0));
      t[1] = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(a.vts + lv0) + 1);
      v4u f = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(a.vflag + lv0));
      __builtin_memcpy(t0v, t, 32);
      __builtin_memcpy(flag, &f, 16);
    }
    uint32_t smax[VPT];
    if (LEVEL >= 1) {
      const uint32_t rx = lv0 & ((1u << a.x_n) - 1), ry = (lv0 >> a.x_n) & ((1u << a.y_n) - 1), rz = lv0 >> (a.x_n + a.y_n);
      const uint32_t b = a.sy[ry], c = a.sz[rz];
      const uint32_t yz = b > c ? b : c;
#pragma unroll
      for (int u = 0; u < VPT; u += 4) {
        v4u s = *reinterpret_cast<const v4u *>(a.sx + rx + u);
        smax[u] = s.x > yz ? s.x : yz;
        smax[u + 1] = s.y > yz ? s.y : yz;
        smax[u + 2] = s.z > yz ? s.z : yz;
        smax[u + 3] = s.w > yz ? s.w : yz;
      }
    } else {
#pragma unroll
      for (int u = 0; u < VPT; ++u) smax[u] = 3;
    }
    if (LEVEL < 2) {
#pragma unroll
      for (int u = 0; u < VPT; ++u) acc += t0v[u] + flag[u] + smax[u];
    } else {
      uint8_t nflag[VPT];
      bool changed = false;
#pragma unroll
      for (int u = 0; u < VPT; ++u) {
        nflag[u] = flag[u];
        const uint32_t state = flag[u] & 3, held = flag[u] & 12;
        if (t0v[u] == 0 || t0v[u] < smax[u]) {
          if (held == 4) continue;
          a.res[lv0 + u] = 0xbf800000ffull;
          nflag[u] = (uint8_t)((state == 1 ? 2 : state) | 4);
          changed = true;
          continue;
        }
        if (state == 0) {
          if (held == 8) continue;
          a.res[lv0 + u] = 0;
          nflag[u] = 8;
          changed = true;
          continue;
        }
        if (state == 1) continue;
        if (LEVEL >= 3) live_list[atomicAdd(&n_live, 1u)] = (uint16_t)(threadIdx.x * VPT + u);
        else acc += u;
      }
      if (changed) __builtin_memcpy(a.vflag + lv0, nflag, VPT);
    }
  }
  if (LEVEL >= 3) {
    __syncthreads();
    const uint32_t nl = n_live;
    for (uint32_t k = threadIdx.x; k < nl; k += 256) acc += live_list[k];
  }
  if (acc == 0x12345678u) a.out[0] = acc;
}

template <int LEVEL, int VPT>
static void run(const Args &a) {
  const unsigned grid = (a.v_count + 256 * VPT - 1) / (256 * VPT);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_phase1<LEVEL, VPT>), dim3(grid), dim3(256), 0, 0, a);
  const int reps = 50;
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_phase1<LEVEL, VPT>), dim3(grid), dim3(256), 0, 0, a);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("level %d  vpt %2d  grid %6u : %7.2f us\n", LEVEL, VPT, grid, ms * 1e3 / reps);
}

int main() {
  Args a{};
  a.x_n = a.y_n = a.z_n = 8;
  a.v_count = 1u << 24;
  std::vector<uint16_t> vts(a.v_count);
  std::vector<uint8_t> vflag(a.v_count);
  uint32_t r = 12345;
  size_t n_dirty = 0;
  for (uint32_t v = 0; v < a.v_count; ++v) {
    r = r * 1664525u + 1013904223u;
    const uint32_t p = (r >> 8) % 10000;
    const bool in_view = ((v >> 8) & 255) > 60 && ((v >> 8) & 255) < 200 && (v >> 16) > 80 && (v >> 16) < 180;
    if (!in_view) { vts[v] = 0; vflag[v] = 4; }              // never observed, result says so
    else if (p < 9700) { vts[v] = 7; vflag[v] = 0 | 8; }      // observed, empty, result says so
    else if (p < 9990) { vts[v] = 7; vflag[v] = 1; }          // holds particles, unchanged
    else { vts[v] = 7; vflag[v] = 2; ++n_dirty; }             // holds particles, changed
  }
  std::vector<uint32_t> stamps(512, 3);
  uint16_t *d_vts;
  hipMalloc(&d_vts, a.v_count * 2);
  hipMalloc(&a.vflag, a.v_count);
  hipMalloc(&a.res, (size_t)a.v_count * 8);
  hipMalloc(&a.out, 4);
  uint32_t *d_s;
  hipMalloc(&d_s, 3 * 512 * 4);
  hipMemcpy(d_vts, vts.data(), a.v_count * 2, hipMemcpyHostToDevice);
  hipMemcpy(a.vflag, vflag.data(), a.v_count, hipMemcpyHostToDevice);
  for (int i = 0; i < 3; ++i) hipMemcpy(d_s + 512 * i, stamps.data(), 512 * 4, hipMemcpyHostToDevice);
  a.vts = d_vts;
  a.sx = d_s;
  a.sy = d_s + 512;
  a.sz = d_s + 1024;
  printf("voxels %u, dirty %zu\n", a.v_count, n_dirty);
  run<0, 8>(a);
  run<1, 8>(a);
  run<2, 8>(a);
  run<3, 8>(a);
  run<0, 16>(a);
  run<1, 16>(a);
  run<2, 16>(a);
  run<3, 16>(a);
  return 0;
}
