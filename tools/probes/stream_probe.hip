// Measurement probe (not part of the library): time of one kernel that only streams a small buffer (the size of the
// sweep's per-voxel stamp + flag arrays) as a function of bytes per thread and workgroup count.
//   hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip && ./stream_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

template <int VEC, bool NT>
__global__ __launch_bounds__(256) void k_stream(const v4u *__restrict__ src, size_t n16, uint32_t *out) {
  size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0;
  v4u v[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    size_t k = ((size_t)i * gridDim.x * 256) + t;  // each request of a wave is contiguous
    if (k < n16) v[i] = NT ? __builtin_nontemporal_load(src + k) : src[k];
    else v[i] = v4u{0, 0, 0, 0};
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc += v[i].x ^ v[i].w;
  if (acc == 0x12345678u) out[0] = acc;
}

template <int VEC, bool NT>
static void run(const v4u *src, size_t bytes, uint32_t *out) {
  const size_t n16 = bytes / 16;
  const unsigned grid = (unsigned)((n16 + 256ull * VEC - 1) / (256ull * VEC));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_stream<VEC, NT>), dim3(grid), dim3(256), 0, 0, src, n16, out);
  const int reps = 50;
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_stream<VEC, NT>), dim3(grid), dim3(256), 0, 0, src, n16, out);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  printf("bytes %6.1f MB  vec %d x16B  nt %d  grid %6u : %7.2f us  %6.2f TB/s\n", bytes / 1e6, VEC, (int)NT, grid, us, bytes / us / 1e6);
}

int main() {
  const size_t max_bytes = (size_t)2 << 30;
  v4u *buf;
  uint32_t *out;
  if (hipMalloc(&buf, max_bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
  hipMemset(buf, 1, max_bytes);
  for (size_t mb : {16, 50, 200, 1470}) {
    const size_t bytes = mb * 1000000ull / 4096 * 4096;
    run<1, true>(buf, bytes, out);
    run<1, false>(buf, bytes, out);
    run<2, true>(buf, bytes, out);
    run<4, true>(buf, bytes, out);
    run<4, false>(buf, bytes, out);
    run<8, true>(buf, bytes, out);
  }
  return 0;
}
