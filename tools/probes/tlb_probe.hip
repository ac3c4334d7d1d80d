// Measurement probe (not part of the library): cost of scattered 64-byte reads as a function of the footprint they
// are spread over, and of how many distinct arrays one logical access touches.
//   hipcc --offload-arch=gfx950 -O3 -o tlb_probe tlb_probe.hip && ./tlb_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__global__ void k_gather(const uint32_t *__restrict__ idx, int n, const uint4 *const *arrays, int n_arrays, size_t stride16,
                         uint32_t *out) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  size_t i = (size_t)idx[t] * stride16;
  uint32_t acc = 0;
  for (int a = 0; a < n_arrays; ++a) {
    uint4 v = arrays[a][i];
    acc += v.x + v.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const size_t max_bytes = (size_t)4 << 30;
  const int n_arrays_max = 8;
  std::vector<uint4 *> bufs(n_arrays_max);
  for (auto &b : bufs) {
    if (hipMalloc(&b, max_bytes / 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(b, 1, max_bytes / 4);
  }
  uint4 **d_arrays;
  hipMalloc(&d_arrays, sizeof(uint4 *) * n_arrays_max);
  hipMemcpy(d_arrays, bufs.data(), sizeof(uint4 *) * n_arrays_max, hipMemcpyHostToDevice);
  uint32_t *d_out;
  hipMalloc(&d_out, 4);
  const int n = 1 << 18;  // 262144 scattered accesses
  std::vector<uint32_t> h(n);
  uint32_t *d_idx;
  hipMalloc(&d_idx, n * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("%10s %8s %10s\n", "footprint", "arrays", "us");
  for (size_t foot = (size_t)16 << 20; foot <= max_bytes / 4; foot <<= 2) {
    uint64_t s = 88172645463325252ull;
    const size_t slots = foot / 64;  // 64-byte granules
    for (int i = 0; i < n; ++i) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      h[i] = (uint32_t)(s % slots);
    }
    hipMemcpy(d_idx, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int na : {1, 2, 4, 8}) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        k_gather<<<(n + 255) / 256, 256>>>(d_idx, n, d_arrays, na, 4, d_out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("%8zu MB %8d %10.1f\n", foot >> 20, na, best * 1e3f);
    }
  }
  return 0;
}
