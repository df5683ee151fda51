// Microbenchmark (dev tool): the access pattern of k_encode_xcd in isolation -- every workgroup gathers aligned
// 8-byte pairs at random indices of ONE 2 MB table slice, the slice chosen from blockIdx % 8 (the XCD the
// dispatcher places the workgroup on), so that each XCD's L2 serves exactly one slice.  Variants:
//   loads in flight per lane (8 / 16 / 32 / 64), 4- vs 8- vs 16-byte gathers, slice in the own L2 vs a table that
//   does not fit (26 MB, all levels from every XCD).
// Output: gathers/s and the implied L2 requests/s -> the empirical ceiling quoted in DESIGN.md for the hash-grid lookup.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/l2gather.hip -o tools/ubench/l2gather && tools/ubench/l2gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int INFLIGHT, int BYTES, bool SHARDED>
__global__ __launch_bounds__(256) void k(const uint32_t* __restrict__ tab, uint32_t slice_words, uint32_t total_words, float* out, int n_iter) {
  const uint32_t* base = SHARDED ? tab + (size_t)(blockIdx.x & 7) * slice_words : tab;
  const uint32_t mask = (SHARDED ? slice_words : total_words) - 1;
  uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  uint32_t acc = 0;
  for (int it = 0; it < n_iter; it++) {
    uint32_t v[INFLIGHT];
#pragma unroll
    for (int k2 = 0; k2 < INFLIGHT; k2++) {
      s = s * 1664525u + 1013904223u;
      const uint32_t i = (s >> 7) & mask;
      if (BYTES == 4) v[k2] = base[i];
      else if (BYTES == 8) { const uint2 p = *reinterpret_cast<const uint2*>(base + (i & ~1u)); v[k2] = p.x ^ p.y; }
      else { const uint4 p = *reinterpret_cast<const uint4*>(base + (i & ~3u)); v[k2] = p.x ^ p.y ^ p.z ^ p.w; }
    }
#pragma unroll
    for (int k2 = 0; k2 < INFLIGHT; k2++) acc ^= v[k2];
  }
  if (acc == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = 1.f;
}

int main() {
  const uint32_t slice_words = 1u << 19;           // 2 MB: one hashed level of the fp16 table
  const uint32_t total_words = 1u << 23;           // 32 MB (>= the whole 26 MB table, power of two)
  uint32_t* tab; float* out;
  hipMalloc(&tab, (size_t)total_words * 4); hipMalloc(&out, 1 << 24);
  hipMemset(tab, 1, (size_t)total_words * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int blocks = 256 * 8;
#define RUN(INF, BYTES, SH, name) { const int n_iter = 4096 / INF; \
    k<INF, BYTES, SH><<<blocks, 256>>>(tab, slice_words, total_words, out, 4); hipDeviceSynchronize(); hipEventRecord(a); \
    k<INF, BYTES, SH><<<blocks, 256>>>(tab, slice_words, total_words, out, n_iter); hipEventRecord(b); hipEventSynchronize(b); \
    float ms; hipEventElapsedTime(&ms, a, b); const double g = (double)blocks * 256 * n_iter * INF; \
    printf("%-46s %2d in flight %2d B : %7.3f ms  %7.1f G gathers/s  (%.2f gathers/clk/CU @2.4GHz)\n", name, INF, BYTES, ms, g / ms * 1e-6, g / (ms * 1e-3) / 2.4e9 / 256); }
  RUN(8, 8, true, "own-XCD 2 MB slice");
  RUN(16, 8, true, "own-XCD 2 MB slice");
  RUN(32, 8, true, "own-XCD 2 MB slice");
  RUN(64, 8, true, "own-XCD 2 MB slice");
  RUN(32, 4, true, "own-XCD 2 MB slice");
  RUN(32, 16, true, "own-XCD 2 MB slice");
  RUN(32, 8, false, "whole 32 MB table from every XCD");
  RUN(32, 4, false, "whole 32 MB table from every XCD");
  return 0;
}
