// Microbenchmark (dev tool): cost of fetching the 8 trilinear corners (48-byte records, x-neighbours
// contiguous) of random cells of the 32x128x128 transform grid, for three lane mappings:
//   A  lane = cell           : 24 x 16-byte loads per lane            (k_search today)
//   B  quad = cell           : lane 4m+k (k<3) loads piece k, 8 loads per quad
//   C  8 lanes = cell        : lane 8m+k (k<6) loads piece k of the 96-byte x-pair, 4 loads per group
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int W = 128, H = 128, D = 32;
template <int MODE>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ tab, float* out, int n_iter, int win) {
  const int lane = threadIdx.x & 63;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.f;
  unsigned s = gid * 2654435761u + 12345u;
  for (int it = 0; it < n_iter; it++) {
    s = s * 1664525u + 1013904223u;
    unsigned r = s >> 8;
    if (MODE == 1) r = __shfl((int)r, lane & ~3, 64);
    if (MODE == 2) r = __shfl((int)r, lane & ~7, 64);
    // cells inside a win^3 window (per workgroup origin): win = 128 -> whole grid
    const int wx = win > W - 1 ? W - 1 : win, wz = win > D - 1 ? D - 1 : win;
    const int ox = (blockIdx.x * 7) % (W - wx), oy = (blockIdx.x * 13) % (H - wx), oz = (blockIdx.x * 3) % (D - wz);
    const int x = ox + r % wx, y = oy + (r / 128) % wx, z = oz + (r / 16384) % wz;
    const int base = (z * H + y) * W + x;
    if (MODE == 0) {
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const float4* p = tab + (size_t)(base + (c & 1) + ((c >> 1) & 1) * W + (c >> 2) * W * H) * 3;
        const float4 a = p[0], b = p[1], d = p[2];
        acc += a.x + b.y + d.z;
      }
    } else if (MODE == 1) {
      const int k = lane & 3;
      if (k < 3) {
#pragma unroll
        for (int c = 0; c < 8; c++) {
          const float4 a = tab[(size_t)(base + (c & 1) + ((c >> 1) & 1) * W + (c >> 2) * W * H) * 3 + k];
          acc += a.x + a.w;
        }
      }
    } else {
      const int k = lane & 7;
      if (k < 6) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const float4 a = tab[(size_t)(base + (c & 1) * W + (c >> 1) * W * H) * 3 + k];
          acc += a.x + a.w;
        }
      }
    }
  }
  if (acc == 123.456f) out[gid] = acc;
}
int main() {
  float4* tab; float* out;
  const size_t n = (size_t)W * H * D * 3;
  hipMalloc(&tab, n * 16); hipMalloc(&out, 1 << 24);
  hipMemset(tab, 0, n * 16);
  const int blocks = 256 * 8;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
#define RUN(MODE, cells_per_wave_iter, name) { const int n_iter = 256 * 64 / cells_per_wave_iter / 8; k<MODE><<<blocks, 256>>>(tab, out, 4, win); hipDeviceSynchronize(); hipEventRecord(a); \
    k<MODE><<<blocks, 256>>>(tab, out, n_iter, win); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); \
    double cells = (double)blocks * 4 * n_iter * cells_per_wave_iter; \
    printf("%-34s %7.3f ms  %6.2f G cells/s  (%.0f clk/CU per 64 cells @2.1GHz)\n", name, ms, cells / ms * 1e-6, ms * 1e-3 * 2.1e9 * 256 / (cells / 64)); }
  for (int win : {128, 32, 16, 8, 4}) {
    printf("--- window %d^3 cells per workgroup\n", win);
    RUN(0, 64, "A lane=cell (24 loads/lane)");
    RUN(1, 16, "B quad=cell (8 loads/quad)");
    RUN(2, 8, "C 8 lanes=cell (4 x 96B rows)");
  }
  return 0;
}
