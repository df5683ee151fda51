// Microbenchmark (dev tool, round 3): the trilinear fetch of k_search with 64-BYTE-ALIGNED corner records
// (12 floats + 4 padding), so that the four lanes of a quad reading the four 16-byte pieces of ONE record touch exactly
// one 64-byte segment:
//   A   lane = cell, 48-byte records : 24 x 16-byte loads per lane                      (k_search r02)
//   A64 lane = cell, 64-byte records : 24 x 16-byte loads per lane                      (padding alone)
//   D   quad = cell, 64-byte records : lane 4m+k loads piece k of each of the 8 corners (8 loads per quad and cell)
//   D4  as D, but every lane owns a cell and the quad serves its four lanes in four rounds (cell broadcast by DPP):
//       what k_search would do
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int W = 128, H = 128, D = 32;
template <int MODE>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ tab, float* out, int n_iter, int win) {
  const int lane = threadIdx.x & 63;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.f;
  unsigned s = gid * 2654435761u + 12345u;
  const int wx = win > W - 1 ? W - 1 : win, wz = win > D - 1 ? D - 1 : win;
  const int ox = (blockIdx.x * 7) % (W - wx), oy = (blockIdx.x * 13) % (H - wx), oz = (blockIdx.x * 3) % (D - wz);
  for (int it = 0; it < n_iter; it++) {
    s = s * 1664525u + 1013904223u;
    unsigned r = s >> 8;
    if (MODE == 2) r = __shfl((int)r, lane & ~3, 64);
    const int x = ox + r % wx, y = oy + (r / 128) % wx, z = oz + (r / 16384) % wz;
    const int base = (z * H + y) * W + x;
    if (MODE == 0 || MODE == 1) {
      constexpr int S = MODE == 0 ? 3 : 4;
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const float4* p = tab + (size_t)(base + (c & 1) + ((c >> 1) & 1) * W + (c >> 2) * W * H) * S;
        const float4 a = p[0], b = p[1], d = p[2];
        acc += a.x + b.y + d.z;
      }
    } else if (MODE == 2) {
      const int k = lane & 3;
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const float4 a = tab[(size_t)(base + (c & 1) + ((c >> 1) & 1) * W + (c >> 2) * W * H) * 4 + k];
        acc += a.x + a.w;
      }
    } else if (MODE == 4) {   // B4: 48-byte records, lanes k < 3 of a quad load the three pieces, four rounds
      const int k = lane & 3;
      float4 row = make_float4(0, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int b = __shfl(base, (lane & ~3) | t, 64);
        float4 a4 = make_float4(0, 0, 0, 0);
        if (k < 3) {
#pragma unroll
          for (int c = 0; c < 8; c++) {
            const float4 a = tab[(size_t)(b + (c & 1) + ((c >> 1) & 1) * W + (c >> 2) * W * H) * 3 + k];
            a4.x += a.x; a4.y += a.y; a4.z += a.z; a4.w += a.w;
          }
        }
        const float r0 = __shfl(a4.x, (lane & ~3) | 0, 64), r1 = __shfl(a4.y, (lane & ~3) | 1, 64), r2 = __shfl(a4.z, (lane & ~3) | 2, 64);
        if (k == t) { row.x += r0; row.y += r1; row.z += r2; }
      }
      acc += row.x + row.y + row.z;
    } else {
      const int k = lane & 3;
      float4 row = make_float4(0, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int b = __shfl(base, (lane & ~3) | t, 64);   // (DPP quad_perm broadcast in the real kernel)
        float4 a4 = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 8; c++) {
          const float4 a = tab[(size_t)(b + (c & 1) + ((c >> 1) & 1) * W + (c >> 2) * W * H) * 4 + k];
          a4.x += a.x; a4.y += a.y; a4.z += a.z; a4.w += a.w;
        }
        // rows 0..2 back to the target lane
        const float r0 = __shfl(a4.x, (lane & ~3) | 0, 64), r1 = __shfl(a4.y, (lane & ~3) | 1, 64), r2 = __shfl(a4.z, (lane & ~3) | 2, 64);
        if (k == t) { row.x += r0; row.y += r1; row.z += r2; }
      }
      acc += row.x + row.y + row.z;
    }
  }
  if (acc == 123.456f) out[gid] = acc;
}
int main() {
  float4* tab; float* out;
  const size_t n = (size_t)W * H * D * 4;
  hipMalloc(&tab, n * 16); hipMalloc(&out, 1 << 24);
  hipMemset(tab, 0, n * 16);
  const int blocks = 256 * 8;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
#define RUN(MODE, cells_per_wave_iter, name) { const int n_iter = 256 * 64 / cells_per_wave_iter / 8; k<MODE><<<blocks, 256>>>(tab, out, 4, win); hipDeviceSynchronize(); hipEventRecord(a); \
    k<MODE><<<blocks, 256>>>(tab, out, n_iter, win); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); \
    double cells = (double)blocks * 4 * n_iter * cells_per_wave_iter; \
    printf("%-44s %7.3f ms  %6.2f G cells/s  (%.0f clk/CU per 64 cells @2.1GHz)\n", name, ms, cells / ms * 1e-6, ms * 1e-3 * 2.1e9 * 256 / (cells / 64)); }
  for (int win : {128, 32, 16, 8, 4}) {
    printf("--- window %d^3 cells per workgroup\n", win);
    RUN(0, 64, "A   lane=cell 48 B records (24 loads/lane)");
    RUN(1, 64, "A64 lane=cell 64 B records (24 loads/lane)");
    RUN(2, 16, "D   quad=cell 64 B records (8 loads/quad)");
    RUN(3, 64, "D4  quad serves its 4 lanes in 4 rounds");
    RUN(4, 64, "B4  as D4 on 48 B records (3 lanes load)");
  }
  return 0;
}
