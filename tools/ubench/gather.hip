// Microbenchmark: cost of divergent 16-byte gathers on gfx950 as a function of the number of
// active lanes and of line sharing (dev tool; informs the k_search / k_field design).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
template <int ACTIVE, int MODE>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ tab, const int* __restrict__ idx, float* out, int n_iter, int tab_n) {
  const int lane = threadIdx.x & 63;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.f;
  unsigned s = gid * 2654435761u + 12345u;
  for (int it = 0; it < n_iter; it++) {
    s = s * 1664525u + 1013904223u;
    int a;
    if (MODE == 0) a = (s >> 8) % tab_n;                   // every lane its own random line
    else if (MODE == 1) a = ((s >> 8) % tab_n) & ~3 | (lane & 3);  // groups of 4 lanes share... still random per lane
    else a = (__shfl((int)((s >> 8) % tab_n), lane & ~3, 64) & ~3) | (lane & 3);  // 4 adjacent lanes -> 4 adjacent 16B (one 64B line)
    if (lane < ACTIVE) {
      const float4 v = tab[a];
      acc += v.x + v.y + v.z + v.w;
    }
  }
  if (acc == 123.456f) out[gid] = acc;
}
int main() {
  const int tab_n = 1 << 22;  // allocation: 64 MB
  int tab_n_g = tab_n;
  float4* tab; int* idx; float* out;
  hipMalloc(&tab, tab_n * 16); hipMalloc(&out, 1 << 24); hipMalloc(&idx, 4);
  hipMemset(tab, 0, tab_n * 16);
  const int blocks = 256 * 8, n_iter = 512;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
#define RUN(ACT, MODE, name) { k<ACT, MODE><<<blocks, 256>>>(tab, idx, out, 16, tab_n_g); hipDeviceSynchronize(); hipEventRecord(a); \
    k<ACT, MODE><<<blocks, 256>>>(tab, idx, out, n_iter, tab_n_g); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); \
    double wave_instr = (double)blocks * 4 * n_iter; double lanes = wave_instr * ACT; \
    printf("%-28s active=%2d  %7.3f ms  %6.2f G wave-instr/s  %7.1f G lane-loads/s  (%.2f clk/CU per wave-instr @2.1GHz)\n", name, ACT, ms, wave_instr / ms * 1e-6, lanes / ms * 1e-6, ms * 1e-3 * 2.1e9 * 256 / wave_instr); }
  for (int sz = 16; sz <= 26; sz += 2) {  // table of 2^sz bytes
    tab_n_g = (1 << sz) / 16;
    printf("--- table %d KB\n", (1 << sz) / 1024);
    RUN(64, 0, "random, own line per lane"); RUN(16, 0, "random, own line per lane");
    RUN(64, 2, "4 lanes share a 64B line");
  }
  return 0;
}
