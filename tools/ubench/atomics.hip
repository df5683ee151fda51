// Microbenchmark (dev tool): fp32 atomic-add throughput on gfx950 for the hash-grid backward.
//   mode 0  random address in the whole 52 MB gradient table, agent scope (what k_hashgrid_bwd does)
//   mode 1  each XCD (s_getreg XCC_ID) adds only into its own 4 MB slice, agent scope
//   mode 2  same slices, workgroup scope (atomic executes in the XCD's own L2)
//   mode 3  lane PAIRS add to adjacent words of a random 8-byte slot (the two features of a table entry in ONE instruction)
//   mode 4  lane QUADS add to the four words of a random 16-byte slot (x-neighbour entries of an even cell)
//   mode 5  packed fp16 pairs (global_atomic_pk_add_f16), random 4-byte slot of a 26 MB fp16 table
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* tab, size_t n_all, size_t n_slice, int n_iter) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned s = gid * 2654435761u + 12345u;
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7;  // HW_REG_XCC_ID[3:0]
  for (int it = 0; it < n_iter; it++) {
    s = s * 1664525u + 1013904223u;
    const size_t r = s >> 4;
    if (MODE == 0) unsafeAtomicAdd(tab + r % n_all, 1.0f);
    else if (MODE == 1) unsafeAtomicAdd(tab + xcc * n_slice + r % n_slice, 1.0f);
    else if (MODE == 2) __hip_atomic_fetch_add(tab + xcc * n_slice + r % n_slice, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (MODE == 3) { const size_t rr = __shfl((unsigned)(r % (n_all / 2)), (threadIdx.x & 63) & ~1, 64); unsafeAtomicAdd(tab + rr * 2 + (threadIdx.x & 1), 1.0f); }
    else if (MODE == 4) { const size_t rr = __shfl((unsigned)(r % (n_all / 4)), (threadIdx.x & 63) & ~3, 64); unsafeAtomicAdd(tab + rr * 4 + (threadIdx.x & 3), 1.0f); }
    else {
      typedef _Float16 h2 __attribute__((ext_vector_type(2)));
      h2 v = {(_Float16)1.0f, (_Float16)0.5f};
      __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) h2 *)tab + r % (n_all / 2), v);
    }
  }
}
int main() {
  const size_t n_all = 13u << 20, n_slice = 1u << 20;  // 52 MB table; 4 MB slices (one hashed level in fp32)
  float* tab; (void)hipMalloc(&tab, n_all * 4); (void)hipMemset(tab, 0, n_all * 4);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int blocks = 4096, n_iter = 256;
#define RUN(MODE, name) { k<MODE><<<blocks, 256>>>(tab, n_all, n_slice, 8); (void)hipDeviceSynchronize(); (void)hipEventRecord(a); \
    k<MODE><<<blocks, 256>>>(tab, n_all, n_slice, n_iter); (void)hipEventRecord(b); (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); \
    printf("%-52s %7.3f ms  %7.1f G atomics/s\n", name, ms, (double)blocks * 256 * n_iter / ms * 1e-6); }
  for (int rep = 0; rep < 2; rep++) {
    RUN(0, "whole table, agent scope");
    RUN(1, "own 4 MB slice per XCD, agent scope");
    RUN(2, "own 4 MB slice per XCD, workgroup scope (L2-local)");
    RUN(3, "lane pairs -> adjacent words of a random 8 B slot");
    RUN(4, "lane quads -> the 4 words of a random 16 B slot");
    RUN(5, "packed fp16 pair per lane, random 4 B slot (26 MB)");
  }
  // sanity: total of mode-2 adds must be exact (sum over table == number of adds of all runs)
  return 0;
}
