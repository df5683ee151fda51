// ia_common.h -- shared helpers for the gfx950 kernels (wave64 everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/instantavatar_hip.h"

#define IA_WAVE 64

extern thread_local char ia_err_buf[512];
int ia_set_error(int code, const char *fmt, ...);

#define IA_CHECK_ARG(cond, ...)                          \
  do {                                                   \
    if (!(cond)) return ia_set_error(IA_ERR_ARG, __VA_ARGS__); \
  } while (0)

// Launch check that never synchronises: only the launch status is read.
#define IA_LAUNCH_CHECK(name)                                               \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess)                                                  \
      return ia_set_error(IA_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
  } while (0)

static inline int ia_div_up(long a, long b) { return (int)((a + b - 1) / b); }
static inline size_t ia_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// ---- device-side descriptors (passed by value as kernel arguments) ---------
struct SnarfGridDev {
  int D, H, W;
  float off[3];
  float scl[3];
};
static inline SnarfGridDev ia_make_grid_dev(const ia_snarf_grid *g) {
  SnarfGridDev d;
  d.D = g->D; d.H = g->H; d.W = g->W;
  for (int i = 0; i < 3; i++) { d.off[i] = g->offset[i]; d.scl[i] = g->scale[i]; }
  return d;
}

struct HashLevelsDev {
  int n_levels;
  float scale[IA_MAX_LEVELS];
  uint32_t res[IA_MAX_LEVELS];
  uint32_t offset[IA_MAX_LEVELS];
  uint32_t size[IA_MAX_LEVELS];
  uint32_t hashed[IA_MAX_LEVELS];  // 1: coherent prime hash, 0: dense index
};

struct FieldDev {
  float center[3];
  float inv_unused[3];
  float scale[3];
  HashLevelsDev lv;
  const uint32_t *table;  // half2 per entry
  const uint16_t *sig_w1, *sig_w2, *col_w1, *col_w2, *col_w3;
  const uint16_t *frags;  // prebuilt MFMA A-fragment image or null
  uint32_t *enc_ws;       // level-plane scratch of the XCD-sharded encoding or null
  size_t enc_ws_samples;
  int enc_split;          // sharded encoding: tiles of every four of the second level group that XCDs 0-3 take (1..3)
  // set when all hashed levels have the same size and follow each other (tcnn default):
  // level l >= n_dense lives at table + hash_base + (l - n_dense) * hash_size
  uint32_t n_dense, hash_base, hash_size;
};
int ia_make_field_dev(const ia_field *f, FieldDev *out);
// field stage (encoding + MLPs) on V samples (n_dev: optional device-side live count); acts: training record or null
int ia_launch_field(const float *x, int V, const int32_t *n_dev, const FieldDev &F, float *rgb, float *sigma,
                    hipStream_t s, uint16_t *acts);

// carves typed, 256-byte aligned pieces out of a caller-provided workspace
struct WsCarver {
  char *base; size_t off, cap;
  WsCarver(void *p, size_t c) : base((char *)p), off(0), cap(c) {}
  template <typename T> T *take(size_t n) {
    T *r = (T *)(base + off);
    off += ia_align(n * sizeof(T));
    return r;
  }
  bool ok() const { return off <= cap; }
};


struct OccDev {
  int G;
  float mn[3];
  float mx[3];
};

// ---- optional profiling (ia_prof.hip) ---------------------------------------
#define IA_PROF_N 2
#define IA_PROF_SEARCH 0   // units[0] = (point,init) solves, units[1] = grid fetches
#define IA_PROF_FIELD 1    // units[0] = samples evaluated
#define IA_PROF_SHARDS 64
unsigned long long *ia_prof_units(int id);  // [IA_PROF_SHARDS][8] words; counters 0/1 of shard s at [s*8 + {0,1}]
void ia_prof_begin(int id, hipStream_t s);
void ia_prof_end(int id, hipStream_t s);

// ---- explicit FMA convention --------------------------------------------------
// Branchy fp32 chains (Broyden root finder, trilinear fetch, marcher) use ONE fixed
// operation sequence shared with the CPU checker: contraction is switched off in
// those functions and every `sum + x*y` is an explicit fma into the running sum.
#define IA_DOT3(a0, b0, a1, b1, a2, b2) __builtin_fmaf((a2), (b2), __builtin_fmaf((a1), (b1), (a0) * (b0)))

// ---- Fast-SNARF helpers shared by the search kernel and its callers ----------------------
struct BoneIds { int32_t id[IA_N_INIT_MAX]; };
static inline int ia_make_bones(const int32_t *bone_ids, int n_init, BoneIds *b) {
  if (!bone_ids || n_init < 1 || n_init > IA_N_INIT_MAX) return -1;
  for (int i = 0; i < IA_N_INIT_MAX; i++) b->id[i] = i < n_init ? bone_ids[i] : 0;
  for (int i = 0; i < n_init; i++) if (bone_ids[i] < 0 || bone_ids[i] >= IA_N_JOINTS) return -1;
  return 0;
}

// grid_sample un-normalisation, align_corners = true (fuse_cuda_kernel_fast.cu:62-91)
__device__ __forceinline__ float src_index(float coord, int size) {
  coord = ((coord + 1.f) / 2) * (size - 1);
  // reference (:84-91): coord > INT_MAX - 1 || coord < INT_MIN || !isfinite(coord) -> -100.  (float)(INT_MAX - 1) and
  // (float)INT_MIN are +-2^31, so the three tests are ONE: NOT (|coord| <= 2^31) -- false for NaN and inf as well
  // (one v_cmp with an abs modifier + one select instead of three compares, two ORs and a class test)
  return __builtin_fabsf(coord) <= 2147483648.0f ? coord : -100.0f;
}

// true when none of the 8 trilinear corners of the fetch at normalised (gx,gy,gz) lies inside
// the grid, i.e. the fetch returns J = 0 without touching memory
__device__ __forceinline__ bool fetch_all_oob(const SnarfGridDev &g, float gx, float gy, float gz) {
  const int x0 = (int)floorf(src_index(gx, g.W)), y0 = (int)floorf(src_index(gy, g.H)), z0 = (int)floorf(src_index(gz, g.D));
  return x0 < -1 || x0 >= g.W || y0 < -1 || y0 >= g.H || z0 < -1 || z0 >= g.D;
}

// A (point x_d, init bone with transform T[4x4]) solve is TRIVIAL when the fetch at its initial
// guess x0 = R^T (x_d - t) (fuse_cuda_kernel_fast.cu:287-293) has all 8 corners outside the
// grid: J = 0 gives J_inv0 = 0, the update is 0, x never moves, every later fetch is zero too and
// the residual stays -x_d -- the reference kernel ends in `diverged`, in `converged` with the
// bounds test failing, or in ten iterations of NaN: never a valid root.
__device__ __forceinline__ bool ia_solve_is_trivial(const SnarfGridDev &g, const float *__restrict__ T, float a0,
                                                    float a1, float a2) {
  const float ixd = a0 - T[3], iyd = a1 - T[7], izd = a2 - T[11];
  const float c0 = IA_DOT3(ixd, T[0], iyd, T[4], izd, T[8]);
  const float c1 = IA_DOT3(ixd, T[1], iyd, T[5], izd, T[9]);
  const float c2 = IA_DOT3(ixd, T[2], iyd, T[6], izd, T[10]);
  return fetch_all_oob(g, g.scl[0] * (c0 + g.off[0]), g.scl[1] * (c1 + g.off[1]), g.scl[2] * (c2 + g.off[2]));
}

// ---- wave helpers -----------------------------------------------------------
__device__ __forceinline__ int ia_lane() { return threadIdx.x & 63; }

// exclusive prefix sum over the 64 lanes of a wave (int), also returns total
__device__ __forceinline__ int ia_wave_excl_scan(int v, int &total) {
  int lane = __lane_id();
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  total = __shfl(x, 63, 64);
  return x - v;
}

__device__ __forceinline__ float ia_wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float ia_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// float atomic min/max through the ordered-int trick (buffer initialised to
// +inf / -inf).  The plain pre-read skips the atomic when it cannot tighten the
// bound (bounds only ever tighten, so a stale read is conservative).
__device__ __forceinline__ void ia_atomic_min_f(float *addr, float v) {
  float cur = __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!(v < cur)) return;
  if (v >= 0.f) atomicMin((int *)addr, __float_as_int(v));
  else atomicMax((unsigned int *)addr, __float_as_uint(v));
}
__device__ __forceinline__ void ia_atomic_max_f(float *addr, float v) {
  float cur = __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!(v > cur)) return;
  if (v >= 0.f) atomicMax((int *)addr, __float_as_int(v));
  else atomicMin((unsigned int *)addr, __float_as_uint(v));
}
