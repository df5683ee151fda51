// ia_common.h -- shared helpers for the gfx950 kernels (wave64 everywhere).
#pragma once
#include <hip/hip_runtime.h>
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
  // set when all hashed levels have the same size and follow each other (tcnn default):
  // level l >= n_dense lives at table + hash_base + (l - n_dense) * hash_size
  uint32_t n_dense, hash_base, hash_size;
};
int ia_make_field_dev(const ia_field *f, FieldDev *out);

struct OccDev {
  int G;
  float mn[3];
  float mx[3];
};

// ---- optional profiling (ia_prof.hip) ---------------------------------------
#define IA_PROF_N 2
#define IA_PROF_SEARCH 0   // units[0] = (point,init) solves, units[1] = grid fetches
#define IA_PROF_FIELD 1    // units[0] = samples evaluated
unsigned long long *ia_prof_units(int id);
void ia_prof_begin(int id, hipStream_t s);
void ia_prof_end(int id, hipStream_t s);

// ---- explicit FMA convention --------------------------------------------------
// Branchy fp32 chains (Broyden root finder, trilinear fetch, marcher) use ONE fixed
// operation sequence shared with the CPU checker: contraction is switched off in
// those functions and every `sum + x*y` is an explicit fma into the running sum.
#define IA_DOT3(a0, b0, a1, b1, a2, b2) __builtin_fmaf((a2), (b2), __builtin_fmaf((a1), (b1), (a0) * (b0)))

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
