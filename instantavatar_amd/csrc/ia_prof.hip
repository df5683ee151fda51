// ia_prof.hip -- optional in-library profiling used by bench.py: HIP events
// around every launch of the two dominant kernels (on the caller's stream) and
// device-side counters of the units they processed.  Disabled by default; when
// disabled no event is recorded and the kernels receive a null counter pointer.
// (Profiling mode is the only place where the library allocates device memory:
// a few KB of sharded counters.)
#include <vector>

#include "ia_common.h"

struct ProfKernel {
  std::vector<hipEvent_t> start, stop;
  size_t used = 0;
};
static bool g_prof_on = false;
static ProfKernel g_prof[IA_PROF_N];
// device [IA_PROF_N][IA_PROF_SHARDS][8]: counters 0/1 of a shard share one 64-byte line; kernels add
// to shard (blockIdx & (IA_PROF_SHARDS-1)) once per workgroup so that the accounting never
// serialises on a single address (one word saturates at ~88 atomics/us)
static unsigned long long *g_prof_units = nullptr;
#define IA_PROF_WORDS (IA_PROF_N * IA_PROF_SHARDS * 8)

unsigned long long *ia_prof_units(int id) { return (g_prof_on && g_prof_units) ? g_prof_units + (size_t)id * IA_PROF_SHARDS * 8 : nullptr; }

void ia_prof_begin(int id, hipStream_t s) {
  if (!g_prof_on) return;
  ProfKernel &k = g_prof[id];
  if (k.used == k.start.size()) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k.start.push_back(a); k.stop.push_back(b);
  }
  (void)hipEventRecord(k.start[k.used], s);
}
void ia_prof_end(int id, hipStream_t s) {
  if (!g_prof_on) return;
  ProfKernel &k = g_prof[id];
  (void)hipEventRecord(k.stop[k.used], s);
  k.used++;
}

extern "C" int ia_profile_enable(int on) {
  if (on && !g_prof_units) {
    if (hipMalloc((void **)&g_prof_units, sizeof(unsigned long long) * IA_PROF_WORDS) != hipSuccess)
      return ia_set_error(IA_ERR_LAUNCH, "ia_profile_enable: hipMalloc failed");
    (void)hipMemset(g_prof_units, 0, sizeof(unsigned long long) * IA_PROF_WORDS);
  }
  g_prof_on = on != 0;
  return IA_OK;
}

extern "C" int ia_profile_reset(void) {
  for (int i = 0; i < IA_PROF_N; i++) g_prof[i].used = 0;
  if (g_prof_units) (void)hipMemset(g_prof_units, 0, sizeof(unsigned long long) * IA_PROF_WORDS);
  return IA_OK;
}

// Synchronises (bench only).  units[0], units[1]: kernel-specific counters.
extern "C" int ia_profile_get(int id, double *total_ms, int64_t *launches, uint64_t *units) {
  IA_CHECK_ARG(id >= 0 && id < IA_PROF_N, "ia_profile_get: bad id");
  ProfKernel &k = g_prof[id];
  double tot = 0;
  for (size_t i = 0; i < k.used; i++) {
    (void)hipEventSynchronize(k.stop[i]);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, k.start[i], k.stop[i]);
    tot += ms;
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = (int64_t)k.used;
  if (units) {
    units[0] = units[1] = 0;
    if (g_prof_units) {
      std::vector<unsigned long long> h((size_t)IA_PROF_SHARDS * 8);
      (void)hipMemcpy(h.data(), g_prof_units + (size_t)id * IA_PROF_SHARDS * 8, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      for (int sh = 0; sh < IA_PROF_SHARDS; sh++) { units[0] += h[(size_t)sh * 8]; units[1] += h[(size_t)sh * 8 + 1]; }
    }
  }
  return IA_OK;
}

// All counters of a kernel (n <= 8 words per shard, summed over the shards).  Synchronises (bench only).
// id 0: {solves, trilinear fetches of the algorithm, fetches that loaded memory}; id 1: {samples}.
extern "C" int ia_profile_get_units(int id, uint64_t *units, int n) {
  IA_CHECK_ARG(id >= 0 && id < IA_PROF_N && units && n >= 1 && n <= 8, "ia_profile_get_units: bad argument");
  for (int i = 0; i < n; i++) units[i] = 0;
  if (!g_prof_units) return IA_OK;
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h((size_t)IA_PROF_SHARDS * 8);
  (void)hipMemcpy(h.data(), g_prof_units + (size_t)id * IA_PROF_SHARDS * 8, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  for (int sh = 0; sh < IA_PROF_SHARDS; sh++)
    for (int i = 0; i < n; i++) units[i] += h[(size_t)sh * 8 + i];
  return IA_OK;
}
