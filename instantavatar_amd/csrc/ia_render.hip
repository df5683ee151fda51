// ia_render.hip -- occupancy-grid ray marcher, compositor, occupancy build and
// the fused per-frame pipelines (no host synchronisation anywhere).
//
// Reference semantics: renderers/cuda/raymarcher.cu:13-73 (march test),
// :116-161 (march train), :200-235 (composite test);
// renderers/raymarcher_acc.py:83-138 (render_test loop);
// models/structures/density_grid.py:95-125 (occupancy build);
// deformers/snarf_deformer.py:95-103,127-141.
#include "ia_common.h"

int ia_launch_field(const float *x, int V, const int32_t *n_dev, const FieldDev &F, float *rgb,
                    float *sigma, hipStream_t s, uint16_t *acts);

__device__ __forceinline__ float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }

__device__ __forceinline__ bool occ_test(const uint32_t *__restrict__ bits, int G, int nx, int ny, int nz) {
  const uint32_t idx = (uint32_t)((nx * G + ny) * G + nz);
  return (bits[idx >> 5] >> (idx & 31)) & 1u;
}

struct MarchRay {
  float ox, oy, oz, dx, dy, dz, cx, cy, cz, sx, sy, sz, far, dt;
};

__device__ __forceinline__ bool march_occupied(const MarchRay &r, const uint32_t *__restrict__ bits, int G,
                                               float t, float &x, float &y, float &z) {
  x = __builtin_fmaf(t, r.dx, r.ox); y = __builtin_fmaf(t, r.dy, r.oy); z = __builtin_fmaf(t, r.dz, r.oz);
  const float fx = (x - r.cx) * r.sx, fy = (y - r.cy) * r.sy, fz = (z - r.cz) * r.sz;
  const int nx = (int)clampf(fx, 0.0f, G - 1.0f);  // raymarcher.cu:49-51
  const int ny = (int)clampf(fy, 0.0f, G - 1.0f);
  const int nz = (int)clampf(fz, 0.0f, G - 1.0f);
  return occ_test(bits, G, nx, ny, nz);
}

// `while (t < lim) t += dt;` -- the reference loop's float adds without its occupancy test -- eight steps per trip: t grows
// monotonically (dt > 0), so "the eighth value is below the limit" says the same of the seven before it.
__device__ __forceinline__ void march_skip(float &t, float dt, float lim) {
  if (dt > 0.f) {
    for (;;) {
      const float a = t + dt, b = a + dt, c = b + dt, d = c + dt, e = d + dt, f = e + dt, g = f + dt;
      if (!(g < lim)) break;
      t = g + dt;
    }
  }
  while (t < lim) t += dt;
}

__device__ __forceinline__ MarchRay load_ray(const float *__restrict__ o, const float *__restrict__ d,
                                             const float *__restrict__ fars, const float *__restrict__ step,
                                             size_t n, const float *aabb_mn, const float *aabb_mx, int G) {
  MarchRay r;
  r.ox = o[n * 3]; r.oy = o[n * 3 + 1]; r.oz = o[n * 3 + 2];
  r.dx = d[n * 3]; r.dy = d[n * 3 + 1]; r.dz = d[n * 3 + 2];
  r.cx = aabb_mn[0]; r.cy = aabb_mn[1]; r.cz = aabb_mn[2];
  // scale = max_corner - min_corner (raymarcher_acc.py:105); sx = grid_size / scale (:41)
  r.sx = G / (aabb_mx[0] - aabb_mn[0]); r.sy = G / (aabb_mx[1] - aabb_mn[1]); r.sz = G / (aabb_mx[2] - aabb_mn[2]);
  r.far = fars[n]; r.dt = step[n];
  return r;
}

// ---------------------------------------------------------------------------
// a13 raymarch_test (dense reference layout)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_raymarch_test(const float *__restrict__ rays_o,
                                                       const float *__restrict__ rays_d, float *nears,
                                                       const float *__restrict__ fars,
                                                       const int64_t *__restrict__ alive, int n_alive,
                                                       const uint32_t *__restrict__ bits, OccDev occ,
                                                       const float *__restrict__ step, int N_steps,
                                                       float *__restrict__ pts, float *__restrict__ deltas,
                                                       float *__restrict__ depths) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_alive) return;
  const size_t n = (size_t)alive[i];
  const MarchRay r = load_ray(rays_o, rays_d, fars, step, n, occ.mn, occ.mx, occ.G);
  int s = 0;
  float t = nears[n];
  while (t < r.far && s < N_steps) {
    float x, y, z;
    if (march_occupied(r, bits, occ.G, t, x, y, z)) {
      const size_t o = (size_t)i * N_steps + s;
      pts[o * 3] = x; pts[o * 3 + 1] = y; pts[o * 3 + 2] = z;
      deltas[o] = r.dt; depths[o] = t;
      s++;
    }
    t += r.dt;
  }
  for (; s < N_steps; s++) {  // the reference zero-fills (raymarcher.cu:89-91)
    const size_t o = (size_t)i * N_steps + s;
    pts[o * 3] = 0.f; pts[o * 3 + 1] = 0.f; pts[o * 3 + 2] = 0.f;
    deltas[o] = 0.f; depths[o] = 0.f;
  }
  nears[n] = t;
}

// a15 raymarch_train
__global__ __launch_bounds__(256) void k_raymarch_train(const float *__restrict__ rays_o,
                                                        const float *__restrict__ rays_d,
                                                        const float *__restrict__ nears,
                                                        const float *__restrict__ fars, int n_rays,
                                                        const uint32_t *__restrict__ bits, OccDev occ,
                                                        const float *__restrict__ step, int N_steps,
                                                        float *__restrict__ depths) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_rays) return;
  const MarchRay r = load_ray(rays_o, rays_d, fars, step, (size_t)n, occ.mn, occ.mx, occ.G);
  int s = 0;
  float t = nears[n];
  while (t < r.far && s < N_steps) {
    float x, y, z;
    if (march_occupied(r, bits, occ.G, t, x, y, z)) { depths[(size_t)n * N_steps + s] = t; s++; }
    t += r.dt;
  }
  for (; s < N_steps; s++) depths[(size_t)n * N_steps + s] = 0.f;
}

// ---------------------------------------------------------------------------
// a14 composite_test (dense reference layout)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void composite_step(float sg, float dl, float tdepth, const float *rgb3,
                                               float thresh, float &T, float &c0, float &c1, float &c2,
                                               float &dep) {
  const float tau = __expf(-sg * dl);  // raymarcher.cu:219
  const float alpha = 1.0f - tau;
  if (alpha < thresh) return;          // :221-224 (T unchanged)
  const float w = alpha * T;
  c0 += w * rgb3[0]; c1 += w * rgb3[1]; c2 += w * rgb3[2];
  dep += w * tdepth;
  T *= tau;
}

__global__ __launch_bounds__(256) void k_composite_test(const float *__restrict__ rgb,
                                                        const float *__restrict__ sigma,
                                                        const float *__restrict__ delta,
                                                        const float *__restrict__ depth,
                                                        const int64_t *__restrict__ alive, int n_alive,
                                                        int N_steps, float *color, float *depth_out,
                                                        float *nohit, float thresh) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_alive) return;
  const size_t n = (size_t)alive[i];
  float T = nohit[n];
  float c0 = color[n * 3], c1 = color[n * 3 + 1], c2 = color[n * 3 + 2], dep = depth_out[n];
  int s = 0;
  while (s < N_steps && (double)T > 1e-4 && delta[(size_t)i * N_steps + s] > 0) {
    const size_t o = (size_t)i * N_steps + s;
    composite_step(sigma[o], delta[o], depth[o], rgb + o * 3, thresh, T, c0, c1, c2, dep);
    s++;
  }
  color[n * 3] = c0; color[n * 3 + 1] = c1; color[n * 3 + 2] = c2;
  depth_out[n] = dep;
  nohit[n] = T;
}

// ---------------------------------------------------------------------------
// a6 candidate reduction (snarf_deformer.py:130-141 / 147-158)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void cand_max(const float *__restrict__ cand_rgb,
                                         const float *__restrict__ cand_sigma, int off, int cnt, int n_init,
                                         float fill, bool nan_to_num, float &sg, float *rgb3) {
  // slots are ordered by init index; invalid slots carry `fill` and rgb 0.  The
  // first maximum wins (torch.max), so a valid candidate must EXCEED every
  // earlier slot.  With cnt < n_init at least one invalid slot exists; whether
  // it precedes the valid ones only matters for ties with `fill`, where rgb of
  // a tied valid candidate would be taken only if it came first -- keep exact
  // semantics by scanning valid candidates in order against the running best,
  // seeded with `fill` when an invalid slot exists.
  float best;
  int bi = -1;
  rgb3[0] = rgb3[1] = rgb3[2] = 0.f;
  if (cnt < n_init) best = fill; else best = -INFINITY;
  for (int c = 0; c < cnt; c++) {
    float s = cand_sigma[off + c];
    if (nan_to_num && !isfinite(s)) s = 0.f;
    if (s > best || (bi < 0 && cnt >= n_init && c == 0)) { best = s; bi = c; }
  }
  if (bi >= 0) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      float v = cand_rgb[(size_t)(off + bi) * 3 + k];
      if (nan_to_num && !isfinite(v)) v = 0.f;
      rgb3[k] = v;
    }
  }
  sg = best;
}

__global__ __launch_bounds__(256) void k_candidate_max(const float *__restrict__ cand_rgb,
                                                       const float *__restrict__ cand_sigma,
                                                       const int32_t *__restrict__ pt_off,
                                                       const uint8_t *__restrict__ pt_cnt, int P,
                                                       const int32_t *__restrict__ n_pts_dev, int n_init,
                                                       float fill, int nan_to_num, float *__restrict__ rgb,
                                                       float *__restrict__ sigma, float *__restrict__ dmax) {
  if (n_pts_dev) P = min(P, *n_pts_dev);
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float sg, c[3];
  cand_max(cand_rgb, cand_sigma, pt_off[p], pt_cnt[p], n_init, fill, nan_to_num != 0, sg, c);
  if (sigma) sigma[p] = sg;
  if (rgb) { rgb[(size_t)p * 3] = c[0]; rgb[(size_t)p * 3 + 1] = c[1]; rgb[(size_t)p * 3 + 2] = c[2]; }
  if (dmax) dmax[p] = fmaxf(dmax[p], sg);  // density_grid.py:102 torch.maximum
}

// ---------------------------------------------------------------------------
// a16/a18 occupancy post-processing
// ---------------------------------------------------------------------------
struct OccWs {
  double sum;               // sum of the max-pooled field
  unsigned long long best;  // (count << 32) | (~label)
  float thr;                // clamp(mean, max = 0.01)
  float pad;
};

// Launch 1 of 5: f = 1 - exp(0.01 * -density) (density_grid.py:104) evaluated on the fly for the
// 27 neighbours, g = maxpool3(f), per-workgroup partial sums of g (fixed-order tree: the mean is
// deterministic); also resets what the later launches accumulate into (component counts,
// union-find parents, best-component word, border flag).
__global__ __launch_bounds__(256) void k_occ_pool(const float *__restrict__ density, int G,
                                                  float *__restrict__ pooled, double *__restrict__ partial,
                                                  int32_t *__restrict__ parent, int32_t *__restrict__ count,
                                                  OccWs *__restrict__ ws, uint32_t *__restrict__ bits) {
  const int n = G * G * G;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) { ws->sum = 0.0; ws->best = 0ull; bits[n >> 5] = 1u; }  // border flag: 1 = no border cell occupied
  float m = 0.f;
  if (i < n) {
    const int x = i / (G * G), y = i / G % G, z = i % G;
    m = -INFINITY;
    for (int a = -1; a <= 1; a++)
      for (int b = -1; b <= 1; b++)
        for (int c = -1; c <= 1; c++) {
          const int xx = x + a, yy = y + b, zz = z + c;
          if (xx < 0 || yy < 0 || zz < 0 || xx >= G || yy >= G || zz >= G) continue;
          const float f = 1.f - expf(0.01f * -density[(xx * G + yy) * G + zz]);
          m = (f > m || isnan(f)) ? f : m;
        }
    pooled[i] = m;
    parent[i] = i;   // every cell starts as its own root; unoccupied cells are never linked nor queried
    count[i] = 0;
  }
  // deterministic mean: fixed-order tree inside the workgroup, one partial per workgroup
  __shared__ double s_part[4];
  double v = (i < n) ? (double)m : 0.0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if (ia_lane() == 0) s_part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

// fixed-order reduction of the workgroup partials, evaluated identically by every workgroup that needs
// the threshold (256 threads): thr = clamp(mean, max = 0.01)  (density_grid.py:106)
__device__ __forceinline__ float occ_threshold(const double *__restrict__ partial, int n_part, int n, double *s /*[256]*/) {
  double v = 0.0;
  for (int k = threadIdx.x; k < n_part; k += 256) v += partial[k];
  s[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  const float mean = (float)(s[0] / (double)n);
  return mean > 0.01f ? 0.01f : mean;
}

__device__ __forceinline__ int uf_find(int32_t *parent, int i) {
  const int start = i;
  int p = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (p != i) { i = p; p = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  // path compression: parents only ever grow towards the (maximum-index) root, so a
  // monotone atomicMax can never undo a concurrent link
  if (start != i) atomicMax(&parent[start], i);
  return i;
}

// Launch 2 of 5: grid = g > thr, 26-connected union (max_pool3d 3x3x3 label propagation,
// density_grid.py:118-125): the surviving label of a component is its LARGEST linear index (+1), so
// roots are the maximum index (parent[i] >= i).
__global__ __launch_bounds__(256) void k_occ_union(int G, const float *__restrict__ pooled,
                                                   const double *__restrict__ partial, int n_part, OccWs *ws,
                                                   int32_t *parent) {
  __shared__ double s_red[256];
  const int n = G * G * G;
  const float thr = occ_threshold(partial, n_part, n, s_red);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) { ws->thr = thr; ws->sum = s_red[0]; }
  if (i >= n || !(pooled[i] > thr)) return;
  const int x = i / (G * G), y = i / G % G, z = i % G;
  // the 13 neighbours with a larger linear index
  for (int a = 0; a <= 1; a++)
    for (int b = -1; b <= 1; b++)
      for (int c = -1; c <= 1; c++) {
        if (a == 0 && (b < 0 || (b == 0 && c <= 0))) continue;
        const int xx = x + a, yy = y + b, zz = z + c;
        if (xx >= G || yy < 0 || yy >= G || zz < 0 || zz >= G) continue;
        const int j = (xx * G + yy) * G + zz;
        if (!(pooled[j] > thr)) continue;
        int ra = i, rb = j;
        while (true) {
          ra = uf_find(parent, ra); rb = uf_find(parent, rb);
          if (ra == rb) break;
          if (ra > rb) { const int t = ra; ra = rb; rb = t; }
          const int old = atomicCAS(&parent[ra], ra, rb);  // link smaller root under larger
          if (old == ra) break;
        }
      }
}

// ---- G = 64: the same components from z-RUNS instead of cells ---------------------------------------------
// A z-column of a 64^3 grid is exactly one wave.  k_occ_runs ballots the column into a 64-bit mask and points every
// occupied cell straight at the last cell of its contiguous run (the run's maximum index: no atomics, the in-column
// connectivity is done).  k_occ_union_runs then lets the HEAD of every run union its run with the runs it touches in
// the four forward neighbour columns ((x, y+1), (x+1, y-1), (x+1, y), (x+1, y+1); overlap in z dilated by one = 26-
// connectivity): ~8 000 unions between run ends with shallow trees instead of ~50 000 between cells, which is what
// the cell-based kernel above spends 40-145 us on (chains of device-scope atomics).
__global__ __launch_bounds__(256) void k_occ_runs(const float *__restrict__ pooled, const double *__restrict__ partial,
                                                  int n_part, OccWs *ws, unsigned long long *__restrict__ colmask,
                                                  int32_t *__restrict__ parent) {
  __shared__ double s_red[256];
  constexpr int G = 64, n = G * G * G;
  const float thr = occ_threshold(partial, n_part, n, s_red);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // wave = column (x, y), lane = z
  if (i == 0) { ws->thr = thr; ws->sum = s_red[0]; }
  const int lane = i & 63;
  const bool occ = pooled[i] > thr;
  const unsigned long long m = __ballot(occ);
  if (lane == 0) colmask[i >> 6] = m;
  if (occ) {
    // end of this cell's run: the first clear bit above `lane` (64 if none) minus one
    const unsigned long long above = ~m & ~((2ull << lane) - 1ull);  // clear bits strictly above lane (lane 63: none)
    const int end = (lane == 63 || above == 0ull) ? 63 : __ffsll((long long)above) - 2;
    parent[i] = (i & ~63) | end;
  }
}

__global__ __launch_bounds__(256) void k_occ_union_runs(const unsigned long long *__restrict__ colmask, int32_t *parent) {
  constexpr int G = 64;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int col = i >> 6, z = i & 63;
  const unsigned long long m = colmask[col];
  const bool head = ((m >> z) & 1ull) && (z == 0 || !((m >> (z - 1)) & 1ull));
  if (!head) return;
  const unsigned long long above = ~m & ~((2ull << z) - 1ull);
  const int z1 = (z == 63 || above == 0ull) ? 63 : __ffsll((long long)above) - 2;  // run = [z, z1]
  // window of neighbour cells touching the run: [z - 1, z1 + 1]
  const int lo = z > 0 ? z - 1 : 0, hi = z1 < 63 ? z1 + 1 : 63;
  const unsigned long long win = (hi == 63 ? ~0ull : ((2ull << hi) - 1ull)) & ~((1ull << lo) - 1ull);
  const int x = col >> 6, y = col & 63;
  const int self_end = (col << 6) | z1;
  const int nb[4][2] = {{0, 1}, {1, -1}, {1, 0}, {1, 1}};
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int xx = x + nb[k][0], yy = y + nb[k][1];
    if (xx >= G || yy < 0 || yy >= G) continue;
    const int ncol = xx * G + yy;
    const unsigned long long nm = colmask[ncol];
    unsigned long long t = nm & win;
    while (t) {
      const int b = __ffsll((long long)t) - 1;             // a touching cell of the neighbour column ...
      const unsigned long long nabove = ~nm & ~((2ull << b) - 1ull);
      const int e = (b == 63 || nabove == 0ull) ? 63 : __ffsll((long long)nabove) - 2;   // ... and the end of ITS run
      t &= ~((e == 63 ? ~0ull : ((2ull << e) - 1ull)));   // skip the rest of that run
      int ra = self_end, rb = (ncol << 6) | e;
      while (true) {
        ra = uf_find(parent, ra); rb = uf_find(parent, rb);
        if (ra == rb) break;
        if (ra > rb) { const int tt = ra; ra = rb; rb = tt; }
        const int old = atomicCAS(&parent[ra], ra, rb);  // link smaller root under larger
        if (old == ra) break;
      }
    }
  }
}

// Launch 3 of 5: labels + component sizes
__global__ __launch_bounds__(256) void k_occ_count(int G, const float *__restrict__ pooled, const OccWs *__restrict__ ws,
                                                   int32_t *parent, int32_t *__restrict__ label, int32_t *count) {
  const int n = G * G * G;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int l = -1;
  if (pooled[i] > ws->thr) l = uf_find(parent, i);
  label[i] = l;
  // most lanes of a wave belong to the same (largest) component: one atomic per wave then
  const unsigned long long m = __ballot(l >= 0);
  if (m) {
    const int first = __shfl(l, __ffsll((long long)m) - 1, 64);
    if (__all(l < 0 || l == first)) { if (ia_lane() == __ffsll((long long)m) - 1) atomicAdd(&count[first], __popcll(m)); }
    else if (l >= 0) atomicAdd(&count[l], 1);
  }
}

// torch.mode(mcc[field]): most frequent label, smallest label on ties (:109)
__global__ __launch_bounds__(256) void k_occ_best(int G, const int32_t *__restrict__ count, OccWs *ws) {
  const int n = G * G * G;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = count[i];
  if (c > 0) atomicMax(&ws->best, ((unsigned long long)c << 32) | (unsigned long long)(0xffffffffu - (uint32_t)i));
}

__global__ __launch_bounds__(256) void k_occ_final(int G, const int32_t *__restrict__ label,
                                                   const OccWs *__restrict__ ws, uint32_t *__restrict__ bits,
                                                   uint8_t *__restrict__ occ_bool) {
  const int n = G * G * G;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long best = ws->best;
  const int best_label = best ? (int)(0xffffffffu - (uint32_t)(best & 0xffffffffu)) : -2;
  const bool on = (i < n) && label[i] == best_label;
  if (i < n && occ_bool) occ_bool[i] = on;
  const unsigned long long m = __ballot(on);
  const int lane = ia_lane();
  if (i < n || (i - lane) < n) {
    if (lane == 0) bits[(i >> 5)] = (uint32_t)m;
    if (lane == 32) bits[(i >> 5)] = (uint32_t)(m >> 32);
  }
  // border flag (word n/32, preset to 1 by k_occ_reset_flag): cleared if a border cell is occupied
  if (on) {
    const int x = i / (G * G), y = i / G % G, z = i % G;
    if (x == 0 || y == 0 || z == 0 || x == G - 1 || y == G - 1 || z == G - 1) bits[n >> 5] = 0u;
  }
}

__global__ void k_occ_set_flag(uint32_t *bits, int n) { bits[n >> 5] = 1u; }

// Bounds of the occupied cells, for the marcher's exact empty-space skip (k_march_compact): the six words behind the border flag
// = {x_min, x_max, y_min, y_max, z_min, z_max} of the set bits (x_min > x_max when nothing is occupied).  One workgroup; the words
// of the bit grid are 32 KB.
#define IA_OCC_TAIL_WORDS 8   // flag + 6 bounds + 1 spare word behind the G^3 / 32 words of the bit grid
__global__ __launch_bounds__(1024) void k_occ_bounds(uint32_t *__restrict__ bits, int G) {
  const int n_words = (G * G * G) >> 5;
  int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, mx[3] = {-1, -1, -1};
  if ((G & (G - 1)) == 0 && G >= 32) {
    // power-of-two grids (the 64^3 of the configs): the 32 cells of a word share x and y, z = word offset + bit -- no per-bit loop, no
    // division; four words per 16-byte load, all loads of a thread issued before the first use
    const int sh = 31 - __clz(G);
    const uint4 *b4 = reinterpret_cast<const uint4 *>(bits);
    for (int q0 = threadIdx.x; q0 < n_words / 4; q0 += 2 * blockDim.x) {
      const int q1 = q0 + blockDim.x;
      const uint4 v0 = b4[q0];
      const uint4 v1 = q1 < n_words / 4 ? b4[q1] : make_uint4(0, 0, 0, 0);
      const uint32_t m[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
      for (int k = 0; k < 8; k++) {
        if (!m[k]) continue;
        const int i = ((k < 4 ? q0 : q1) * 4 + (k & 3)) * 32;
        const int x = i >> (2 * sh), y = (i >> sh) & (G - 1), z0 = i & (G - 1);
        mn[0] = min(mn[0], x); mx[0] = max(mx[0], x);
        mn[1] = min(mn[1], y); mx[1] = max(mx[1], y);
        mn[2] = min(mn[2], z0 + __ffs((int)m[k]) - 1); mx[2] = max(mx[2], z0 + 31 - __clz((int)m[k]));
      }
    }
  } else {
    for (int w = threadIdx.x; w < n_words; w += blockDim.x) {
      uint32_t m = bits[w];
      while (m) {
        const int b = __ffs((int)m) - 1;
        m &= m - 1u;
        const int i = w * 32 + b;
        const int c[3] = {i / (G * G), i / G % G, i % G};
#pragma unroll
        for (int d = 0; d < 3; d++) { mn[d] = min(mn[d], c[d]); mx[d] = max(mx[d], c[d]); }
      }
    }
  }
  __shared__ int s_mn[16][3], s_mx[16][3];
#pragma unroll
  for (int d = 0; d < 3; d++) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn[d] = min(mn[d], __shfl_xor(mn[d], o, 64)); mx[d] = max(mx[d], __shfl_xor(mx[d], o, 64)); }
  }
  const int wv = threadIdx.x >> 6;
  if (ia_lane() == 0) {
#pragma unroll
    for (int d = 0; d < 3; d++) { s_mn[wv][d] = mn[d]; s_mx[wv][d] = mx[d]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int d = threadIdx.x;
    int a = 0x7fffffff, b = -1;
    for (int k = 0; k < (int)(blockDim.x >> 6); k++) { a = min(a, s_mn[k][d]); b = max(b, s_mx[k][d]); }
    bits[n_words + 1 + 2 * d] = (uint32_t)a;
    bits[n_words + 2 + 2 * d] = (uint32_t)b;
  }
}

__global__ __launch_bounds__(256) void k_occ_pack(const uint8_t *__restrict__ occ_bool, int n, int G,
                                                  uint32_t *__restrict__ bits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool on = (i < n) && occ_bool[i] != 0;
  if (on) {
    const int x = i / (G * G), y = i / G % G, z = i % G;
    if (x == 0 || y == 0 || z == 0 || x == G - 1 || y == G - 1 || z == G - 1) bits[n >> 5] = 0u;
  }
  const unsigned long long m = __ballot(on);
  const int lane = ia_lane();
  if ((i - lane) < n) {
    if (lane == 0) bits[(i >> 5)] = (uint32_t)m;
    if (lane == 32 && i < ((n + 31) / 32) * 32) bits[(i >> 5)] = (uint32_t)(m >> 32);
  }
}

__global__ void k_fill_f32(float *p, float v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void k_fill_i32(int32_t *p, int32_t v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// probe points of DensityGrid.initialize (density_grid.py:100):
// coords = (idx/G + rand/G) * (aabb1 - aabb0) + aabb0
// Batched probes are laid out CELL-major: probe p is jitter set p % iters of cell p / iters, so the
// `iters` jittered points of one cell are neighbours in the point list (same workgroup, adjacent
// lanes): their solves start within one cell of each other and share transform-grid lines in L1.
// Enumeration of the cells: probe p belongs to cell ia_probe_cell(p / iters) of the occupancy grid's own index [x][y][z] (z
// fastest, raymarcher.cu:49-52).  IA_PROBE_ORDER_X = 0: in that order; 1: x fastest; 2: Morton.  The probe launch of the
// search is bound by its vector-L1 miss path (hit rate 71 % against 92 % in the render launches, profiles/r04_pmc_search.json),
// and which cells share a workgroup decides how many lines its lanes share.
// Only the ORDER of the points changes: every cell keeps its jitter and its probes, the density is bit-identical.
// Measured (profiles/r05_ab_probe.txt): x-fastest (1) +7 % on the probe launch, Morton (2) -2 %: a workgroup's 13 cells as a
// compact block share the most 128-byte lines (simulated unique lines per live solve 2.37 -> 2.05).
#ifndef IA_PROBE_ORDER_X
#define IA_PROBE_ORDER_X 2
#endif
__device__ __forceinline__ int ia_probe_cell(int c, int G) {
  if (IA_PROBE_ORDER_X == 0) return c;
  if (IA_PROBE_ORDER_X == 2 && G == 64) {   // Morton: bits of c dealt to z, y, x in turn -> a workgroup's 13 cells form a compact block
    int x = 0, y = 0, z = 0;
#pragma unroll
    for (int b = 0; b < 6; b++) {
      z |= ((c >> (3 * b)) & 1) << b;
      y |= ((c >> (3 * b + 1)) & 1) << b;
      x |= ((c >> (3 * b + 2)) & 1) << b;
    }
    return (x * G + y) * G + z;
  }
  const int x = c % G, y = c / G % G, z = c / (G * G);
  return (x * G + y) * G + z;
}

__global__ __launch_bounds__(256) void k_probe_points(const float *__restrict__ jitter, int G, int iters,
                                                      const float *__restrict__ aabb,
                                                      float *__restrict__ pts, int32_t *__restrict__ n_cand) {
  const int n = G * G * G;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;  // jitter is [iters][n cells][3]
  if (p == 0) *n_cand = 0;  // candidate counter of the following search
  if (p >= n * iters) return;
  const int it = p % iters, i = ia_probe_cell(p / iters, G);
  const int idx[3] = {i / (G * G), i / G % G, i % G};
#pragma unroll
  for (int d = 0; d < 3; d++) {
    const float c0 = (float)idx[d] / (float)G;
    const float c = c0 + jitter[((size_t)it * n + i) * 3 + d] / (float)G;
    pts[(size_t)p * 3 + d] = c * (aabb[3 + d] - aabb[d]) + aabb[d];
  }
}

// transform_rays_w2s (snarf_deformer.py:95-103)
__global__ __launch_bounds__(256) void k_transform_rays(const float *__restrict__ o, const float *__restrict__ d,
                                                        const float *__restrict__ w2s, int R,
                                                        float *__restrict__ o2, float *__restrict__ d2,
                                                        float *__restrict__ near, float *__restrict__ far) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R) return;
  const float ox = o[(size_t)i * 3], oy = o[(size_t)i * 3 + 1], oz = o[(size_t)i * 3 + 2];
  const float dx = d[(size_t)i * 3], dy = d[(size_t)i * 3 + 1], dz = d[(size_t)i * 3 + 2];
  float po[3], pd[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    po[a] = ox * w2s[a * 4] + oy * w2s[a * 4 + 1] + oz * w2s[a * 4 + 2] + w2s[a * 4 + 3];
    pd[a] = dx * w2s[a * 4] + dy * w2s[a * 4 + 1] + dz * w2s[a * 4 + 2];
  }
  const float dist = sqrtf(po[0] * po[0] + po[1] * po[1] + po[2] * po[2]);
#pragma unroll
  for (int a = 0; a < 3; a++) { o2[(size_t)i * 3 + a] = po[a]; d2[(size_t)i * 3 + a] = pd[a]; }
  near[i] = dist - 1; far[i] = dist + 1;
}

// ---------------------------------------------------------------------------
// fused render_test loop state (device resident).  ONE record per wave-front iteration: the
// kernels of iteration `it` work on st[it]; the alive compaction of iteration it - 1 counts into
// st[it].n_alive.  Every kernel derives the loop condition and the N_step schedule
// (raymarcher_acc.py:107-112) itself from (n_alive, k_before) -- a pure function -- so no
// single-thread "begin iteration" launch sits between two iterations; the march kernel's first
// thread records the derived values for the kernels that follow it and for the next iteration.
// ---------------------------------------------------------------------------
#define IA_RENDER_MAX_ITERS 1024
struct RenderState {
  int32_t n_alive;    // rays handed over by the previous iteration's compaction (R for iteration 0)
  int32_t n_samples;  // compact sample counter of this iteration
  int32_t n_cand;     // compact candidate counter of this iteration
  int32_t k_before;   // samples-per-ray budget consumed before this iteration (raymarcher_acc.py:107,127)
  int32_t iters;      // iterations before this one that had rays to process
  int32_t n_step;     // N_step of this iteration   } written by the march kernel's first thread,
  int32_t n_eff;      // rays actually processed     } read by the kernels behind it
  int32_t pad;
};

__device__ __forceinline__ void render_schedule(const RenderState *st, int max_samples, int max_batch, int &na, int &ns) {
  na = st->n_alive;
  if (st->k_before >= max_samples) na = 0;  // while k < MAX_SAMPLES
  ns = 0;
  if (na > 0) {
    ns = max_batch / na;
    ns = ns < max_samples ? ns : max_samples;
    ns = ns > 1 ? ns : 1;
  }
}

__global__ __launch_bounds__(256) void k_render_init_rays(int R, const float *__restrict__ near,
                                                          const float *__restrict__ far,
                                                          float *__restrict__ near_w, float *__restrict__ step,
                                                          float *__restrict__ color, float *__restrict__ depth,
                                                          float *__restrict__ nohit, float *__restrict__ counter,
                                                          int32_t *__restrict__ alive, int max_samples,
                                                          RenderState *st) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  // all iteration records zeroed, st[0].n_alive = R (first word)
  for (int w = i; w < IA_RENDER_MAX_ITERS * (int)(sizeof(RenderState) / 4); w += gridDim.x * blockDim.x)
    reinterpret_cast<int32_t *>(st)[w] = (w == 0) ? R : 0;
  if (i >= R) return;
  near_w[i] = near[i];
  step[i] = (far[i] - near[i]) / max_samples;  // raymarcher_acc.py:101
  color[(size_t)i * 3] = 0.f; color[(size_t)i * 3 + 1] = 0.f; color[(size_t)i * 3 + 2] = 0.f;
  depth[i] = 0.f; nohit[i] = 1.f; counter[i] = 0.f;
  alive[i] = i;
}

// march + sample compaction: every alive ray counts its samples (pass 1), the
// wave reserves a contiguous range with one atomic (prefix sum), pass 2 re-marches
// and writes positions + depths.  A ray's samples are contiguous and ordered.
// LDS_BITS: the 64^3 bit grid (32 KB) is staged in LDS by every workgroup that has a ray inside the box, and the
// occupancy tests of the march read it from there: a wave's 64 divergent 4-byte loads cost the vector L1 one look-up
// each (the same limit k_search runs into), the LDS serves them in a few clocks.
#define IA_MARCH_LDS_WORDS (64 * 64 * 64 / 32)
template <bool LDS_BITS>
__global__ __launch_bounds__(256) void k_march_compact(
    const float *__restrict__ rays_o, const float *__restrict__ rays_d, float *near_w,
    const float *__restrict__ fars, const float *__restrict__ step, const int32_t *__restrict__ alive,
    RenderState *st, const uint32_t *bits_global, int G, const float *__restrict__ aabb,
    float *__restrict__ s_pts, float *__restrict__ s_t, int32_t *__restrict__ ray_off,
    int32_t *__restrict__ ray_cnt, float *__restrict__ counter, int sample_cap, int max_samples, int max_batch) {
  int n_alive, N_steps;
  render_schedule(st, max_samples, max_batch, n_alive, N_steps);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {  // the derived schedule, for the kernels behind this one and for the next iteration
    st->n_eff = n_alive; st->n_step = N_steps;
    st[1].k_before = st->k_before + N_steps;
    st[1].iters = st->iters + (n_alive > 0 ? 1 : 0);
  }
  if ((i - ia_lane()) >= n_alive) return;  // whole wave idle (exited waves do not take part in the barriers below)
  const bool live = i < n_alive;
  int cnt = 0;
  size_t n = 0;
  MarchRay r;
  float t0 = 0.f, t_end = 0.f, t_start = 0.f;
  if (live) {
    n = (size_t)alive[i];
    r = load_ray(rays_o, rays_d, fars, step, n, aabb, aabb + 3, G);
    t_start = near_w[n];
    t_end = t_start;
  }
  // Exact empty-space skip (round 6).  The occupancy post-process keeps a flag word behind the bit grid: 1 = no BORDER cell is
  // occupied (k_occ_final / k_occ_pack), and behind it the index bounds of the occupied cells (k_occ_bounds).  The cell of a step
  // is clamped (raymarcher.cu:49-51), so a point outside the grid is tested against a border cell -- with the flag set those tests
  // all fail, and a step can only be occupied while the UNCLAMPED grid coordinates of the point lie inside the box of the occupied
  // cells.  [t_lo, t_hi] bounds that part of the ray (slab test in grid coordinates, the box grown by 0.05 cell and the interval
  // by two steps: the rounding of the slab arithmetic is ~1e-4 cell); outside it t advances by the SAME sequence of float adds
  // without the occupancy arithmetic (2 instead of ~25 VALU operations per step).  The grid spans the box of the deformed voxels
  // (2.8 x 2.8 x 1.5 m), the body fills 2 % of its cells: most rays of iteration 0 -- every pixel marches from near to its first
  // hit or to far -- never come near an occupied cell.
  float t_lo = -INFINITY, t_hi = INFINITY;
  const uint32_t *tail = bits_global + ((G * G * G) >> 5);
  if (live && tail[0] == 1u) {
    const float lo[3] = {(float)(int)tail[1] - 0.05f, (float)(int)tail[3] - 0.05f, (float)(int)tail[5] - 0.05f};
    const float hi[3] = {(float)((int)tail[2] + 1) + 0.05f, (float)((int)tail[4] + 1) + 0.05f, (float)((int)tail[6] + 1) + 0.05f};
    const float A[3] = {(r.ox - r.cx) * r.sx, (r.oy - r.cy) * r.sy, (r.oz - r.cz) * r.sz};
    const float B[3] = {r.dx * r.sx, r.dy * r.sy, r.dz * r.sz};
    float a = -INFINITY, b = INFINITY;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      if (!(lo[c] <= hi[c])) { a = INFINITY; b = -INFINITY; }               // nothing occupied at all
      else if (fabsf(B[c]) < 1e-12f) {
        if (!(A[c] >= lo[c] && A[c] <= hi[c])) { a = INFINITY; b = -INFINITY; }   // parallel to the slab and outside it: never inside
      } else {
        const float t1 = (lo[c] - A[c]) / B[c], t2 = (hi[c] - A[c]) / B[c];
        a = fmaxf(a, fminf(t1, t2));
        b = fminf(b, fmaxf(t1, t2));
      }
    }
    if (a <= b) { t_lo = a - 2.f * r.dt; t_hi = b + 2.f * r.dt; }
    else { t_lo = INFINITY; t_hi = -INFINITY; }
  }
  __shared__ uint32_t s_bits[LDS_BITS ? IA_MARCH_LDS_WORDS : 1];
  __shared__ int s_any;
  if (LDS_BITS) {
    // "does anybody in this workgroup march?" by hand: the idle waves at the end of the workgroup have exited, and a
    // library work-group reduction (__syncthreads_or) may count on every wave of the workgroup.  Wave 0 is present
    // whenever any wave is.
    if (threadIdx.x == 0) s_any = 0;
    __syncthreads();
    if (__any(live && t_start < r.far && t_lo <= t_hi && t_start <= t_hi) && ia_lane() == 0) s_any = 1;
    __syncthreads();
    if (s_any) {
      const uint4 *src = reinterpret_cast<const uint4 *>(bits_global);
      uint4 *dst = reinterpret_cast<uint4 *>(s_bits);
      // only the waves that did not exit above take part: stride = their thread count, not blockDim
      const int present = min(256, ((n_alive - (int)(blockIdx.x * blockDim.x)) + 63) & ~63);
      for (int w = threadIdx.x; w < IA_MARCH_LDS_WORDS / 4; w += present) dst[w] = src[w];
    }
    __syncthreads();
  }
  const uint32_t *bits = LDS_BITS ? s_bits : bits_global;
  // (a ray with far <= near has step <= 0: the reference loop would never end on it -- such a ray takes no sample here)
  if (live && r.dt > 0.f) {
    float t = t_start;
    t0 = t;
    bool found = false;
    // The reference loop (raymarcher.cu:44-69)
    //     while (t < far && cnt < N_steps) { if (occupied(t)) cnt++; t += dt; }
    // four steps at a time: the occupancy of a step does not depend on the previous step's, so the four
    // (always in-range: coordinates are clamped) bit loads are issued together and only the bookkeeping is
    // sequential.  t advances by the same sequence of float adds; exits happen at exactly the same step.
    march_skip(t, r.dt, fminf(r.far, t_lo));   // before the box: no step can be occupied
    while (t < r.far && cnt < N_steps && t <= t_hi) {
      const float t1 = t + r.dt, t2 = t1 + r.dt, t3 = t2 + r.dt;
      float x, y, z;
      const bool o0 = march_occupied(r, bits, G, t, x, y, z), o1 = march_occupied(r, bits, G, t1, x, y, z);
      const bool o2 = march_occupied(r, bits, G, t2, x, y, z), o3 = march_occupied(r, bits, G, t3, x, y, z);
      if (o0) { if (!found) { found = true; t0 = t; } cnt++; }  // t0: exact depth of the first hit
      t = t1;
      if (!(t < r.far && cnt < N_steps)) break;
      if (o1) { if (!found) { found = true; t0 = t; } cnt++; }
      t = t2;
      if (!(t < r.far && cnt < N_steps)) break;
      if (o2) { if (!found) { found = true; t0 = t; } cnt++; }
      t = t3;
      if (!(t < r.far && cnt < N_steps)) break;
      if (o3) { if (!found) { found = true; t0 = t; } cnt++; }
      t = t3 + r.dt;
    }
    if (cnt < N_steps) march_skip(t, r.dt, r.far);   // behind the box (cnt cannot change any more)
    t_end = t;
  }
  int total;
  const int excl = ia_wave_excl_scan(cnt, total);
  int base = 0;
  if (ia_lane() == 0 && total > 0) base = atomicAdd(&st->n_samples, total);
  base = __shfl(base, 0, 64) + excl;
  if (!live) return;
  ray_off[i] = base;
  ray_cnt[i] = cnt;
  counter[n] += (float)cnt;  // raymarcher_acc.py:116
  near_w[n] = t_end;         // raymarcher.cu:72
  // second pass: re-march from the first hit (same float sequence t0, t0+dt, ...) and write; again four
  // occupancy tests in flight per round
  int s = 0;
  float t = t0;
  auto emit = [&](float tt, float x, float y, float z) {
    const int o = base + s;
    if (o < sample_cap) { s_pts[(size_t)o * 3] = x; s_pts[(size_t)o * 3 + 1] = y; s_pts[(size_t)o * 3 + 2] = z; s_t[o] = tt; }
    s++;
  };
  while (t < r.far && s < cnt) {
    const float t1 = t + r.dt, t2 = t1 + r.dt, t3 = t2 + r.dt;
    float x0, y0, z0, x1, y1, z1, x2, y2, z2, x3, y3, z3;
    const bool o0 = march_occupied(r, bits, G, t, x0, y0, z0), o1 = march_occupied(r, bits, G, t1, x1, y1, z1);
    const bool o2 = march_occupied(r, bits, G, t2, x2, y2, z2), o3 = march_occupied(r, bits, G, t3, x3, y3, z3);
    if (o0) emit(t, x0, y0, z0);
    if (!(t1 < r.far && s < cnt)) break;
    if (o1) emit(t1, x1, y1, z1);
    if (!(t2 < r.far && s < cnt)) break;
    if (o2) emit(t2, x2, y2, z2);
    if (!(t3 < r.far && s < cnt)) break;
    if (o3) emit(t3, x3, y3, z3);
    t = t3 + r.dt;
  }
}

// composite + alive compaction (raymarcher_acc.py:118-127)
__global__ __launch_bounds__(256) void k_composite_compact(
    const int32_t *__restrict__ alive, int32_t *__restrict__ alive_next, RenderState *st,
    const int32_t *__restrict__ ray_off, const int32_t *__restrict__ ray_cnt,
    const float *__restrict__ s_t, const float *__restrict__ step, const int32_t *__restrict__ pt_off,
    const uint8_t *__restrict__ pt_cnt, const float *__restrict__ cand_rgb,
    const float *__restrict__ cand_sigma, int n_init, float *color, float *depth, float *nohit,
    float thresh) {
  const int n_alive = st->n_eff;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if ((i - ia_lane()) >= n_alive) return;
  const int N_steps = st->n_step;
  bool keep = false;
  int n = 0;
  if (i < n_alive) {
    n = alive[i];
    const int off = ray_off[i], cnt = ray_cnt[i];
    const float dt = step[n];
    float T = nohit[n];
    float c0 = color[(size_t)n * 3], c1 = color[(size_t)n * 3 + 1], c2 = color[(size_t)n * 3 + 2];
    float dep = depth[n];
    int s = 0;
    // `delta > 0` (raymarcher.cu:218) == slot filled; dt > 0 for filled slots.
    // Four samples per trip (round 6): the chain sample -> candidate range -> candidate densities -> winner's colour is three
    // dependent loads deep and the compositing is sequential in T only, so the loads of four samples are issued together (the
    // candidate scan runs over the four ranges in lock-step) and the four compositing steps follow in order, each behind the
    // reference's `T > 1e-4` test -- the same arithmetic in the same order as one sample per trip; a ray that terminates inside a
    // group has loaded up to three samples for nothing.
    while (s < cnt && (double)T > 1e-4 && dt > 0) {
      const int m = min(4, cnt - s);
      int po[4], pc[4], bi[4];
      float tz[4], best[4], rgb[4][3];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int q = off + s + (j < m ? j : 0);
        po[j] = pt_off[q]; pc[j] = j < m ? (int)pt_cnt[q] : 0; tz[j] = s_t[q];
        best[j] = pc[j] < n_init ? 0.f : -INFINITY;   // cand_max(fill = 0, nan_to_num = true), see there
        bi[j] = -1;
      }
      const int cmax = max(max(pc[0], pc[1]), max(pc[2], pc[3]));
      for (int c = 0; c < cmax; c++) {
        float sv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) sv[j] = c < pc[j] ? cand_sigma[po[j] + c] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if (c < pc[j]) {
            float v = sv[j];
            if (!isfinite(v)) v = 0.f;
            if (v > best[j] || (bi[j] < 0 && pc[j] >= n_init && c == 0)) { best[j] = v; bi[j] = c; }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
          float v = bi[j] >= 0 ? cand_rgb[(size_t)(po[j] + bi[j]) * 3 + k] : 0.f;
          if (!isfinite(v)) v = 0.f;
          rgb[j][k] = v;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (j < m && (double)T > 1e-4) composite_step(best[j], dt, tz[j], rgb[j], thresh, T, c0, c1, c2, dep);
      }
      s += m;
    }
    color[(size_t)n * 3] = c0; color[(size_t)n * 3 + 1] = c1; color[(size_t)n * 3 + 2] = c2;
    depth[n] = dep;
    nohit[n] = T;
    // alive = alive[(no_hit > 1e-4) & (z_new[:, -1] > 0)]  (raymarcher_acc.py:126)
    const bool last_filled = (cnt == N_steps) && (s_t[off + cnt - 1] > 0.f);
    keep = ((double)T > 1e-4) && last_filled;
  }
  const unsigned long long m = __ballot(keep);
  const int total = __popcll(m);
  int base = 0;
  if (ia_lane() == 0 && total > 0) base = atomicAdd(&st[1].n_alive, total);  // hand-over to the next iteration
  base = __shfl(base, 0, 64);
  if (keep) alive_next[base + __popcll(m & ((1ull << ia_lane()) - 1ull))] = n;
}

__global__ __launch_bounds__(256) void k_render_finalize(int R, const float *__restrict__ color,
                                                         const float *__restrict__ depth,
                                                         const float *__restrict__ nohit,
                                                         const float *__restrict__ counter,
                                                         const float *__restrict__ bg, const RenderState *st_end,
                                                         int max_samples, int32_t *__restrict__ n_alive_out,
                                                         float *__restrict__ rgb_out, float *__restrict__ depth_out,
                                                         float *__restrict__ alpha_out,
                                                         float *__restrict__ counter_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && n_alive_out) {
    // st_end = record of the first iteration NOT enqueued: would the loop continue?
    int na = st_end->n_alive;
    if (st_end->k_before >= max_samples) na = 0;
    n_alive_out[0] = na; n_alive_out[1] = st_end->iters;  // [1]: iterations that had rays to process
  }
  if (i >= R) return;
  const float T = nohit[i];
#pragma unroll
  for (int c = 0; c < 3; c++)  // raymarcher_acc.py:128-132
    rgb_out[(size_t)i * 3 + c] = color[(size_t)i * 3 + c] + T * (bg ? bg[(size_t)i * 3 + c] : 1.0f);
  depth_out[i] = depth[i];
  alpha_out[i] = 1.f - T;
  counter_out[i] = counter[i];
}

// ---------------------------------------------------------------------------
// host helpers
// ---------------------------------------------------------------------------
static OccDev make_occ(const ia_occ_grid *o) {
  OccDev d;
  d.G = o->G;
  for (int i = 0; i < 3; i++) { d.mn[i] = o->aabb_min[i]; d.mx[i] = o->aabb_max[i]; }
  return d;
}

extern "C" int ia_raymarch_test(const float *rays_o, const float *rays_d, float *nears, const float *fars,
                                const int64_t *alive, int n_alive, const uint32_t *occ_bits,
                                const ia_occ_grid *occ, const float *step_size, int N_steps, float *pts,
                                float *deltas, float *depths, void *stream) {
  IA_CHECK_ARG(n_alive >= 0 && N_steps > 0, "ia_raymarch_test: bad sizes");
  if (n_alive == 0) return IA_OK;
  IA_CHECK_ARG(rays_o && rays_d && nears && fars && alive && occ_bits && occ && step_size && pts && deltas && depths,
               "ia_raymarch_test: null pointer");
  hipLaunchKernelGGL(k_raymarch_test, dim3(ia_div_up(n_alive, 256)), dim3(256), 0, (hipStream_t)stream, rays_o,
                     rays_d, nears, fars, alive, n_alive, occ_bits, make_occ(occ), step_size, N_steps, pts, deltas,
                     depths);
  IA_LAUNCH_CHECK("k_raymarch_test");
  return IA_OK;
}

extern "C" int ia_raymarch_train(const float *rays_o, const float *rays_d, const float *nears,
                                 const float *fars, int n_rays, const uint32_t *occ_bits,
                                 const ia_occ_grid *occ, const float *step_size, int N_steps, float *depths,
                                 void *stream) {
  IA_CHECK_ARG(n_rays >= 0 && N_steps > 0, "ia_raymarch_train: bad sizes");
  if (n_rays == 0) return IA_OK;
  IA_CHECK_ARG(rays_o && rays_d && nears && fars && occ_bits && occ && step_size && depths,
               "ia_raymarch_train: null pointer");
  hipLaunchKernelGGL(k_raymarch_train, dim3(ia_div_up(n_rays, 256)), dim3(256), 0, (hipStream_t)stream, rays_o,
                     rays_d, nears, fars, n_rays, occ_bits, make_occ(occ), step_size, N_steps, depths);
  IA_LAUNCH_CHECK("k_raymarch_train");
  return IA_OK;
}

extern "C" int ia_composite_test(const float *rgb, const float *sigma, const float *delta, const float *depth,
                                 const int64_t *alive, int n_alive, int N_steps, float *color,
                                 float *depth_out, float *no_hit, float thresh, void *stream) {
  IA_CHECK_ARG(n_alive >= 0 && N_steps > 0, "ia_composite_test: bad sizes");
  if (n_alive == 0) return IA_OK;
  IA_CHECK_ARG(rgb && sigma && delta && depth && alive && color && depth_out && no_hit,
               "ia_composite_test: null pointer");
  hipLaunchKernelGGL(k_composite_test, dim3(ia_div_up(n_alive, 256)), dim3(256), 0, (hipStream_t)stream, rgb,
                     sigma, delta, depth, alive, n_alive, N_steps, color, depth_out, no_hit, thresh);
  IA_LAUNCH_CHECK("k_composite_test");
  return IA_OK;
}

extern "C" int ia_candidate_max(const float *cand_rgb, const float *cand_sigma, const int32_t *pt_off,
                                const uint8_t *pt_cnt, int P, const int32_t *n_pts_dev, int n_init, float fill,
                                int nan_to_num, float *rgb, float *sigma, void *stream) {
  IA_CHECK_ARG(P >= 0, "ia_candidate_max: P < 0");
  if (P == 0) return IA_OK;
  IA_CHECK_ARG(cand_rgb && cand_sigma && pt_off && pt_cnt, "ia_candidate_max: null pointer");
  hipLaunchKernelGGL(k_candidate_max, dim3(ia_div_up(P, 256)), dim3(256), 0, (hipStream_t)stream, cand_rgb,
                     cand_sigma, pt_off, pt_cnt, P, n_pts_dev, n_init, fill, nan_to_num, rgb, sigma,
                     (float *)nullptr);
  IA_LAUNCH_CHECK("k_candidate_max");
  return IA_OK;
}

// ---- occupancy -------------------------------------------------------------
extern "C" size_t ia_occupancy_workspace_bytes(int G) {
  const size_t n = (size_t)G * G * G;
  return ia_align(sizeof(OccWs)) + 5 * ia_align(n * 4) + ia_align((n / 256 + 1) * 8) + 1024;
}

extern "C" int ia_occupancy_from_density(const float *density, int G, uint32_t *occ_bits, uint8_t *occ_bool,
                                         void *ws, size_t ws_bytes, void *stream) {
  IA_CHECK_ARG(density && occ_bits && ws, "ia_occupancy_from_density: null pointer");
  IA_CHECK_ARG(G >= 4 && G <= 256 && (G * G * G) % 64 == 0, "ia_occupancy_from_density: unsupported G=%d", G);
  if (ws_bytes < ia_occupancy_workspace_bytes(G)) return ia_set_error(IA_ERR_WORKSPACE, "ia_occupancy_from_density: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int n = G * G * G;
  WsCarver w(ws, ws_bytes);
  OccWs *ow = w.take<OccWs>(1);
  float *pooled = w.take<float>(n);
  int32_t *parent = w.take<int32_t>(n);
  int32_t *label = w.take<int32_t>(n);
  int32_t *count = w.take<int32_t>(n);
  float *fval = w.take<float>(n);
  double *partial = w.take<double>(ia_div_up(n, 256));
  const dim3 grid(ia_div_up(n, 256)), blk(256);
  // (`fval` holds no f values any more -- f is evaluated inside the pooling kernel; its n floats serve as scratch)
  const int n_part = ia_div_up(n, 256);
  hipLaunchKernelGGL(k_occ_pool, grid, blk, 0, s, density, G, pooled, partial, parent, count, ow, occ_bits);
  if (G == 64) {  // z-runs: one wave per column (colmask: 4 096 x 64 bit, carved from `label`, which k_occ_count fills later)
    unsigned long long *colmask = reinterpret_cast<unsigned long long *>(fval);
    hipLaunchKernelGGL(k_occ_runs, grid, blk, 0, s, pooled, partial, n_part, ow, colmask, parent);
    hipLaunchKernelGGL(k_occ_union_runs, grid, blk, 0, s, colmask, parent);
  } else {
    hipLaunchKernelGGL(k_occ_union, grid, blk, 0, s, G, pooled, partial, n_part, ow, parent);
  }
  hipLaunchKernelGGL(k_occ_count, grid, blk, 0, s, G, pooled, ow, parent, label, count);
  hipLaunchKernelGGL(k_occ_best, grid, blk, 0, s, G, count, ow);
  hipLaunchKernelGGL(k_occ_final, grid, blk, 0, s, G, label, ow, occ_bits, occ_bool);
  hipLaunchKernelGGL(k_occ_bounds, dim3(1), dim3(1024), 0, s, occ_bits, G);
  IA_LAUNCH_CHECK("occupancy_from_density");
  return IA_OK;
}

extern "C" int ia_occupancy_pack(const uint8_t *occ_bool, int G, uint32_t *occ_bits, void *stream) {
  IA_CHECK_ARG(occ_bool && occ_bits && G > 0 && (G * G * G) % 64 == 0, "ia_occupancy_pack: bad arguments");
  const int n = G * G * G;
  hipLaunchKernelGGL(k_occ_set_flag, dim3(1), dim3(1), 0, (hipStream_t)stream, occ_bits, n);
  hipLaunchKernelGGL(k_occ_pack, dim3(ia_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, occ_bool, n, G, occ_bits);
  hipLaunchKernelGGL(k_occ_bounds, dim3(1), dim3(1024), 0, (hipStream_t)stream, occ_bits, G);
  IA_LAUNCH_CHECK("k_occ_pack");
  return IA_OK;
}

// ---- fused deformer query ----------------------------------------------------
struct QueryWs {
  int32_t *n_cand; int32_t *pt_off; uint8_t *pt_cnt; float *cand_xc; float *cand_rgb; float *cand_sigma;
  int cand_cap;
};
static size_t query_ws_bytes(int P, int n_init) {
  const size_t cap = (size_t)P * n_init;
  return ia_align(256) + ia_align((size_t)P * 4) + ia_align((size_t)P) + 2 * ia_align(cap * 12) + ia_align(cap * 4);
}
static QueryWs carve_query(WsCarver &w, int P, int n_init) {
  QueryWs q;
  const size_t cap = (size_t)P * n_init;
  q.n_cand = w.take<int32_t>(64);
  q.pt_off = w.take<int32_t>(P);
  q.pt_cnt = w.take<uint8_t>(P);
  q.cand_xc = w.take<float>(cap * 3);
  q.cand_rgb = w.take<float>(cap * 3);
  q.cand_sigma = w.take<float>(cap);
  q.cand_cap = (int)cap;
  return q;
}

extern "C" size_t ia_query_workspace_bytes(int P, int n_init) { return query_ws_bytes(P, n_init) + 1024; }

static int query_impl(const float *pts, int P, const int32_t *n_pts_dev, const float *voxel_J, const float *tfs,
                      const int32_t *bone_ids, int n_init, const ia_snarf_grid *grid, const FieldDev &F,
                      const QueryWs &q, hipStream_t s, int zero_counter = 1) {
  int rc = ia_snarf_search_compact(pts, P, n_pts_dev, voxel_J, tfs, bone_ids, n_init, grid, 1e-5f, 1e-1f,
                                   q.cand_xc, q.cand_cap, q.pt_off, q.pt_cnt, q.n_cand, zero_counter, s);
  if (rc) return rc;
  return ia_launch_field(q.cand_xc, q.cand_cap, q.n_cand, F, q.cand_rgb, q.cand_sigma, s, nullptr);
}

extern "C" int ia_deform_query(const float *pts, int P, const int32_t *n_pts_dev, const float *voxel_J,
                               const float *tfs, const int32_t *bone_ids, int n_init, const ia_snarf_grid *grid,
                               const ia_field *field, float *rgb, float *sigma, float *dmax, void *ws,
                               size_t ws_bytes, void *stream) {
  IA_CHECK_ARG(P >= 0, "ia_deform_query: P < 0");
  if (P == 0) return IA_OK;
  IA_CHECK_ARG(pts && ws, "ia_deform_query: null pointer");
  if (ws_bytes < ia_query_workspace_bytes(P, n_init)) return ia_set_error(IA_ERR_WORKSPACE, "ia_deform_query: workspace too small");
  FieldDev F;
  int rc = ia_make_field_dev(field, &F);
  IA_CHECK_ARG(rc == 0, "ia_deform_query: bad field descriptor (%d)", rc);
  WsCarver w(ws, ws_bytes);
  QueryWs q = carve_query(w, P, n_init);
  hipStream_t s = (hipStream_t)stream;
  rc = query_impl(pts, P, n_pts_dev, voxel_J, tfs, bone_ids, n_init, grid, F, q, s);
  if (rc) return rc;
  hipLaunchKernelGGL(k_candidate_max, dim3(ia_div_up(P, 256)), dim3(256), 0, s, q.cand_rgb, q.cand_sigma, q.pt_off,
                     q.pt_cnt, P, n_pts_dev, n_init, 0.f, 1, rgb, sigma, dmax);
  IA_LAUNCH_CHECK("k_candidate_max");
  return IA_OK;
}

// ---- fused DensityGrid.initialize -------------------------------------------
extern "C" size_t ia_density_init_workspace_bytes(int G, int n_init) {
  const int n = G * G * G;
  return ia_align((size_t)n * 12) + query_ws_bytes(n, n_init) + ia_occupancy_workspace_bytes(G) + 4096;
}

// workspace for probing all `iters` jittered point sets in ONE search / field launch
extern "C" size_t ia_density_init_workspace_bytes_batched(int G, int n_init, int iters) {
  const size_t n = (size_t)G * G * G * (size_t)(iters > 0 ? iters : 1);
  if (n * (size_t)n_init > 0x7fffffffu) return 0;  // does not fit the 32-bit candidate indices: not available
  return ia_align(n * 12) + query_ws_bytes((int)n, n_init) + ia_occupancy_workspace_bytes(G) + 4096;
}

// max over the probe sets of the per-point candidate max (density_grid.py:98-102: density starts at
// 0 and is torch.maximum-ed with every set -- max is order independent)
__global__ __launch_bounds__(256) void k_probe_max(const float *__restrict__ cand_rgb,
                                                   const float *__restrict__ cand_sigma,
                                                   const int32_t *__restrict__ pt_off,
                                                   const uint8_t *__restrict__ pt_cnt, int n, int G, int iters, int n_init,
                                                   float *__restrict__ density, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  const int i = ia_probe_cell(c, G);
  float m = accumulate ? density[i] : 0.f;   // accumulate: one probe set per launch (the small-workspace route)
  for (int it = 0; it < iters; it++) {
    const size_t p = (size_t)c * iters + it;  // cell-major probe order (k_probe_points)
    float sg, col[3];
    cand_max(cand_rgb, cand_sigma, pt_off[p], pt_cnt[p], n_init, 0.f, true, sg, col);
    m = fmaxf(m, sg);
  }
  density[i] = m;
}

extern "C" int ia_density_grid_init(const float *jitter, int iters, int G, const float *aabb,
                                    const float *voxel_J, const float *tfs, const int32_t *bone_ids, int n_init,
                                    const ia_snarf_grid *grid, const ia_field *field, float *density,
                                    uint32_t *occ_bits, uint8_t *occ_bool, void *ws, size_t ws_bytes,
                                    void *stream) {
  IA_CHECK_ARG(jitter && aabb && density && occ_bits && ws && iters >= 1, "ia_density_grid_init: bad arguments");
  if (ws_bytes < ia_density_init_workspace_bytes(G, n_init)) return ia_set_error(IA_ERR_WORKSPACE, "ia_density_grid_init: workspace too small");
  FieldDev F;
  int rc = ia_make_field_dev(field, &F);
  IA_CHECK_ARG(rc == 0, "ia_density_grid_init: bad field descriptor (%d)", rc);
  hipStream_t s = (hipStream_t)stream;
  const int n = G * G * G;
  const dim3 grd(ia_div_up(n, 256)), blk(256);
  const size_t need_batched = ia_density_init_workspace_bytes_batched(G, n_init, iters);
  if (need_batched && ws_bytes >= need_batched) {
    // The probe sets are independent: one launch over iters x G^3 points instead of `iters` launches
    // whose sparse tails (most probe points lie in empty space) cannot overlap.
    const int n_all = n * iters;
    WsCarver w(ws, ws_bytes);
    float *pts = w.take<float>((size_t)n_all * 3);
    QueryWs q = carve_query(w, n_all, n_init);
    void *occ_ws = w.take<char>(ia_occupancy_workspace_bytes(G));
    hipLaunchKernelGGL(k_probe_points, dim3(ia_div_up(n_all, 256)), blk, 0, s, jitter, G, iters, aabb, pts, q.n_cand);
    rc = query_impl(pts, n_all, nullptr, voxel_J, tfs, bone_ids, n_init, grid, F, q, s, 0);
    if (rc) return rc;
    hipLaunchKernelGGL(k_probe_max, grd, blk, 0, s, q.cand_rgb, q.cand_sigma, q.pt_off, q.pt_cnt, n, G, iters, n_init, density, 0);
    IA_LAUNCH_CHECK("density_grid_init");
    return ia_occupancy_from_density(density, G, occ_bits, occ_bool, occ_ws, ia_occupancy_workspace_bytes(G), s);
  }
  WsCarver w(ws, ws_bytes);
  float *pts = w.take<float>((size_t)n * 3);
  QueryWs q = carve_query(w, n, n_init);
  void *occ_ws = w.take<char>(ia_occupancy_workspace_bytes(G));
  hipLaunchKernelGGL(k_fill_f32, grd, blk, 0, s, density, 0.f, n);  // density_grid.py:98
  for (int it = 0; it < iters; it++) {
    hipLaunchKernelGGL(k_probe_points, grd, blk, 0, s, jitter + (size_t)it * n * 3, G, 1, aabb, pts, q.n_cand);
    rc = query_impl(pts, n, nullptr, voxel_J, tfs, bone_ids, n_init, grid, F, q, s, 0);
    if (rc) return rc;
    // (point p of k_probe_points(iters = 1) is cell ia_probe_cell(p), not cell p: the maximum must land there -- ADVICE r05)
    hipLaunchKernelGGL(k_probe_max, grd, blk, 0, s, q.cand_rgb, q.cand_sigma, q.pt_off, q.pt_cnt, n, G, 1, n_init, density, 1);
  }
  IA_LAUNCH_CHECK("density_grid_init");
  return ia_occupancy_from_density(density, G, occ_bits, occ_bool, occ_ws, ia_occupancy_workspace_bytes(G), s);
}

// ---- fused render_test -------------------------------------------------------
struct RenderWs {
  RenderState *st; float *near_w, *step, *color, *depth, *nohit, *counter;
  int32_t *alive_a, *alive_b, *ray_off, *ray_cnt; float *s_pts, *s_t; QueryWs q; int sample_cap;
};
static size_t render_ws_bytes(int R, int max_batch, int n_init) {
  const size_t cap = (size_t)(R > max_batch ? R : max_batch);
  return ia_align(sizeof(RenderState) * IA_RENDER_MAX_ITERS) + 4 * ia_align((size_t)R * 4) + ia_align((size_t)R * 12) + ia_align((size_t)R * 4) +
         4 * ia_align((size_t)R * 4) + ia_align(cap * 12) + ia_align(cap * 4) + query_ws_bytes((int)cap, n_init);
}
extern "C" size_t ia_render_workspace_bytes(int R, int max_batch, int n_init) {
  return render_ws_bytes(R, max_batch, n_init) + 4096;
}

extern "C" int ia_render_test(const float *rays_o, const float *rays_d, const float *near, const float *far, int R,
                              const float *bg, const uint32_t *occ_bits, int G, const float *aabb,
                              const float *voxel_J, const float *tfs, const int32_t *bone_ids, int n_init,
                              const ia_snarf_grid *grid, const ia_field *field, int max_samples, int max_batch,
                              int n_iters, int resume, float *rgb, float *depth, float *alpha, float *counter,
                              int32_t *n_alive_out, void *ws, size_t ws_bytes, void *stream) {
  IA_CHECK_ARG(R > 0 && max_samples > 0 && max_batch > 0 && n_iters >= 0, "ia_render_test: bad sizes");
  IA_CHECK_ARG(rays_o && rays_d && near && far && occ_bits && aabb && voxel_J && tfs && rgb && depth && alpha && counter && ws,
               "ia_render_test: null pointer");
  if (ws_bytes < ia_render_workspace_bytes(R, max_batch, n_init)) return ia_set_error(IA_ERR_WORKSPACE, "ia_render_test: workspace too small");
  FieldDev F;
  int rc = ia_make_field_dev(field, &F);
  IA_CHECK_ARG(rc == 0, "ia_render_test: bad field descriptor (%d)", rc);
  hipStream_t s = (hipStream_t)stream;
  const int cap = R > max_batch ? R : max_batch;
  WsCarver w(ws, ws_bytes);
  RenderWs rw;
  rw.st = w.take<RenderState>(IA_RENDER_MAX_ITERS);
  rw.near_w = w.take<float>(R); rw.step = w.take<float>(R); rw.depth = w.take<float>(R); rw.nohit = w.take<float>(R);
  rw.color = w.take<float>((size_t)R * 3); rw.counter = w.take<float>(R);
  rw.alive_a = w.take<int32_t>(R); rw.alive_b = w.take<int32_t>(R); rw.ray_off = w.take<int32_t>(R); rw.ray_cnt = w.take<int32_t>(R);
  rw.s_pts = w.take<float>((size_t)cap * 3); rw.s_t = w.take<float>(cap);
  rw.q = carve_query(w, cap, n_init);
  rw.sample_cap = cap;
  const dim3 blk(256), gR(ia_div_up(R, 256));
  // `resume` = number of iterations enqueued by earlier calls for this frame (0: new frame)
  const int it0 = resume;
  IA_CHECK_ARG(it0 >= 0 && it0 + n_iters < IA_RENDER_MAX_ITERS, "ia_render_test: more than %d wave-front iterations", IA_RENDER_MAX_ITERS - 1);
  if (it0 == 0)
    hipLaunchKernelGGL(k_render_init_rays, gR, blk, 0, s, R, near, far, rw.near_w, rw.step, rw.color, rw.depth,
                       rw.nohit, rw.counter, rw.alive_a, max_samples, rw.st);
  for (int it = it0; it < it0 + n_iters; it++) {
    int32_t *cur = (it & 1) ? rw.alive_b : rw.alive_a;  // alive lists ping-pong on the absolute iteration index
    int32_t *nxt = (it & 1) ? rw.alive_a : rw.alive_b;
    RenderState *st = rw.st + it;
    // upper bounds for the launches: iteration 0 may have R alive rays, later
    // ones never more than the first compaction leaves; keep R (idle waves exit).
#ifndef IA_MARCH_LDS
#define IA_MARCH_LDS 1
#endif
    if (IA_MARCH_LDS && G == 64 && ((size_t)occ_bits & 15) == 0)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_march_compact<true>), gR, blk, 0, s, rays_o, rays_d, rw.near_w, far, rw.step, cur, st, occ_bits,
                         G, aabb, rw.s_pts, rw.s_t, rw.ray_off, rw.ray_cnt, rw.counter, rw.sample_cap, max_samples, max_batch);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_march_compact<false>), gR, blk, 0, s, rays_o, rays_d, rw.near_w, far, rw.step, cur, st, occ_bits,
                         G, aabb, rw.s_pts, rw.s_t, rw.ray_off, rw.ray_cnt, rw.counter, rw.sample_cap, max_samples, max_batch);
    rw.q.n_cand = &st->n_cand;  // zeroed with the record: no zero-fill launch per iteration
    rc = query_impl(rw.s_pts, cap, &st->n_samples, voxel_J, tfs, bone_ids, n_init, grid, F, rw.q, s, 0);
    if (rc) return rc;
    hipLaunchKernelGGL(k_composite_compact, gR, blk, 0, s, cur, nxt, st, rw.ray_off, rw.ray_cnt, rw.s_t, rw.step,
                       rw.q.pt_off, rw.q.pt_cnt, rw.q.cand_rgb, rw.q.cand_sigma, n_init, rw.color, rw.depth, rw.nohit,
                       0.01f);
  }
  hipLaunchKernelGGL(k_render_finalize, gR, blk, 0, s, R, rw.color, rw.depth, rw.nohit, rw.counter, bg,
                     rw.st + it0 + n_iters, max_samples, n_alive_out, rgb, depth, alpha, counter);
  IA_LAUNCH_CHECK("ia_render_test");
  return IA_OK;
}

// per-frame statistics a caller accumulates over a sequence (bench.py: samples per ray, alpha coverage): ONE launch
// instead of a chain of framework reductions behind every frame.  acc[0] += mean(counter), acc[1] += mean(alpha > 0.5).
__global__ __launch_bounds__(256) void k_frame_stats(const float *__restrict__ counter, const float *__restrict__ alpha, int R,
                                                     float *__restrict__ acc) {
  float c = 0.f, a = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < R; i += gridDim.x * blockDim.x) {
    c += counter[i];
    a += alpha[i] > 0.5f ? 1.f : 0.f;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { c += __shfl_xor(c, o, 64); a += __shfl_xor(a, o, 64); }
  // one atomic pair per WORKGROUP: atomics on one address are served one after the other (~11 ns each, measured round 6), 1 024
  // of them (one pair per wave) were 11 of this kernel's 16 us
  __shared__ float s_c[4], s_a[4];
  const int w = threadIdx.x >> 6;
  if (ia_lane() == 0) { s_c[w] = c; s_a[w] = a; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(acc, (s_c[0] + s_c[1] + s_c[2] + s_c[3]) / (float)R);
    atomicAdd(acc + 1, (s_a[0] + s_a[1] + s_a[2] + s_a[3]) / (float)R);
  }
}

extern "C" int ia_frame_stats(const float *counter, const float *alpha, int R, float *acc2, void *stream) {
  IA_CHECK_ARG(counter && alpha && acc2 && R > 0, "ia_frame_stats: bad arguments");
  int blocks = ia_div_up(R, 256 * 8);
  if (blocks > 128) blocks = 128;
  hipLaunchKernelGGL(k_frame_stats, dim3(blocks), dim3(256), 0, (hipStream_t)stream, counter, alpha, R, acc2);
  IA_LAUNCH_CHECK("k_frame_stats");
  return IA_OK;
}

// 8-bit RGBA frame of a rendered image, as animate.py:107-113 stores it: img = cat(rgb, alpha) ; (img * 255).astype(uint8)
// -- the product in fp32, truncated towards zero; values outside [0, 1] (rgb = colour + T * bg can exceed 1 by an ulp) are
// clamped first (numpy's out-of-range float -> uint8 cast is undefined).  One launch, one 32-bit store per ray: the frame leaves the
// device as 4 bytes per ray instead of 16, and the caller's stream carries one kernel instead of cat / clamp / mul / cast.
__global__ __launch_bounds__(256) void k_pack_rgba8(const float *__restrict__ rgb, const float *__restrict__ alpha, int R,
                                                    uint32_t *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R) return;
  auto q = [](float v) -> uint32_t { return (uint32_t)(fminf(fmaxf(v, 0.f), 1.f) * 255.f); };
  out[i] = q(rgb[3 * i]) | (q(rgb[3 * i + 1]) << 8) | (q(rgb[3 * i + 2]) << 16) | (q(alpha[i]) << 24);
}

extern "C" int ia_pack_rgba8(const float *rgb, const float *alpha, int R, uint8_t *rgba, void *stream) {
  IA_CHECK_ARG(R >= 0, "ia_pack_rgba8: R < 0");
  if (R == 0) return IA_OK;
  IA_CHECK_ARG(rgb && alpha && rgba, "ia_pack_rgba8: null pointer");
  IA_CHECK_ARG((reinterpret_cast<uintptr_t>(rgba) & 3) == 0, "ia_pack_rgba8: rgba must be 4-byte aligned");
  hipLaunchKernelGGL(k_pack_rgba8, dim3(ia_div_up(R, 256)), dim3(256), 0, (hipStream_t)stream, rgb, alpha, R,
                     reinterpret_cast<uint32_t *>(rgba));
  IA_LAUNCH_CHECK("k_pack_rgba8");
  return IA_OK;
}

extern "C" int ia_transform_rays_w2s(const float *rays_o, const float *rays_d, const float *w2s, int R, float *o_out,
                                     float *d_out, float *near, float *far, void *stream) {
  IA_CHECK_ARG(R >= 0, "ia_transform_rays_w2s: R < 0");
  if (R == 0) return IA_OK;
  IA_CHECK_ARG(rays_o && rays_d && w2s && o_out && d_out && near && far, "ia_transform_rays_w2s: null pointer");
  hipLaunchKernelGGL(k_transform_rays, dim3(ia_div_up(R, 256)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d,
                     w2s, R, o_out, d_out, near, far);
  IA_LAUNCH_CHECK("k_transform_rays");
  return IA_OK;
}

// ===========================================================================
// Training-side renderer (rows a15 / a8 of SURVEY.md section 8)
// ===========================================================================
// raymarch_train + jitter + sample compaction (raymarcher_acc.py:153-159):
// every ray marches all `max_samples` slots; occupied slots become compact samples
// (contiguous per ray, in slot order) with z = t + jitter * dt and p = z * d + o.
// One WAVE per ray (a training batch has only a few thousand rays: a thread per ray would occupy
// 16 of 256 CUs and walk 2 x 256 dependent steps).  Lane l owns the IA_MT_SPL consecutive steps
// [l*SPL, (l+1)*SPL); it first replays the reference's running sum t += dt up to its first step (so
// every sample depth has exactly the bits of the sequential loop), tests its steps against the
// occupancy bits, and the slot / sample indices come from two wave prefix sums.
#define IA_MT_RAYS 8  // rays (waves) per workgroup: one global atomic per workgroup
__global__ __launch_bounds__(64 * IA_MT_RAYS) void k_march_train_compact(
    const float *__restrict__ rays_o, const float *__restrict__ rays_d, const float *__restrict__ nears,
    const float *__restrict__ fars, int n_rays, const uint32_t *__restrict__ bits, OccDev occ, int max_samples,
    const float *__restrict__ jitter, float *__restrict__ s_pts, float *__restrict__ s_z,
    int32_t *__restrict__ s_slot, int32_t *__restrict__ ray_off, int32_t *__restrict__ ray_cnt,
    int32_t *__restrict__ n_samples, int sample_cap) {
  constexpr int SPL_MAX = 8;
  __shared__ int s_tot[IA_MT_RAYS];
  __shared__ int s_base[IA_MT_RAYS];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = blockIdx.x * IA_MT_RAYS + wave;
  const bool live = n < n_rays;
  const int spl = (max_samples + 2 + 63) / 64;  // the loop runs max_samples (+-1) steps
  MarchRay r;
  float tk[SPL_MAX];
  bool occf[SPL_MAX];
  int c = 0;
  float dt = 0.f;
  if (live) {
    const float t0 = nears[n];
    dt = (fars[n] - t0) / max_samples;  // step_size (raymarcher_acc.py:147)
    r.ox = rays_o[(size_t)n * 3]; r.oy = rays_o[(size_t)n * 3 + 1]; r.oz = rays_o[(size_t)n * 3 + 2];
    r.dx = rays_d[(size_t)n * 3]; r.dy = rays_d[(size_t)n * 3 + 1]; r.dz = rays_d[(size_t)n * 3 + 2];
    r.cx = occ.mn[0]; r.cy = occ.mn[1]; r.cz = occ.mn[2];
    r.sx = occ.G / (occ.mx[0] - occ.mn[0]); r.sy = occ.G / (occ.mx[1] - occ.mn[1]); r.sz = occ.G / (occ.mx[2] - occ.mn[2]);
    r.far = fars[n]; r.dt = dt;
    float t = t0;
    for (int k = 0; k < lane * spl; k++) t += dt;  // the sequential sum of the reference loop
#pragma unroll
    for (int q = 0; q < SPL_MAX; q++) {
      tk[q] = t;
      float x, y, z;
      occf[q] = q < spl && t < r.far && march_occupied(r, bits, occ.G, t, x, y, z);
      c += occf[q] ? 1 : 0;
      t += dt;
    }
  } else {
#pragma unroll
    for (int q = 0; q < SPL_MAX; q++) { tk[q] = 0.f; occf[q] = false; }
  }
  // slots: every occupied step consumes one, the loop stops after max_samples of them
  int tot_c;
  const int excl_c = ia_wave_excl_scan(c, tot_c);
  int kcnt = 0;
#pragma unroll
  for (int q = 0, sl = excl_c; q < SPL_MAX; q++) {
    if (occf[q]) {
      if (sl >= max_samples) occf[q] = false;  // never created by the reference
      sl++;
    }
    kcnt += (occf[q] && tk[q] > 0.f) ? 1 : 0;  // slots with depth <= 0 are masked out (z_vals > 0)
  }
  int tot_k;
  const int excl_k = ia_wave_excl_scan(kcnt, tot_k);
  // reserve exactly the kept samples: slots whose depth is <= 0 (camera within one unit of the root) are
  // masked out by the reference (z_vals > 0) and never written here, so they must not be queued either
  if (lane == 0) s_tot[wave] = live ? tot_k : 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int w = 0; w < IA_MT_RAYS; w++) { s_base[w] = tot; tot += s_tot[w]; }
    const int b = tot > 0 ? atomicAdd(n_samples, tot) : 0;
    for (int w = 0; w < IA_MT_RAYS; w++) s_base[w] += b;
  }
  __syncthreads();
  if (!live) return;
  const int base = s_base[wave];
  if (lane == 0) { ray_off[n] = base; ray_cnt[n] = tot_k; }
  int sl = excl_c, kept = excl_k;
#pragma unroll
  for (int q = 0; q < SPL_MAX; q++) {
    if (occf[q]) {
      const float t = tk[q];
      if (t > 0.f) {
        const int o = base + kept;
        if (o < sample_cap) {
          const float zj = t + (jitter ? jitter[(size_t)n * max_samples + sl] : 0.5f) * dt;
          s_z[o] = zj;
          s_slot[o] = sl;  // position in the reference's dense [n_rays, max_samples] layout
          s_pts[(size_t)o * 3] = zj * r.dx + r.ox; s_pts[(size_t)o * 3 + 1] = zj * r.dy + r.oy; s_pts[(size_t)o * 3 + 2] = zj * r.dz + r.oz;
        }
        kept++;
      }
      sl++;
    }
  }
}

// candidate max for training: invalid slots carry -1e5 (snarf_deformer.py:147), no
// nan_to_num; returns sigma and the index of the winning candidate (-1: an invalid slot won)
__device__ __forceinline__ void cand_max_train(const float *__restrict__ cand_sigma, int off, int cnt, int n_init,
                                               float &sg, int &arg) {
  float best = cnt < n_init ? -1e5f : -INFINITY;
  arg = -1;
  for (int c = 0; c < cnt; c++) {
    const float s = cand_sigma[off + c];
    if (s > best || (arg < 0 && cnt >= n_init && c == 0)) { best = s; arg = off + c; }
  }
  sg = best;
}

// composite() of raymarcher_acc.py:25-36 + render_train tail (:161-186) over compact samples.
// One WAVE per ray: lane l owns a run of consecutive samples; the transmittance
// T_k = prod_{j<k} (1 - alpha_j + 1e-10) is a wave prefix product of the per-lane products (torch's
// cumprod is a parallel scan as well: the association order is not part of the reference).
#define IA_CT_RAYS 4
__global__ __launch_bounds__(64 * IA_CT_RAYS) void k_composite_train_fwd(
    const float *__restrict__ cand_rgb, const float *__restrict__ cand_sigma, int cand_cap,
    const int32_t *__restrict__ pt_off, const uint8_t *__restrict__ pt_cnt, int n_init,
    const int32_t *__restrict__ ray_off, const int32_t *__restrict__ ray_cnt, const float *__restrict__ s_z,
    const float *__restrict__ nears, const float *__restrict__ fars, int n_rays, int max_samples,
    const float *__restrict__ noise, float noise_scale, const float *__restrict__ bg, float *__restrict__ color,
    float *__restrict__ depth, float *__restrict__ alpha_out, float *__restrict__ weights_dense,
    const int32_t *__restrict__ s_slot, int32_t *__restrict__ s_arg, float *__restrict__ s_sigma,
    float *__restrict__ s_alpha, float *__restrict__ s_T) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * IA_CT_RAYS + (threadIdx.x >> 6);
  if (n >= n_rays) return;  // uniform per wave
  const int off = ray_off[n], cnt = ray_cnt[n];
  const float dt = (fars[n] - nears[n]) / max_samples;
  const int per = (cnt + 63) >> 6, k0 = min(lane * per, cnt), k1 = min(k0 + per, cnt);
  // pass 1: sigma / alpha of the lane's samples and the product of their (1 - alpha + 1e-10)
  float prod = 1.f;
  for (int k = k0; k < k1; k++) {
    const int s = off + k;
    float sg; int arg;
    // (candidates past the capacity of the candidate arrays were dropped by the search: never read)
    const int po = pt_off[s], pc = max(0, min((int)pt_cnt[s], cand_cap - po));
    cand_max_train(cand_sigma, po, pc, n_init, sg, arg);
    // raymarcher_acc.py:166-167: randn_like(sigma_vals), a [n_rays, MAX_SAMPLES] tensor -> the draw of (ray, slot); the
    // compact index s depends on the order in which the march kernel's waves allocated their runs
    if (noise) sg += noise_scale * noise[(size_t)n * max_samples + s_slot[s]];
    const float tau = fmaxf(sg, 0.f) * dt;                 // relu(sigma) * dists
    const float a = 1.0f - expf(-tau);
    s_arg[s] = arg; s_sigma[s] = sg; s_alpha[s] = a;
    prod *= (1.0f - a + 1e-10f);
  }
  float incl = prod;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float y = __shfl_up(incl, o, 64);
    if (lane >= o) incl = y * incl;
  }
  float T = __shfl_up(incl, 1, 64);
  if (lane == 0) T = 1.f;
  const float T_end = __shfl(incl, 63, 64);
  // pass 2: weights and the per-ray sums
  float c0 = 0.f, c1 = 0.f, c2 = 0.f, dep = 0.f, asum = 0.f;  // weights_dense is zero-filled by the caller
  for (int k = k0; k < k1; k++) {
    const int s = off + k;
    const float a = s_alpha[s];
    const int arg = s_arg[s];
    const float w = a * T;
    s_T[s] = T;
    if (arg >= 0) { c0 += w * cand_rgb[(size_t)arg * 3]; c1 += w * cand_rgb[(size_t)arg * 3 + 1]; c2 += w * cand_rgb[(size_t)arg * 3 + 2]; }
    dep += w * s_z[s];
    asum += w;
    weights_dense[(size_t)n * max_samples + s_slot[s]] = w;
    T = T * (1.0f - a + 1e-10f);                           // cumprod(1 - alpha + 1e-10)
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    c0 += __shfl_xor(c0, o, 64); c1 += __shfl_xor(c1, o, 64); c2 += __shfl_xor(c2, o, 64);
    dep += __shfl_xor(dep, o, 64); asum += __shfl_xor(asum, o, 64);
  }
  if (lane == 0) {
    const float b0 = bg ? bg[(size_t)n * 3] : 1.f, b1 = bg ? bg[(size_t)n * 3 + 1] : 1.f, b2 = bg ? bg[(size_t)n * 3 + 2] : 1.f;
    color[(size_t)n * 3] = c0 + T_end * b0; color[(size_t)n * 3 + 1] = c1 + T_end * b1; color[(size_t)n * 3 + 2] = c2 + T_end * b2;
    depth[n] = dep;
    alpha_out[n] = asum;   // alpha_coarse = weights.sum(-1) (:184)
  }
}

// Backward: gT_k = gw_k a_k + gT_{k+1} (1 - a_k + 1e-10) is an affine recurrence running from the
// last sample to the first; each lane composes the affine map of its run, a reverse wave scan of
// (slope, offset) pairs yields the value entering every run, then the lane walks its run.
__global__ __launch_bounds__(64 * IA_CT_RAYS) void k_composite_train_bwd(
    const float *__restrict__ d_color, const float *__restrict__ d_depth, const float *__restrict__ d_alpha,
    const float *__restrict__ d_weights, const float *__restrict__ cand_rgb, const int32_t *__restrict__ ray_off,
    const int32_t *__restrict__ ray_cnt, const float *__restrict__ s_z, const float *__restrict__ nears,
    const float *__restrict__ fars, int n_rays, int max_samples, const float *__restrict__ bg,
    const int32_t *__restrict__ s_slot, const int32_t *__restrict__ s_arg, const float *__restrict__ s_sigma,
    const float *__restrict__ s_alpha, const float *__restrict__ s_T, float *__restrict__ d_cand_rgb,
    float *__restrict__ d_cand_sigma) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * IA_CT_RAYS + (threadIdx.x >> 6);
  if (n >= n_rays) return;  // uniform per wave
  const int off = ray_off[n], cnt = ray_cnt[n];
  if (cnt == 0) return;
  const float dt = (fars[n] - nears[n]) / max_samples;
  const float dc0 = d_color ? d_color[(size_t)n * 3] : 0.f, dc1 = d_color ? d_color[(size_t)n * 3 + 1] : 0.f,
              dc2 = d_color ? d_color[(size_t)n * 3 + 2] : 0.f;
  const float dd = d_depth ? d_depth[n] : 0.f, da = d_alpha ? d_alpha[n] : 0.f;
  const float b0 = bg ? bg[(size_t)n * 3] : 1.f, b1 = bg ? bg[(size_t)n * 3 + 1] : 1.f, b2 = bg ? bg[(size_t)n * 3 + 2] : 1.f;
  const float gT_end = dc0 * b0 + dc1 * b1 + dc2 * b2;  // dL/dT_end
  const int per = (cnt + 63) >> 6, k0 = min(lane * per, cnt), k1 = min(k0 + per, cnt);
  auto gw_of = [&](int s, int arg) {
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
    if (arg >= 0) { r0 = cand_rgb[(size_t)arg * 3]; r1 = cand_rgb[(size_t)arg * 3 + 1]; r2 = cand_rgb[(size_t)arg * 3 + 2]; }
    return dc0 * r0 + dc1 * r1 + dc2 * r2 + dd * s_z[s] + da + (d_weights ? d_weights[(size_t)n * max_samples + s_slot[s]] : 0.f);
  };
  // affine map of the lane's run: gT_out = B + M * gT_in (gT_in enters from the later samples)
  float M = 1.f, B = 0.f;
  for (int k = k1 - 1; k >= k0; k--) {
    const int s = off + k;
    const float a = s_alpha[s], m = 1.0f - a + 1e-10f, b = gw_of(s, s_arg[s]) * a;
    B = b + m * B;
    M = m * M;
  }
  // suffix composition S_l = F_l o F_{l+1} o ... o F_63
  float SM = M, SB = B;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float m2 = __shfl_down(SM, o, 64), b2s = __shfl_down(SB, o, 64);
    if (lane + o < 64) { SB = SB + SM * b2s; SM = SM * m2; }
  }
  float inM = __shfl_down(SM, 1, 64), inB = __shfl_down(SB, 1, 64);
  if (lane == 63) { inM = 1.f; inB = 0.f; }
  float gT = inB + inM * gT_end;
  for (int k = k1 - 1; k >= k0; k--) {
    const int s = off + k;
    const float a = s_alpha[s], T = s_T[s];
    const int arg = s_arg[s];
    const float gw = gw_of(s, arg);
    const float w = a * T;
    const float g_alpha = gw * T - gT * T;
    gT = gw * a + gT * (1.0f - a + 1e-10f);
    if (arg >= 0) {
      d_cand_rgb[(size_t)arg * 3] = w * dc0; d_cand_rgb[(size_t)arg * 3 + 1] = w * dc1; d_cand_rgb[(size_t)arg * 3 + 2] = w * dc2;
      // d alpha / d sigma = exp(-tau) * dt for sigma > 0 (relu), exp(-tau) = 1 - alpha
      d_cand_sigma[arg] = (s_sigma[s] > 0.f) ? g_alpha * (1.0f - a) * dt : 0.f;
    }
  }
}

extern "C" int ia_march_train_compact(const float *rays_o, const float *rays_d, const float *nears, const float *fars,
                                      int n_rays, const uint32_t *occ_bits, const ia_occ_grid *occ, int max_samples,
                                      const float *jitter, float *s_pts, float *s_z, int32_t *s_slot,
                                      int32_t *ray_off, int32_t *ray_cnt, int32_t *n_samples, int sample_cap,
                                      void *stream) {
  IA_CHECK_ARG(n_rays >= 0 && max_samples > 0 && max_samples + 2 <= 64 * 8, "ia_march_train_compact: bad sizes (max_samples <= 510)");
  IA_CHECK_ARG(n_samples, "ia_march_train_compact: n_samples is null");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_fill_i32, dim3(1), dim3(64), 0, s, n_samples, 0, 1);
  if (n_rays == 0) return IA_OK;
  IA_CHECK_ARG(rays_o && rays_d && nears && fars && occ_bits && occ && s_pts && s_z && s_slot && ray_off && ray_cnt,
               "ia_march_train_compact: null pointer");
  hipLaunchKernelGGL(k_march_train_compact, dim3(ia_div_up(n_rays, IA_MT_RAYS)), dim3(64 * IA_MT_RAYS), 0, s, rays_o, rays_d, nears, fars,
                     n_rays, occ_bits, make_occ(occ), max_samples, jitter, s_pts, s_z, s_slot, ray_off, ray_cnt,
                     n_samples, sample_cap);
  IA_LAUNCH_CHECK("k_march_train_compact");
  return IA_OK;
}

extern "C" int ia_composite_train_fwd(const float *cand_rgb, const float *cand_sigma, int cand_cap,
                                      const int32_t *pt_off, const uint8_t *pt_cnt, int n_init, const int32_t *ray_off, const int32_t *ray_cnt,
                                      const float *s_z, const float *nears, const float *fars, int n_rays,
                                      int max_samples, const float *noise, float noise_scale, const float *bg,
                                      float *color, float *depth, float *alpha, float *weights_dense,
                                      const int32_t *s_slot, int32_t *s_arg, float *s_sigma, float *s_alpha,
                                      float *s_T, void *stream) {
  IA_CHECK_ARG(n_rays >= 0 && max_samples > 0, "ia_composite_train_fwd: bad sizes");
  if (n_rays == 0) return IA_OK;
  IA_CHECK_ARG(pt_off && pt_cnt && ray_off && ray_cnt && s_z && nears && fars && color && depth && alpha &&
               weights_dense && s_slot && s_arg && s_sigma && s_alpha && s_T, "ia_composite_train_fwd: null pointer");
  hipLaunchKernelGGL(k_composite_train_fwd, dim3(ia_div_up(n_rays, IA_CT_RAYS)), dim3(64 * IA_CT_RAYS), 0, (hipStream_t)stream, cand_rgb,
                     cand_sigma, cand_cap, pt_off, pt_cnt, n_init, ray_off, ray_cnt, s_z, nears, fars, n_rays, max_samples, noise,
                     noise_scale, bg, color, depth, alpha, weights_dense, s_slot, s_arg, s_sigma, s_alpha, s_T);
  IA_LAUNCH_CHECK("k_composite_train_fwd");
  return IA_OK;
}

extern "C" int ia_composite_train_bwd(const float *d_color, const float *d_depth, const float *d_alpha,
                                      const float *d_weights, const float *cand_rgb, const int32_t *ray_off,
                                      const int32_t *ray_cnt, const float *s_z, const float *nears, const float *fars,
                                      int n_rays, int max_samples, const float *bg, const int32_t *s_slot,
                                      const int32_t *s_arg, const float *s_sigma, const float *s_alpha, const float *s_T,
                                      float *d_cand_rgb, float *d_cand_sigma, void *stream) {
  IA_CHECK_ARG(n_rays >= 0 && max_samples > 0, "ia_composite_train_bwd: bad sizes");
  if (n_rays == 0) return IA_OK;
  IA_CHECK_ARG(ray_off && ray_cnt && s_z && nears && fars && s_slot && s_arg && s_sigma && s_alpha && s_T && d_cand_rgb &&
               d_cand_sigma, "ia_composite_train_bwd: null pointer");
  hipLaunchKernelGGL(k_composite_train_bwd, dim3(ia_div_up(n_rays, IA_CT_RAYS)), dim3(64 * IA_CT_RAYS), 0, (hipStream_t)stream, d_color,
                     d_depth, d_alpha, d_weights, cand_rgb, ray_off, ray_cnt, s_z, nears, fars, n_rays, max_samples, bg,
                     s_slot, s_arg, s_sigma, s_alpha, s_T, d_cand_rgb, d_cand_sigma);
  IA_LAUNCH_CHECK("k_composite_train_bwd");
  return IA_OK;
}

// arg-max candidate per point for the training path (fill = -1e5 wins -> -1); candidates past
// cand_cap (the length of cand_sigma) were dropped by the compaction and are not read
__global__ __launch_bounds__(256) void k_candidate_argmax(const float *__restrict__ cand_sigma, int cand_cap,
                                                          const int32_t *__restrict__ pt_off,
                                                          const uint8_t *__restrict__ pt_cnt, int P, int n_init,
                                                          int32_t *__restrict__ arg) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float sg; int a;
  const int po = pt_off[p], pc = max(0, min((int)pt_cnt[p], cand_cap - po));
  cand_max_train(cand_sigma, po, pc, n_init, sg, a);
  arg[p] = a;
}

extern "C" int ia_candidate_argmax(const float *cand_sigma, int cand_cap, const int32_t *pt_off, const uint8_t *pt_cnt,
                                   int P, int n_init, int32_t *arg, void *stream) {
  IA_CHECK_ARG(P >= 0, "ia_candidate_argmax: P < 0");
  if (P == 0) return IA_OK;
  IA_CHECK_ARG(pt_off && pt_cnt && arg, "ia_candidate_argmax: null pointer");
  hipLaunchKernelGGL(k_candidate_argmax, dim3(ia_div_up(P, 256)), dim3(256), 0, (hipStream_t)stream, cand_sigma, cand_cap,
                     pt_off, pt_cnt, P, n_init, arg);
  IA_LAUNCH_CHECK("k_candidate_argmax");
  return IA_OK;
}

// ---------------------------------------------------------------------------
// deform_train's arg-max gather (snarf_deformer.py:150-158) as a kernel pair: forward takes the
// winning candidate's (rgb, sigma) per point (fill values where an invalid slot won), backward
// scatters the point gradients back.  Every candidate belongs to exactly one point, so the scatter
// is unique: plain stores, no atomics, no sort (torch's index backward sorts the indices first).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_candidate_gather_fwd(const float *__restrict__ cand_rgb,
                                                              const float *__restrict__ cand_sigma,
                                                              const int32_t *__restrict__ arg, int P, float fill,
                                                              float *__restrict__ rgb, float *__restrict__ sigma) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int a = arg[p];
  sigma[p] = a >= 0 ? cand_sigma[a] : fill;
#pragma unroll
  for (int c = 0; c < 3; c++) rgb[(size_t)p * 3 + c] = a >= 0 ? cand_rgb[(size_t)a * 3 + c] : 0.f;
}

__global__ __launch_bounds__(256) void k_candidate_gather_bwd(const float *__restrict__ d_rgb,
                                                              const float *__restrict__ d_sigma,
                                                              const int32_t *__restrict__ arg, int P,
                                                              float *__restrict__ d_cand_rgb,
                                                              float *__restrict__ d_cand_sigma) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int a = arg[p];
  if (a < 0) return;
  if (d_sigma) d_cand_sigma[a] = d_sigma[p];
  if (d_rgb) {
#pragma unroll
    for (int c = 0; c < 3; c++) d_cand_rgb[(size_t)a * 3 + c] = d_rgb[(size_t)p * 3 + c];
  }
}

extern "C" int ia_candidate_gather_fwd(const float *cand_rgb, const float *cand_sigma, const int32_t *arg, int P,
                                       float fill, float *rgb, float *sigma, void *stream) {
  IA_CHECK_ARG(P >= 0, "ia_candidate_gather_fwd: P < 0");
  if (P == 0) return IA_OK;
  IA_CHECK_ARG(cand_rgb && cand_sigma && arg && rgb && sigma, "ia_candidate_gather_fwd: null pointer");
  hipLaunchKernelGGL(k_candidate_gather_fwd, dim3(ia_div_up(P, 256)), dim3(256), 0, (hipStream_t)stream, cand_rgb,
                     cand_sigma, arg, P, fill, rgb, sigma);
  IA_LAUNCH_CHECK("k_candidate_gather_fwd");
  return IA_OK;
}

extern "C" int ia_candidate_gather_bwd(const float *d_rgb, const float *d_sigma, const int32_t *arg, int P,
                                       float *d_cand_rgb, float *d_cand_sigma, void *stream) {
  IA_CHECK_ARG(P >= 0, "ia_candidate_gather_bwd: P < 0");
  if (P == 0) return IA_OK;
  IA_CHECK_ARG(arg && d_cand_rgb && d_cand_sigma, "ia_candidate_gather_bwd: null pointer");
  hipLaunchKernelGGL(k_candidate_gather_bwd, dim3(ia_div_up(P, 256)), dim3(256), 0, (hipStream_t)stream, d_rgb, d_sigma,
                     arg, P, d_cand_rgb, d_cand_sigma);
  IA_LAUNCH_CHECK("k_candidate_gather_bwd");
  return IA_OK;
}
