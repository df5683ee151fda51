// ia_smpl_nn.hip -- SMPLDeformer (instant_avatar/deformers/smpl_deformer.py:86-131): the deformer
// plugin that maps a point to canonical space with the inverse transform of its NEAREST SMPL vertex
// (pytorch3d knn_points, K = 1) when that vertex is closer than `threshold`.
//
// Exact 1-NN by brute force: 6 890 vertices are 83 KB -- the whole vertex set is staged in LDS
// once per workgroup (all lanes of a wave read the SAME vertex: an LDS broadcast, no bank
// conflicts), every thread keeps the running minimum of its own point.  Squared distance in the
// checker's operation order (dx*dx, fma dy, fma dz), first minimum wins.  Optional compaction of
// the valid points (ballot + block scan + one global atomic) feeds the field kernels directly, so
// deform_test / deform_train run without the reference's boolean-mask gathers and `.any()` syncs.
#include "ia_common.h"

#define IA_NN_THREADS 256

__global__ __launch_bounds__(IA_NN_THREADS) void k_smpl_nn(
    const float *__restrict__ pts, int P, const int32_t *__restrict__ n_pts_dev, const float *__restrict__ verts,
    const float *__restrict__ T_inv, int NV, float thr2, float *__restrict__ pts_cano, uint8_t *__restrict__ valid,
    int32_t *__restrict__ idx_out,
    // compaction (optional): canonical positions of the valid points, per-point offset / count (0|1)
    float *__restrict__ cand_xc, int32_t *__restrict__ pt_off, uint8_t *__restrict__ pt_cnt,
    int32_t *__restrict__ n_cand, int32_t *__restrict__ cand_pt) {
  extern __shared__ __attribute__((aligned(16))) float s_v[];  // [NV][3]
  __shared__ int s_wtot[IA_NN_THREADS / 64];
  __shared__ int s_base;
  if (n_pts_dev) P = min(P, *n_pts_dev);
  if ((int)(blockIdx.x * IA_NN_THREADS) >= P) return;  // uniform per workgroup
  for (int e = threadIdx.x; e < NV * 3; e += IA_NN_THREADS) s_v[e] = verts[e];
  __syncthreads();
  const int i = blockIdx.x * IA_NN_THREADS + threadIdx.x;
  const bool live = i < P;
  float px = 0.f, py = 0.f, pz = 0.f;
  if (live) { px = pts[(size_t)i * 3]; py = pts[(size_t)i * 3 + 1]; pz = pts[(size_t)i * 3 + 2]; }
  float best = INFINITY;
  int bi = 0;
#pragma unroll 4
  for (int v = 0; v < NV; v++) {
    const float dx = px - s_v[v * 3], dy = py - s_v[v * 3 + 1], dz = pz - s_v[v * 3 + 2];
    const float dist = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    if (dist < best) { best = dist; bi = v; }
  }
  const float *T = T_inv + (size_t)bi * 16;
  float c[3];
#pragma unroll
  for (int r = 0; r < 3; r++) c[r] = IA_DOT3(T[r * 4], px, T[r * 4 + 1], py, T[r * 4 + 2], pz) + T[r * 4 + 3];
  const bool ok = live && best < thr2;
  if (live) {
    if (pts_cano) { pts_cano[(size_t)i * 3] = c[0]; pts_cano[(size_t)i * 3 + 1] = c[1]; pts_cano[(size_t)i * 3 + 2] = c[2]; }
    if (valid) valid[i] = ok;
    if (idx_out) idx_out[i] = bi;
  }
  if (!cand_xc) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long m = __ballot(ok);
  if (lane == 0) s_wtot[wave] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int w = 0; w < IA_NN_THREADS / 64; w++) { const int t = s_wtot[w]; s_wtot[w] = tot; tot += t; }
    s_base = tot > 0 ? atomicAdd(n_cand, tot) : 0;
  }
  __syncthreads();
  if (!live) return;
  const int o = s_base + s_wtot[wave] + __popcll(m & ((1ull << lane) - 1ull));
  pt_off[i] = o;
  pt_cnt[i] = ok ? 1 : 0;
  if (ok) {
    cand_xc[(size_t)o * 3] = c[0]; cand_xc[(size_t)o * 3 + 1] = c[1]; cand_xc[(size_t)o * 3 + 2] = c[2];
    if (cand_pt) cand_pt[o] = i;     // the point a compact candidate belongs to (the training route's backward)
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Vertex grid for the FUSED queries (round 6).  The brute-force kernel above tests every point against all 6 890 vertices
// (0.9 ms per ~2 x 10^5 samples: the largest kernel of a fit step, ~6 ms for the 1.3 M occupancy probes of a frame).  The fused
// queries (`ia_smpl_deform_query`, `ia_smpl_nn_compact`) only ever use the nearest vertex of a point when it is closer than
// `threshold` (smpl_deformer.py:102-104: everything else is invalid and takes the fill values), and every vertex closer than
// `threshold` to a point lies in the 3 x 3 x 3 cells around the point's cell when the cells are at least `threshold` wide.
// So: bin the posed vertices once per frame (counting sort on the device, no host read: bounding box, dimensions and cell size
// are computed in the kernel), and test a point against the ~50 vertices of its 27 cells.  The winner -- smallest distance,
// lowest vertex index among equals, the distance expression of k_smpl_nn -- is the brute-force winner whenever the point is
// valid; for invalid points nothing is output.  `ia_smpl_nn_deform` (all points, exact index) stays brute force.
// ---------------------------------------------------------------------------------------------------------------------
#define IA_NNG_MAX_DIM 64
#define IA_NNG_MAX_CELLS (IA_NNG_MAX_DIM * IA_NNG_MAX_DIM * IA_NNG_MAX_DIM)
struct NnGridHeader {
  float origin[3], inv_h, h;
  int nx, ny, nz, n_cells, n_verts;
  int pad[6];
};
struct NnGrid {            // views into the caller's buffer
  NnGridHeader *hdr;
  int32_t *cell_start;     // [IA_NNG_MAX_CELLS + 1]  exclusive prefix of the per-cell counts
  int32_t *cursor;         // [IA_NNG_MAX_CELLS]      counts, then fill cursors
  int32_t *vidx;           // [V]                     vertex index of every slot
  float *vpos;             // [V, 3]                  its position (gathered: the query reads slots, not vertices)
};
extern "C" size_t ia_smpl_nn_grid_bytes(int n_verts) {
  const size_t v = (size_t)(n_verts > 0 ? n_verts : 1);
  return ia_align(sizeof(NnGridHeader)) + ia_align((size_t)(IA_NNG_MAX_CELLS + 1) * 4) + ia_align((size_t)IA_NNG_MAX_CELLS * 4) + ia_align(v * 4) + ia_align(v * 12);
}
static inline NnGrid nn_grid_carve(void *buf, int n_verts) {
  char *p = (char *)buf;
  NnGrid g;
  g.hdr = (NnGridHeader *)p; p += ia_align(sizeof(NnGridHeader));
  g.cell_start = (int32_t *)p; p += ia_align((size_t)(IA_NNG_MAX_CELLS + 1) * 4);
  g.cursor = (int32_t *)p; p += ia_align((size_t)IA_NNG_MAX_CELLS * 4);
  g.vidx = (int32_t *)p; p += ia_align((size_t)n_verts * 4);
  g.vpos = (float *)p;
  return g;
}

__device__ __forceinline__ int nn_cell_of(const NnGridHeader &h, float x, float y, float z, int &cx, int &cy, int &cz) {
  cx = (int)floorf((x - h.origin[0]) * h.inv_h); cy = (int)floorf((y - h.origin[1]) * h.inv_h); cz = (int)floorf((z - h.origin[2]) * h.inv_h);
  return (cz * h.ny + cy) * h.nx + cx;
}

// one workgroup: bounding box of the vertices -> header (cell size >= min_cell, at most 62 interior cells per axis + a border
// cell on either side), counts zeroed
__global__ __launch_bounds__(1024) void k_nn_grid_header(const float *__restrict__ verts, int V, float min_cell, NnGrid g) {
  __shared__ float s_mn[16][3], s_mx[16][3];
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int v = threadIdx.x; v < V; v += 1024)
#pragma unroll
    for (int c = 0; c < 3; c++) { const float t = verts[(size_t)v * 3 + c]; mn[c] = fminf(mn[c], t); mx[c] = fmaxf(mx[c], t); }
#pragma unroll
  for (int c = 0; c < 3; c++) { mn[c] = ia_wave_min(mn[c]); mx[c] = ia_wave_max(mx[c]); }
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int c = 0; c < 3; c++) { s_mn[threadIdx.x >> 6][c] = mn[c]; s_mx[threadIdx.x >> 6][c] = mx[c]; }
  __syncthreads();
  __shared__ NnGridHeader s_h;
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; w++)
      for (int c = 0; c < 3; c++) { s_mn[0][c] = fminf(s_mn[0][c], s_mn[w][c]); s_mx[0][c] = fmaxf(s_mx[0][c], s_mx[w][c]); }
    float ext = 0.f;
    for (int c = 0; c < 3; c++) ext = fmaxf(ext, s_mx[0][c] - s_mn[0][c]);
    float h = fmaxf(min_cell, ext / (float)(IA_NNG_MAX_DIM - 2) * 1.0001f);
    if (!(h > 0.f)) h = 1.f;                     // (degenerate input: one cell)
    NnGridHeader H;
    H.h = h; H.inv_h = 1.0f / h; H.n_verts = V;
    int dims[3];
    for (int c = 0; c < 3; c++) {
      H.origin[c] = s_mn[0][c] - h;              // one border cell below ...
      dims[c] = min(IA_NNG_MAX_DIM, (int)floorf((s_mx[0][c] - H.origin[c]) * H.inv_h) + 2);   // ... and one above the last vertex cell
    }
    H.nx = dims[0]; H.ny = dims[1]; H.nz = dims[2]; H.n_cells = dims[0] * dims[1] * dims[2];
    s_h = H;
    *g.hdr = H;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < s_h.n_cells; c += 1024) g.cursor[c] = 0;
}

__global__ __launch_bounds__(256) void k_nn_grid_count(const float *__restrict__ verts, int V, NnGrid g) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const NnGridHeader h = *g.hdr;
  int cx, cy, cz;
  nn_cell_of(h, verts[(size_t)v * 3], verts[(size_t)v * 3 + 1], verts[(size_t)v * 3 + 2], cx, cy, cz);
  cx = min(max(cx, 0), h.nx - 1); cy = min(max(cy, 0), h.ny - 1); cz = min(max(cz, 0), h.nz - 1);   // (rounding at the upper face)
  atomicAdd(g.cursor + (cz * h.ny + cy) * h.nx + cx, 1);
}

// one workgroup: exclusive prefix of the counts -> cell_start; the counts become the fill cursors
__global__ __launch_bounds__(1024) void k_nn_grid_scan(NnGrid g) {
  __shared__ int s_part[1024];
  const int n = g.hdr->n_cells;
  const int per = (n + 1023) / 1024, c0 = min(threadIdx.x * per, n), c1 = min(c0 + per, n);
  int sum = 0;
  for (int c = c0; c < c1; c++) sum += g.cursor[c];
  s_part[threadIdx.x] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int t = threadIdx.x >= o ? s_part[threadIdx.x - o] : 0;
    __syncthreads();
    s_part[threadIdx.x] += t;
    __syncthreads();
  }
  int run = s_part[threadIdx.x] - sum;
  for (int c = c0; c < c1; c++) { const int k = g.cursor[c]; g.cell_start[c] = run; g.cursor[c] = run; run += k; }
  if (threadIdx.x == 1023) g.cell_start[n] = s_part[1023];
}

__global__ __launch_bounds__(256) void k_nn_grid_fill(const float *__restrict__ verts, int V, NnGrid g) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const NnGridHeader h = *g.hdr;
  const float x = verts[(size_t)v * 3], y = verts[(size_t)v * 3 + 1], z = verts[(size_t)v * 3 + 2];
  int cx, cy, cz;
  nn_cell_of(h, x, y, z, cx, cy, cz);
  cx = min(max(cx, 0), h.nx - 1); cy = min(max(cy, 0), h.ny - 1); cz = min(max(cz, 0), h.nz - 1);
  const int slot = atomicAdd(g.cursor + (cz * h.ny + cy) * h.nx + cx, 1);     // (order inside a cell: arrival -- the query breaks ties by index)
  g.vidx[slot] = v;
  g.vpos[(size_t)slot * 3] = x; g.vpos[(size_t)slot * 3 + 1] = y; g.vpos[(size_t)slot * 3 + 2] = z;
}

extern "C" int ia_smpl_nn_grid_build(const float *verts, int n_verts, float threshold, void *grid, size_t grid_bytes, void *stream) {
  IA_CHECK_ARG(verts && grid && n_verts > 0 && threshold > 0.f, "ia_smpl_nn_grid_build: bad arguments");
  IA_CHECK_ARG(grid_bytes >= ia_smpl_nn_grid_bytes(n_verts), "ia_smpl_nn_grid_build: buffer of %zu bytes, %zu needed", grid_bytes, ia_smpl_nn_grid_bytes(n_verts));
  hipStream_t s = (hipStream_t)stream;
  NnGrid g = nn_grid_carve(grid, n_verts);
  // cells a little WIDER than the threshold: |v - p| < threshold then differs by less than 0.999 cells per axis, so that the rounding
  // of the two cell coordinates (~1e-5 of a cell) can never put a valid vertex two cells away
  hipLaunchKernelGGL(k_nn_grid_header, dim3(1), dim3(1024), 0, s, verts, n_verts, threshold * 1.001f, g);
  hipLaunchKernelGGL(k_nn_grid_count, dim3(ia_div_up(n_verts, 256)), dim3(256), 0, s, verts, n_verts, g);
  hipLaunchKernelGGL(k_nn_grid_scan, dim3(1), dim3(1024), 0, s, g);
  hipLaunchKernelGGL(k_nn_grid_fill, dim3(ia_div_up(n_verts, 256)), dim3(256), 0, s, verts, n_verts, g);
  IA_LAUNCH_CHECK("ia_smpl_nn_grid_build");
  return IA_OK;
}

// the grid-pruned query: the outputs of k_smpl_nn's compaction branch (cand_xc, pt_off, pt_cnt, n_cand, cand_pt) and idx for the
// VALID points (-1 elsewhere)
__global__ __launch_bounds__(IA_NN_THREADS) void k_smpl_nn_grid(
    const float *__restrict__ pts, int P, const int32_t *__restrict__ n_pts_dev, NnGrid g, const float *__restrict__ T_inv, float thr2,
    int32_t *__restrict__ idx_out, float *__restrict__ cand_xc, int32_t *__restrict__ pt_off, uint8_t *__restrict__ pt_cnt,
    int32_t *__restrict__ n_cand, int32_t *__restrict__ cand_pt) {
  __shared__ int s_wtot[IA_NN_THREADS / 64];
  __shared__ int s_base;
  if (n_pts_dev) P = min(P, *n_pts_dev);
  if ((int)(blockIdx.x * IA_NN_THREADS) >= P) return;  // uniform per workgroup
  const NnGridHeader h = *g.hdr;
  const int i = blockIdx.x * IA_NN_THREADS + threadIdx.x;
  const bool live = i < P;
  float px = 0.f, py = 0.f, pz = 0.f;
  if (live) { px = pts[(size_t)i * 3]; py = pts[(size_t)i * 3 + 1]; pz = pts[(size_t)i * 3 + 2]; }
  float best = INFINITY;
  int bi = -1;
  int cx, cy, cz;
  nn_cell_of(h, px, py, pz, cx, cy, cz);
  // (a point further than one cell outside the grid has no vertex within `threshold`; NaN coordinates fail every test)
  if (live && cx >= -1 && cx <= h.nx && cy >= -1 && cy <= h.ny && cz >= -1 && cz <= h.nz) {
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, h.nx - 1);
    for (int dz = -1; dz <= 1; dz++) {
      const int z = cz + dz;
      if (z < 0 || z >= h.nz) continue;
      for (int dy = -1; dy <= 1; dy++) {
        const int y = cy + dy;
        if (y < 0 || y >= h.ny || x0 > x1) continue;
        const int row = (z * h.ny + y) * h.nx;
        const int s0 = g.cell_start[row + x0], s1 = g.cell_start[row + x1 + 1];   // the x-neighbours are consecutive cells: one slot range
        for (int sidx = s0; sidx < s1; sidx++) {
          const float dx = px - g.vpos[(size_t)sidx * 3], dy2 = py - g.vpos[(size_t)sidx * 3 + 1], dz2 = pz - g.vpos[(size_t)sidx * 3 + 2];
          const float dist = __builtin_fmaf(dz2, dz2, __builtin_fmaf(dy2, dy2, dx * dx));     // k_smpl_nn's expression
          const int v = g.vidx[sidx];
          if (dist < best || (dist == best && v < bi)) { best = dist; bi = v; }                // lowest index among equals
        }
      }
    }
  }
  const bool ok = live && bi >= 0 && best < thr2;
  float c[3] = {0.f, 0.f, 0.f};
  if (ok) {
    const float *T = T_inv + (size_t)bi * 16;
#pragma unroll
    for (int r = 0; r < 3; r++) c[r] = IA_DOT3(T[r * 4], px, T[r * 4 + 1], py, T[r * 4 + 2], pz) + T[r * 4 + 3];
  }
  if (live && idx_out) idx_out[i] = ok ? bi : -1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long m = __ballot(ok);
  if (lane == 0) s_wtot[wave] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int w = 0; w < IA_NN_THREADS / 64; w++) { const int t = s_wtot[w]; s_wtot[w] = tot; tot += t; }
    s_base = tot > 0 ? atomicAdd(n_cand, tot) : 0;
  }
  __syncthreads();
  if (!live) return;
  const int o = s_base + s_wtot[wave] + __popcll(m & ((1ull << lane) - 1ull));
  pt_off[i] = o;
  pt_cnt[i] = ok ? 1 : 0;
  if (ok) {
    cand_xc[(size_t)o * 3] = c[0]; cand_xc[(size_t)o * 3 + 1] = c[1]; cand_xc[(size_t)o * 3 + 2] = c[2];
    if (cand_pt) cand_pt[o] = i;
  }
}

// Zero-fill as a kernel of this library, not hipMemsetAsync: recorded into a HIP graph a memset becomes a memset NODE, and a graph
// with memset nodes replayed back-to-back without host synchronisation faulted on ROCm 7.2 (fit stage, round 6: "Memory access fault
// by GPU" at a non-deterministic replay, never with a synchronisation between replays, never in eager mode; NOTES.md).  Every other
// zero-fill of the library already was a kernel.
__global__ void k_zero_words(uint32_t *__restrict__ p, size_t n_words) {
  size_t n4 = n_words / 4;
  uint4 *p4 = reinterpret_cast<uint4 *>(p);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p4[i] = make_uint4(0, 0, 0, 0);
  if (blockIdx.x == 0 && threadIdx.x < (n_words & 3)) p[n4 * 4 + threadIdx.x] = 0;
}

// bytes: multiple of 4; p: 16-byte aligned (torch allocations, workspace slices)
static void zero_fill(void *p, size_t bytes, hipStream_t s) {
  size_t n_words = bytes / 4;
  if (n_words == 0) return;
  int blocks = (int)std::min<size_t>((n_words / 4 + 255) / 256 + 1, 2048);
  hipLaunchKernelGGL(k_zero_words, dim3(blocks), dim3(256), 0, s, (uint32_t *)p, n_words);
}

static int launch_nn_grid(const float *pts, int P, const int32_t *n_pts_dev, const void *grid, int NV, const float *T_inv, float threshold,
                          int32_t *idx, float *cand_xc, int32_t *pt_off, uint8_t *pt_cnt, int32_t *n_cand, int32_t *cand_pt, hipStream_t s) {
  NnGrid g = nn_grid_carve(const_cast<void *>(grid), NV);
  hipLaunchKernelGGL(k_smpl_nn_grid, dim3(ia_div_up(P, IA_NN_THREADS)), dim3(IA_NN_THREADS), 0, s, pts, P, n_pts_dev, g, T_inv,
                     threshold * threshold, idx, cand_xc, pt_off, pt_cnt, n_cand, cand_pt);
  IA_LAUNCH_CHECK("k_smpl_nn_grid");
  return IA_OK;
}

static int launch_nn(const float *pts, int P, const int32_t *n_pts_dev, const float *verts, const float *T_inv, int NV,
                     float threshold, float *pts_cano, uint8_t *valid, int32_t *idx, float *cand_xc, int32_t *pt_off,
                     uint8_t *pt_cnt, int32_t *n_cand, hipStream_t s, int32_t *cand_pt = nullptr) {
  const size_t shmem = (size_t)NV * 12;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_smpl_nn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    attr_done = true;
  }
  hipLaunchKernelGGL(k_smpl_nn, dim3(ia_div_up(P, IA_NN_THREADS)), dim3(IA_NN_THREADS), shmem, s, pts, P, n_pts_dev, verts,
                     T_inv, NV, threshold * threshold, pts_cano, valid, idx, cand_xc, pt_off, pt_cnt, n_cand, cand_pt);
  IA_LAUNCH_CHECK("k_smpl_nn");
  return IA_OK;
}

extern "C" int ia_smpl_nn_deform(const float *pts, int P, const int32_t *n_pts_dev, const float *verts,
                                 const float *T_inv, int n_verts, float threshold, float *pts_cano, uint8_t *valid,
                                 int32_t *idx, void *stream) {
  IA_CHECK_ARG(P >= 0, "ia_smpl_nn_deform: P < 0");
  if (P == 0) return IA_OK;
  IA_CHECK_ARG(pts && verts && T_inv && pts_cano && valid, "ia_smpl_nn_deform: null pointer");
  IA_CHECK_ARG(n_verts > 0 && (size_t)n_verts * 12 <= 160 * 1024 - 4096, "ia_smpl_nn_deform: %d vertices do not fit LDS", n_verts);
  return launch_nn(pts, P, n_pts_dev, verts, T_inv, n_verts, threshold, pts_cano, valid, idx, nullptr, nullptr, nullptr,
                   nullptr, (hipStream_t)stream);
}

// fused deform_test / deform_train of the SMPLDeformer: NN + transform + compaction -> field on the
// valid points -> per-point result (invalid: sigma = fill, rgb = 0)
extern "C" size_t ia_smpl_query_workspace_bytes(int P) {
  const size_t p = (size_t)(P > 0 ? P : 1);
  return ia_align(256) + ia_align(p * 4) + ia_align(p) + 2 * ia_align(p * 12) + ia_align(p * 4) + 1024;
}

extern "C" int ia_smpl_deform_query(const float *pts, int P, const int32_t *n_pts_dev, const float *verts,
                                    const float *T_inv, int n_verts, float threshold, const ia_field *field,
                                    float fill, int nan_to_num, float *rgb, float *sigma, void *ws, size_t ws_bytes,
                                    const void *nn_grid, void *stream) {
  IA_CHECK_ARG(P >= 0, "ia_smpl_deform_query: P < 0");
  if (P == 0) return IA_OK;
  IA_CHECK_ARG(pts && verts && T_inv && sigma && ws, "ia_smpl_deform_query: null pointer");
  IA_CHECK_ARG(n_verts > 0 && (size_t)n_verts * 12 <= 160 * 1024 - 4096, "ia_smpl_deform_query: %d vertices do not fit LDS", n_verts);
  if (ws_bytes < ia_smpl_query_workspace_bytes(P)) return ia_set_error(IA_ERR_WORKSPACE, "ia_smpl_deform_query: workspace too small");
  FieldDev F;
  int rc = ia_make_field_dev(field, &F);
  IA_CHECK_ARG(rc == 0, "ia_smpl_deform_query: bad field descriptor (%d)", rc);
  hipStream_t s = (hipStream_t)stream;
  WsCarver w(ws, ws_bytes);
  int32_t *n_cand = w.take<int32_t>(64);
  int32_t *pt_off = w.take<int32_t>(P);
  uint8_t *pt_cnt = w.take<uint8_t>(P);
  float *cand_xc = w.take<float>((size_t)P * 3);
  float *cand_rgb = w.take<float>((size_t)P * 3);
  float *cand_sigma = w.take<float>(P);
  zero_fill(n_cand, 4, s);
  // nn_grid (ia_smpl_nn_grid_build on THESE vertices with a cell >= threshold): the nearest vertex is looked for in the 27 cells
  // around the point only -- the same result for every point that has a vertex within `threshold`, nothing for the others
  if (nn_grid) rc = launch_nn_grid(pts, P, n_pts_dev, nn_grid, n_verts, T_inv, threshold, nullptr, cand_xc, pt_off, pt_cnt, n_cand, nullptr, s);
  else rc = launch_nn(pts, P, n_pts_dev, verts, T_inv, n_verts, threshold, nullptr, nullptr, nullptr, cand_xc, pt_off, pt_cnt, n_cand, s);
  if (rc) return rc;
  rc = ia_launch_field(cand_xc, P, n_cand, F, cand_rgb, cand_sigma, s, nullptr);
  if (rc) return rc;
  return ia_candidate_max(cand_rgb, cand_sigma, pt_off, pt_cnt, P, n_pts_dev, 1, fill, nan_to_num, rgb, sigma, stream);
}


// ---------------------------------------------------------------------------------------------------------------------
// The TRAINING query of the SMPLDeformer on compact samples (fit stage: fit.py, DNeRF.py:112-161 with deformer=smpl).
// Reference: smpl_deformer.py:88-120 -- knn_points(K = 1), `pts_cano = T_inv[idx] @ [pts, 1]`, the field on the valid points,
// (rgb, sigma) = (0, -1e5) elsewhere -- under autograd: the gradient reaches the per-vertex transforms (-> betas, pose,
// translation through `ia_smpl_lbs_bwd`) and, through the sample points, the rays (-> w2s: transform_rays_w2s is differentiable).
//   ia_smpl_nn_compact      nearest vertex + transform + compaction of the valid points: cand_xc, cand_pt (the point of every
//                           candidate), idx (the vertex of every point), pt_off / pt_cnt (0 | 1) for the compositor (n_init = 1)
//   ia_smpl_nn_compact_bwd  d cand_xc -> d T_inv[idx] += g [x, 1]^T (fp32 atomics: ~10^5 candidates onto 6 890 vertices),
//                           d pts = R^T g (zero for points without a candidate)
//   ia_ray_samples_bwd      pts = o + z d per compact sample (raymarcher_acc.py:158; z is not differentiated):
//                           d o[r] = sum_s d pts[s],  d d[r] = sum_s z[s] d pts[s]  -- one wave per ray, fixed order
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int ia_smpl_nn_compact(const float *pts, int P, const int32_t *n_pts_dev, const float *verts, const float *T_inv,
                                  int n_verts, float threshold, float *cand_xc, int32_t *cand_pt, int32_t *idx, int32_t *pt_off,
                                  uint8_t *pt_cnt, int32_t *n_cand, const void *nn_grid, void *stream) {
  IA_CHECK_ARG(P >= 0, "ia_smpl_nn_compact: P < 0");
  IA_CHECK_ARG(n_cand, "ia_smpl_nn_compact: null counter");
  hipStream_t s = (hipStream_t)stream;
  zero_fill(n_cand, 4, s);
  if (P == 0) return IA_OK;
  IA_CHECK_ARG(pts && verts && T_inv && cand_xc && cand_pt && idx && pt_off && pt_cnt, "ia_smpl_nn_compact: null pointer");
  IA_CHECK_ARG(n_verts > 0 && (size_t)n_verts * 12 <= 160 * 1024 - 4096, "ia_smpl_nn_compact: %d vertices do not fit LDS", n_verts);
  if (nn_grid) return launch_nn_grid(pts, P, n_pts_dev, nn_grid, n_verts, T_inv, threshold, idx, cand_xc, pt_off, pt_cnt, n_cand, cand_pt, s);
  return launch_nn(pts, P, n_pts_dev, verts, T_inv, n_verts, threshold, nullptr, nullptr, idx, cand_xc, pt_off, pt_cnt, n_cand, s, cand_pt);
}

__global__ __launch_bounds__(256) void k_smpl_nn_bwd(const float *__restrict__ pts, const int32_t *__restrict__ cand_pt,
                                                     const int32_t *__restrict__ idx, const int32_t *__restrict__ n_cand, int cap,
                                                     const float *__restrict__ T_inv, const float *__restrict__ d_cand_xc,
                                                     float *__restrict__ d_T_inv, float *__restrict__ d_pts) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= min(cap, *n_cand)) return;
  const int i = cand_pt[c], v = idx[i];
  const float g[3] = {d_cand_xc[(size_t)c * 3], d_cand_xc[(size_t)c * 3 + 1], d_cand_xc[(size_t)c * 3 + 2]};
  const float x[3] = {pts[(size_t)i * 3], pts[(size_t)i * 3 + 1], pts[(size_t)i * 3 + 2]};
  const float *T = T_inv + (size_t)v * 16;
  if (d_T_inv) {
    float *D = d_T_inv + (size_t)v * 16;
#pragma unroll
    for (int r = 0; r < 3; r++) {
      if (g[r] != 0.f) {
        atomicAdd(D + r * 4, g[r] * x[0]); atomicAdd(D + r * 4 + 1, g[r] * x[1]); atomicAdd(D + r * 4 + 2, g[r] * x[2]);
        atomicAdd(D + r * 4 + 3, g[r]);
      }
    }
  }
  if (d_pts)
#pragma unroll
    for (int b = 0; b < 3; b++) d_pts[(size_t)i * 3 + b] = T[b] * g[0] + T[4 + b] * g[1] + T[8 + b] * g[2];
}

extern "C" int ia_smpl_nn_compact_bwd(const float *pts, int P, const int32_t *cand_pt, const int32_t *idx, const int32_t *n_cand, int cap,
                                      const float *T_inv, int n_verts, const float *d_cand_xc, float *d_T_inv, float *d_pts,
                                      void *stream) {
  IA_CHECK_ARG(P >= 0 && cap >= 0 && n_verts > 0, "ia_smpl_nn_compact_bwd: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  if (d_T_inv) zero_fill(d_T_inv, (size_t)n_verts * 64, s);
  if (d_pts && P > 0) zero_fill(d_pts, (size_t)P * 12, s);
  if (cap == 0 || P == 0) return IA_OK;
  IA_CHECK_ARG(pts && cand_pt && idx && n_cand && T_inv && d_cand_xc, "ia_smpl_nn_compact_bwd: null pointer");
  hipLaunchKernelGGL(k_smpl_nn_bwd, dim3(ia_div_up(cap, 256)), dim3(256), 0, s, pts, cand_pt, idx, n_cand, cap, T_inv, d_cand_xc, d_T_inv, d_pts);
  IA_LAUNCH_CHECK("k_smpl_nn_bwd");
  return IA_OK;
}

#define IA_RS_RAYS 4
__global__ __launch_bounds__(64 * IA_RS_RAYS) void k_ray_samples_bwd(const int32_t *__restrict__ ray_off, const int32_t *__restrict__ ray_cnt,
                                                                     const float *__restrict__ s_z, const float *__restrict__ d_pts, int n_rays,
                                                                     float *__restrict__ d_o, float *__restrict__ d_d) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * IA_RS_RAYS + (threadIdx.x >> 6);
  if (n >= n_rays) return;   // uniform per wave
  const int off = ray_off[n], cnt = ray_cnt[n];
  float a[3] = {0.f, 0.f, 0.f}, b[3] = {0.f, 0.f, 0.f};
  for (int k = lane; k < cnt; k += 64) {
    const int s = off + k;
    const float z = s_z[s];
#pragma unroll
    for (int c = 0; c < 3; c++) { const float g = d_pts[(size_t)s * 3 + c]; a[c] += g; b[c] += z * g; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int c = 0; c < 3; c++) { a[c] += __shfl_xor(a[c], o, 64); b[c] += __shfl_xor(b[c], o, 64); }
  if (lane == 0)
#pragma unroll
    for (int c = 0; c < 3; c++) { d_o[(size_t)n * 3 + c] = a[c]; d_d[(size_t)n * 3 + c] = b[c]; }
}

extern "C" int ia_ray_samples_bwd(const int32_t *ray_off, const int32_t *ray_cnt, const float *s_z, const float *d_pts, int n_rays,
                                  float *d_o, float *d_d, void *stream) {
  IA_CHECK_ARG(n_rays >= 0, "ia_ray_samples_bwd: n_rays < 0");
  if (n_rays == 0) return IA_OK;
  IA_CHECK_ARG(ray_off && ray_cnt && s_z && d_pts && d_o && d_d, "ia_ray_samples_bwd: null pointer");
  hipLaunchKernelGGL(k_ray_samples_bwd, dim3(ia_div_up(n_rays, IA_RS_RAYS)), dim3(64 * IA_RS_RAYS), 0, (hipStream_t)stream, ray_off, ray_cnt,
                     s_z, d_pts, n_rays, d_o, d_d);
  IA_LAUNCH_CHECK("k_ray_samples_bwd");
  return IA_OK;
}
