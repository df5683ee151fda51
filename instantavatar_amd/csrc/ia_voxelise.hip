// ia_voxelise.hip -- one-time skinning-weight voxelisation (row a20 of SURVEY.md section 8):
// ForwardDeformer.switch_to_explicit -> query_weights_smpl
// (fast_snarf/deformer_torch.py:225-244; the K-NN comes from third_parties/pytorch3d knn_points).
//
//   k_knn_blend   per voxel: exact 30 nearest SMPL vertices (brute force, vertices staged
//                 in LDS, per-thread candidate list in LDS), inverse-distance blend of their
//                 skinning weights (distance clamped to [1e-4, 1])
//   k_smooth      one pass of the 6-neighbour smoothing (interior voxels) + renormalisation
#include "ia_common.h"

#define KNN_K 30
#define KNN_THREADS 256
#define KNN_CHUNK 2048  // vertices staged per LDS chunk

__global__ __launch_bounds__(KNN_THREADS) void k_knn_blend(const float *__restrict__ pts, int N,
                                                           const float *__restrict__ verts, int Vn,
                                                           const float *__restrict__ vw,  // [Vn,24]
                                                           float *__restrict__ out /*[24,N]*/) {
  __shared__ float s_v[KNN_CHUNK][3];
  __shared__ float s_d[KNN_K][KNN_THREADS];   // per-thread top-K (unsorted), column = thread
  __shared__ int s_i[KNN_K][KNN_THREADS];
  const int t = threadIdx.x;
  const int q = blockIdx.x * KNN_THREADS + t;
  const bool live = q < N;
  float px = 0.f, py = 0.f, pz = 0.f;
  if (live) { px = pts[(size_t)q * 3]; py = pts[(size_t)q * 3 + 1]; pz = pts[(size_t)q * 3 + 2]; }
  for (int k = 0; k < KNN_K; k++) { s_d[k][t] = INFINITY; s_i[k][t] = 0; }
  float worst = INFINITY;
  int worst_k = 0;
  for (int v0 = 0; v0 < Vn; v0 += KNN_CHUNK) {
    const int nv = min(KNN_CHUNK, Vn - v0);
    __syncthreads();
    for (int e = t; e < nv * 3; e += KNN_THREADS) (&s_v[0][0])[e] = verts[(size_t)v0 * 3 + e];
    __syncthreads();
    if (live) {
      for (int v = 0; v < nv; v++) {
        const float dx = px - s_v[v][0], dy = py - s_v[v][1], dz = pz - s_v[v][2];
        const float d = dx * dx + dy * dy + dz * dz;
        if (d < worst) {  // replace the current worst, then find the new worst
          s_d[worst_k][t] = d; s_i[worst_k][t] = v0 + v;
          worst = -1.f;
          // the entry to evict next: largest distance, among equal distances the LARGEST vertex index (pytorch3d's
          // priority queue of (distance, index) tuples pops exactly that one, knn_cpu.cpp:36-56)
          int worst_i = -1;
          for (int k = 0; k < KNN_K; k++) {
            const float dk = s_d[k][t]; const int ik = s_i[k][t];
            if (dk > worst || (dk == worst && ik > worst_i)) { worst = dk; worst_k = k; worst_i = ik; }
          }
        }
      }
    }
  }
  if (!live) return;
  // sort ascending (selection sort in LDS), as knn_points returns them
  for (int a = 0; a < KNN_K - 1; a++) {
    int m = a; float dm = s_d[a][t];
    for (int b = a + 1; b < KNN_K; b++) { const float db = s_d[b][t]; if (db < dm || (db == dm && s_i[b][t] < s_i[m][t])) { dm = db; m = b; } }
    const float td = s_d[a][t]; const int ti = s_i[a][t];
    s_d[a][t] = s_d[m][t]; s_i[a][t] = s_i[m][t]; s_d[m][t] = td; s_i[m][t] = ti;
  }
  // :228-232  dist = sqrt(d).clamp(1e-4, 1); ws = 1/dist; ws /= sum(ws)
  float wsum = 0.f;
  for (int k = 0; k < KNN_K; k++) {
    float dd = sqrtf(s_d[k][t]);
    dd = dd < 0.0001f ? 0.0001f : (dd > 1.f ? 1.f : dd);
    const float w = 1.f / dd;
    s_d[k][t] = w;
    wsum += w;
  }
  float acc[24];
#pragma unroll
  for (int j = 0; j < 24; j++) acc[j] = 0.f;
  for (int k = 0; k < KNN_K; k++) {
    const float w = s_d[k][t] / wsum;
    const float4 *r = reinterpret_cast<const float4 *>(vw + (size_t)s_i[k][t] * 24);
#pragma unroll
    for (int j4 = 0; j4 < 6; j4++) {
      const float4 a = r[j4];
      acc[4 * j4] += w * a.x; acc[4 * j4 + 1] += w * a.y; acc[4 * j4 + 2] += w * a.z; acc[4 * j4 + 3] += w * a.w;
    }
  }
#pragma unroll
  for (int j = 0; j < 24; j++) out[(size_t)j * N + q] = acc[j];
}

// one smoothing pass (deformer_torch.py:237-243): interior voxels move 30 % towards the mean of
// their 6 neighbours (Jacobi), then every voxel is renormalised over the 24 joints
__global__ __launch_bounds__(256) void k_smooth(const float *__restrict__ in, float *__restrict__ out, int d, int h,
                                                int w) {
  const int n = d * h * w;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int z = i / (h * w), y = i / w % h, x = i % w;
  const bool interior = z > 0 && z < d - 1 && y > 0 && y < h - 1 && x > 0 && x < w - 1;
  float v[24];
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < 24; c++) {
    const float *p = in + (size_t)c * n;
    float val = p[i];
    if (interior) {
      const float mean = (p[i + h * w] + p[i - h * w] + p[i + w] + p[i - w] + p[i + 1] + p[i - 1]) / 6.0f;
      val = (val - mean) * 0.7f + mean;
    }
    v[c] = val;
    sum += val;
  }
#pragma unroll
  for (int c = 0; c < 24; c++) out[(size_t)c * n + i] = v[c] / sum;
}

extern "C" size_t ia_voxelise_workspace_bytes(int d, int h, int w) { return ia_align((size_t)24 * d * h * w * 4) + 256; }

extern "C" int ia_voxelise_weights(const float *pts, const float *verts, int n_verts, const float *vert_weights, int d,
                                   int h, int w, int n_smooth, float *voxel_w, void *ws, size_t ws_bytes,
                                   void *stream) {
  IA_CHECK_ARG(pts && verts && vert_weights && voxel_w && ws, "ia_voxelise_weights: null pointer");
  IA_CHECK_ARG(d > 2 && h > 2 && w > 2 && n_verts >= KNN_K && n_smooth >= 0, "ia_voxelise_weights: bad sizes");
  if (ws_bytes < ia_voxelise_workspace_bytes(d, h, w)) return ia_set_error(IA_ERR_WORKSPACE, "ia_voxelise_weights: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int n = d * h * w;
  float *tmp = (float *)ws;
  // ping-pong so that the final result lands in voxel_w
  float *a = (n_smooth & 1) ? tmp : voxel_w, *b = (n_smooth & 1) ? voxel_w : tmp;
  hipLaunchKernelGGL(k_knn_blend, dim3(ia_div_up(n, KNN_THREADS)), dim3(KNN_THREADS), 0, s, pts, n, verts, n_verts,
                     vert_weights, a);
  for (int it = 0; it < n_smooth; it++) {
    hipLaunchKernelGGL(k_smooth, dim3(ia_div_up(n, 256)), dim3(256), 0, s, a, b, d, h, w);
    float *t = a; a = b; b = t;
  }
  IA_LAUNCH_CHECK("ia_voxelise_weights");
  return IA_OK;
}
