// ia_mesh.hip -- the `smpl_init` bootstrap of the training occupancy grid
// (instant_avatar/models/structures/density_grid.py:53-75, reached through Raymarcher(smpl_init=True),
// confs/demo.yaml:37): signed distance of the grid-cell centres to the posed SMPL mesh; cells with
// signed distance < 0.01 start occupied.
//
// The reference gets the two ingredients from NVIDIA kaolin (absent here, not installable offline):
//   kaolin.metrics.trianglemesh.point_to_mesh_distance -> squared distance to the closest triangle
//   kaolin.ops.mesh.check_sign                          -> inside / outside of a watertight mesh
// Restated from their definitions -- parity with kaolin's floating-point details is unpinned: exact
// point-triangle distance by the closest-point regions of the triangle (Ericson, Real-Time Collision
// Detection 5.1.5) and inside = odd number of crossings of the ray p + t (1,0,0), t > 0, with the mesh
// (projected-triangle test in double precision with a consistent edge rule, so that a ray through a shared
// edge counts once).  Brute force over all triangles, staged through LDS: 64^3 points x 13 776 faces is
// 3.6 G pair tests, a few milliseconds, once per training frame.
#include "ia_common.h"

#define IA_MESH_CHUNK 1024  // triangles per LDS stage (36 KB)

__device__ __forceinline__ float tri_dist2(const float *p, const float *a, const float *b, const float *c) {
  const float ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, ac[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
  const float ap[3] = {p[0] - a[0], p[1] - a[1], p[2] - a[2]};
  auto dot = [](const float *u, const float *v) { return u[0] * v[0] + u[1] * v[1] + u[2] * v[2]; };
  float q[3];
  const float d1 = dot(ab, ap), d2 = dot(ac, ap);
  if (d1 <= 0.f && d2 <= 0.f) { q[0] = a[0]; q[1] = a[1]; q[2] = a[2]; }
  else {
    const float bp[3] = {p[0] - b[0], p[1] - b[1], p[2] - b[2]};
    const float d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0.f && d4 <= d3) { q[0] = b[0]; q[1] = b[1]; q[2] = b[2]; }
    else {
      const float vc = d1 * d4 - d3 * d2;
      if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) {
        const float v = d1 / (d1 - d3);
        for (int k = 0; k < 3; k++) q[k] = a[k] + v * ab[k];
      } else {
        const float cp[3] = {p[0] - c[0], p[1] - c[1], p[2] - c[2]};
        const float d5 = dot(ab, cp), d6 = dot(ac, cp);
        if (d6 >= 0.f && d5 <= d6) { q[0] = c[0]; q[1] = c[1]; q[2] = c[2]; }
        else {
          const float vb = d5 * d2 - d1 * d6;
          if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) {
            const float w = d2 / (d2 - d6);
            for (int k = 0; k < 3; k++) q[k] = a[k] + w * ac[k];
          } else {
            const float va = d3 * d6 - d5 * d4;
            if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) {
              const float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
              for (int k = 0; k < 3; k++) q[k] = b[k] + w * (c[k] - b[k]);
            } else {
              const float denom = 1.f / (va + vb + vc);
              const float v = vb * denom, w = vc * denom;
              for (int k = 0; k < 3; k++) q[k] = a[k] + ab[k] * v + ac[k] * w;
            }
          }
        }
      }
    }
  }
  const float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
  return dx * dx + dy * dy + dz * dz;
}

// does the ray p + t (1,0,0), t > 0, cross triangle abc?  Projection onto (y, z); an edge belongs to the
// triangle on whose side the half-open rule puts it, so a ray through a shared edge is counted exactly once.
__device__ __forceinline__ bool ray_x_crosses(const float *p, const float *a, const float *b, const float *c) {
  const double py = p[1], pz = p[2];
  auto edge = [&](const float *u, const float *v, double &e, bool &top_left) {
    const double uy = u[1], uz = u[2], vy = v[1], vz = v[2];
    e = (vy - uy) * (pz - uz) - (vz - uz) * (py - uy);
    top_left = (vz == uz) ? (vy < uy) : (vz < uz);   // tie rule on the edge itself
  };
  double e0, e1, e2;
  bool t0, t1, t2;
  edge(a, b, e0, t0); edge(b, c, e1, t1); edge(c, a, e2, t2);
  const double area = ((double)b[1] - a[1]) * ((double)c[2] - a[2]) - ((double)b[2] - a[2]) * ((double)c[1] - a[1]);
  if (area == 0.0) return false;  // triangle seen edge-on
  const double s = area > 0.0 ? 1.0 : -1.0;
  e0 *= s; e1 *= s; e2 *= s;
  if (area < 0.0) { t0 = !t0; t1 = !t1; t2 = !t2; }
  const bool in = (e0 > 0.0 || (e0 == 0.0 && t0)) && (e1 > 0.0 || (e1 == 0.0 && t1)) && (e2 > 0.0 || (e2 == 0.0 && t2));
  if (!in) return false;
  const double A = fabs(area);
  const double x = (e1 * a[0] + e2 * b[0] + e0 * c[0]) / A;  // barycentric: e1 ~ weight of a, e2 ~ b, e0 ~ c
  return x > (double)p[0];
}

__global__ __launch_bounds__(256) void k_mesh_sdf(const float *__restrict__ pts, long N, const float *__restrict__ verts,
                                                  const int32_t *__restrict__ faces, int F, float *__restrict__ sdf) {
  __shared__ float s_tri[IA_MESH_CHUNK][9];
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float p[3] = {0.f, 0.f, 0.f};
  if (i < N) { p[0] = pts[i * 3]; p[1] = pts[i * 3 + 1]; p[2] = pts[i * 3 + 2]; }
  float best = INFINITY;
  int crossings = 0;
  for (int f0 = 0; f0 < F; f0 += IA_MESH_CHUNK) {
    const int nf = min(IA_MESH_CHUNK, F - f0);
    __syncthreads();
    for (int e = threadIdx.x; e < nf * 3; e += blockDim.x) {
      const int v = faces[(size_t)f0 * 3 + e];
      s_tri[e / 3][(e % 3) * 3 + 0] = verts[(size_t)v * 3]; s_tri[e / 3][(e % 3) * 3 + 1] = verts[(size_t)v * 3 + 1];
      s_tri[e / 3][(e % 3) * 3 + 2] = verts[(size_t)v * 3 + 2];
    }
    __syncthreads();
    if (i < N) {
      for (int f = 0; f < nf; f++) {
        const float *t = s_tri[f];
        best = fminf(best, tri_dist2(p, t, t + 3, t + 6));
        crossings += ray_x_crosses(p, t, t + 3, t + 6) ? 1 : 0;
      }
    }
  }
  if (i < N) sdf[i] = ((crossings & 1) ? -1.f : 1.f) * sqrtf(best);  // density_grid.py:62-70
}

// cell centres of a G^3 grid: denormalize(coords + 0.5 / G, aabb)  (density_grid.py:55)
__global__ __launch_bounds__(256) void k_cell_centres(int G, const float *__restrict__ aabb, float *__restrict__ pts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= G * G * G) return;
  const int idx[3] = {i / (G * G), i / G % G, i % G};
#pragma unroll
  for (int d = 0; d < 3; d++) pts[(size_t)i * 3 + d] = ((float)idx[d] / (float)G + 0.5f / (float)G) * (aabb[3 + d] - aabb[d]) + aabb[d];
}

extern "C" int ia_mesh_signed_distance(const float *pts, long N, const float *verts, const int32_t *faces, int n_faces,
                                       float *sdf, void *stream) {
  IA_CHECK_ARG(N >= 0 && n_faces > 0, "ia_mesh_signed_distance: bad sizes");
  if (N == 0) return IA_OK;
  IA_CHECK_ARG(pts && verts && faces && sdf, "ia_mesh_signed_distance: null pointer");
  hipLaunchKernelGGL(k_mesh_sdf, dim3(ia_div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, pts, N, verts, faces, n_faces, sdf);
  IA_LAUNCH_CHECK("k_mesh_sdf");
  return IA_OK;
}

extern "C" int ia_grid_cell_centres(int G, const float *aabb, float *pts, void *stream) {
  IA_CHECK_ARG(G > 0 && aabb && pts, "ia_grid_cell_centres: bad arguments");
  hipLaunchKernelGGL(k_cell_centres, dim3(ia_div_up((long)G * G * G, 256)), dim3(256), 0, (hipStream_t)stream, G, aabb, pts);
  IA_LAUNCH_CHECK("k_cell_centres");
  return IA_OK;
}
