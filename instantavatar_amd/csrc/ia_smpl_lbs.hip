// ia_smpl_lbs.hip -- the SMPL body model of the SMPLDeformer plugin, forward and backward, as a handful of launches (gfx950).
//
// Reference: deformers/smpl_deformer.py:32-77 (`initialize` + `prepare_deformer`): two evaluations of smplx's SMPL.forward
// (body_models.py:289-372 -> lbs.py:152-250: shape blend, joint regression, Rodrigues, pose-corrective blend, the 24-joint
// kinematic chain, per-vertex blend T = W A, skinning) -- one in the template pose, one in the frame's pose -- then two batched
// 4x4 inverses and a product per vertex:
//     T_inv[v] = T_t[v] . ( T[v]^-1 . s2w  with  t += po_t[v] - po[v] (+ so_t[v] - so[v] = 0) ),   s2w = A_0,   w2s = s2w^-1
//     vertices[v] = w2s . T[v] (v_shaped[v] + po[v])
// In the fit stage (fit.py; DNeRF.py:113-128) this runs under autograd every step -- betas, pose and translation are optimised --
// which is ~1 200 small torch launches forward + backward (7.5 of the 13 ms of a fit step, tools/prof_fit_split.py).  Here:
//
//   forward   k_lbs_chain_fwd   one wave: J = J0 + JS beta, Rodrigues, both chains -> A, A_t, pose feature, s2w, w2s
//             k_lbs_vertex_fwd  one thread per vertex: pose blend (207 x 3 MAC, coalesced rows of posedirs), shape blend, the two
//                               transform blends, closed-form affine inverse, T_inv, the posed vertex in the SMPL-root frame
//   backward  k_lbs_vertex_bwd  one thread per vertex: d T_inv[v] -> d T[v], d T_t[v], d po[v], its share of d s2w
//             k_lbs_reduce      deterministic sums over the vertices (fixed order, no atomics): d A_j = sum_v w[v,j] d T[v],
//                               d A_t,j, d s2w, d pf[k] = sum_v posedirs[k, v, :] . d po[v]
//             k_lbs_chain_bwd   one wave: both chains reversed (children before parents), Rodrigues backward -> d pose [72],
//                               d J -> d betas [10], d transl [3]
// The joint regression uses J_regressor (v_template + shapedirs beta) = J0 + JS beta with J0, JS folded once per subject.
// Arithmetic is fp32 with explicit operation order (-ffp-contract=off); results are checked against the oracle's restatement
// (pinned to the reference's lbs.py) and, for the gradients, against autograd through the lbs.py-style torch ops.
#include "ia_common.h"

#define IA_LBS_THREADS 256

struct LbsBodyDev {
  const float *v_template, *shapedirs, *posedirs, *lbs_weights, *J0, *JS, *po_t;
  const int32_t *parents;
  int V;
};

// workspace layout (floats)
struct LbsWs {
  float *J, *A, *At, *pf, *S, *W;                    // chain outputs: [72] [288] [288] [207] [12] [12]
  float *dT, *dTt, *dpo, *dSv;                       // per-vertex gradients: [V,12] [V,12] [V,3] [V,12]
  float *dA, *dAt, *dpf, *dS;                        // reduced: [288] [288] [207] [12]
};
static inline size_t lbs_ws_floats(int V) { return 1024 + (size_t)V * 39 + 1024; }
static inline LbsWs lbs_carve(void *ws, int V) {
  float *p = (float *)ws;
  LbsWs w;
  w.J = p; w.A = p + 72; w.At = p + 360; w.pf = p + 648; w.S = p + 855; w.W = p + 867;   // 879 < 1024
  p += 1024;
  w.dT = p; p += (size_t)V * 12; w.dTt = p; p += (size_t)V * 12; w.dpo = p; p += (size_t)V * 3; w.dSv = p; p += (size_t)V * 12;
  w.dA = p; w.dAt = p + 288; w.dpf = p + 576; w.dS = p + 783;
  return w;
}
extern "C" size_t ia_smpl_lbs_workspace_bytes(int n_verts) { return ia_align(lbs_ws_floats(n_verts > 0 ? n_verts : 1) * sizeof(float)); }

// ---- small affine helpers (3x4 row-major: [R | t], element (a, b) at a * 4 + b) -------------------------------------
__device__ __forceinline__ void aff_mul(const float *X, const float *Y, float *Z) {   // Z = X . Y
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int b = 0; b < 3; b++) Z[a * 4 + b] = X[a * 4] * Y[b] + X[a * 4 + 1] * Y[4 + b] + X[a * 4 + 2] * Y[8 + b];
    Z[a * 4 + 3] = X[a * 4] * Y[3] + X[a * 4 + 1] * Y[7] + X[a * 4 + 2] * Y[11] + X[a * 4 + 3];
  }
}
// adj(M) / det(M) by cross products of the rows, t' = -M^-1 t  (the product's `affine_inverse`, snarf_deformer.py)
__device__ __forceinline__ void aff_inv(const float *X, float *Y) {
  const float r0[3] = {X[0], X[1], X[2]}, r1[3] = {X[4], X[5], X[6]}, r2[3] = {X[8], X[9], X[10]};
  const float c0[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
  const float c1[3] = {r2[1] * r0[2] - r2[2] * r0[1], r2[2] * r0[0] - r2[0] * r0[2], r2[0] * r0[1] - r2[1] * r0[0]};
  const float c2[3] = {r0[1] * r1[2] - r0[2] * r1[1], r0[2] * r1[0] - r0[0] * r1[2], r0[0] * r1[1] - r0[1] * r1[0]};
  const float det = r0[0] * c0[0] + r0[1] * c0[1] + r0[2] * c0[2];
#pragma unroll
  for (int a = 0; a < 3; a++) { Y[a * 4] = c0[a] / det; Y[a * 4 + 1] = c1[a] / det; Y[a * 4 + 2] = c2[a] / det; }
#pragma unroll
  for (int a = 0; a < 3; a++) Y[a * 4 + 3] = -(Y[a * 4] * X[3] + Y[a * 4 + 1] * X[7] + Y[a * 4 + 2] * X[11]);
}
// Y = X^-1 (affine), dY given -> dX:  G^ = dY.R - dY.t X.t^T;  dX.R = -Y.R^T G^ Y.R^T;  dX.t = -Y.R^T dY.t
__device__ __forceinline__ void aff_inv_bwd(const float *X, const float *Y, const float *dY, float *dX) {
  float Gh[9];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 3; b++) Gh[a * 3 + b] = dY[a * 4 + b] - dY[a * 4 + 3] * X[b * 4 + 3];
  float U[9];   // U = Y.R^T G^
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 3; b++) U[a * 3 + b] = Y[a] * Gh[b] + Y[4 + a] * Gh[3 + b] + Y[8 + a] * Gh[6 + b];
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int b = 0; b < 3; b++) dX[a * 4 + b] = -(U[a * 3] * Y[b * 4] + U[a * 3 + 1] * Y[b * 4 + 1] + U[a * 3 + 2] * Y[b * 4 + 2]);   // (U Y.R^T)[a][b]
    dX[a * 4 + 3] = -(Y[a] * dY[3] + Y[4 + a] * dY[7] + Y[8 + a] * dY[11]);
  }
}

// ---- Rodrigues (lbs.py:295-329): angle = |theta + 1e-8|, dir = theta / angle, R = I + sin K + (1 - cos) K^2 -----------
struct Rod { float ang, sn, cs, K[9], KK[9], R[9]; };
__device__ __forceinline__ void rodrigues(const float *th, Rod &r) {
  const float ax = th[0] + 1e-8f, ay = th[1] + 1e-8f, az = th[2] + 1e-8f;
  r.ang = sqrtf(ax * ax + ay * ay + az * az);
  const float d0 = th[0] / r.ang, d1 = th[1] / r.ang, d2 = th[2] / r.ang;
  r.cs = cosf(r.ang); r.sn = sinf(r.ang);
  const float k[9] = {0, -d2, d1, d2, 0, -d0, -d1, d0, 0};
  for (int a = 0; a < 9; a++) r.K[a] = k[a];
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      float v = 0.f;
      for (int q = 0; q < 3; q++) v += r.K[a * 3 + q] * r.K[q * 3 + b];
      r.KK[a * 3 + b] = v;
    }
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) r.R[a * 3 + b] = (a == b ? 1.f : 0.f) + r.sn * r.K[a * 3 + b] + (1.f - r.cs) * r.KK[a * 3 + b];
}
__device__ __forceinline__ void rodrigues_bwd(const float *th, const Rod &r, const float *dR, float *dth) {
  float dK[9];
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      float v = r.sn * dR[a * 3 + b];
      for (int q = 0; q < 3; q++) v += (1.f - r.cs) * (dR[a * 3 + q] * r.K[b * 3 + q] + r.K[q * 3 + a] * dR[q * 3 + b]);   // dR K^T + K^T dR
      dK[a * 3 + b] = v;
    }
  float dRK = 0.f, dRKK = 0.f;
  for (int a = 0; a < 9; a++) { dRK += dR[a] * r.K[a]; dRKK += dR[a] * r.KK[a]; }
  const float d_ang = r.cs * dRK + r.sn * dRKK;
  const float d_dir[3] = {dK[7] - dK[5], dK[2] - dK[6], dK[3] - dK[1]};
  const float dot = d_dir[0] * th[0] + d_dir[1] * th[1] + d_dir[2] * th[2];
  const float coef = d_ang - dot / (r.ang * r.ang);
  for (int a = 0; a < 3; a++) dth[a] = d_dir[a] / r.ang + coef * (th[a] + 1e-8f) / r.ang;
}

// One kinematic chain in LDS: L_j = [R_j | rel_j], G_0 = L_0, G_j = G_p L_j (lbs.py:345-401), sequentially over the joints with
// lane (a, b) of the first twelve owning one element.  R [24][9], Jn [24][3] -> G [24][12] (rows 0..2 of the 4x4).
__device__ __forceinline__ void chain_forward(const float (*R)[9], const float (*Jn)[3], const int *par, float (*G)[12], int j) {
  if (j < 12) {
    const int a = j >> 2, b = j & 3;
    G[0][j] = b < 3 ? R[0][a * 3 + b] : Jn[0][a];
  }
  __syncthreads();
  for (int i = 1; i < 24; i++) {
    if (j < 12) {
      const int a = j >> 2, b = j & 3, p = par[i];
      float acc;
      if (b < 3) acc = G[p][a * 4] * R[i][b] + G[p][a * 4 + 1] * R[i][3 + b] + G[p][a * 4 + 2] * R[i][6 + b];
      else acc = G[p][a * 4] * (Jn[i][0] - Jn[p][0]) + G[p][a * 4 + 1] * (Jn[i][1] - Jn[p][1]) + G[p][a * 4 + 2] * (Jn[i][2] - Jn[p][2]) + G[p][a * 4 + 3];
      G[i][j] = acc;
    }
    __syncthreads();
  }
}

__global__ void k_lbs_chain_fwd(LbsBodyDev B, const float *__restrict__ betas, const float *__restrict__ pose,
                                const float *__restrict__ transl, const float *__restrict__ pose_t, LbsWs w) {
  __shared__ float R[24][9], Rt[24][9], Jn[24][3], G[24][12], Gt[24][12];
  __shared__ int par[24];
  const int j = threadIdx.x;
  if (j < 24) {
    par[j] = B.parents[j];
    for (int c = 0; c < 3; c++) {   // J = J_regressor (v_template + shapedirs beta) = J0 + JS beta (lbs.py:185-190)
      float v = B.J0[j * 3 + c];
      for (int l = 0; l < 10; l++) v += B.JS[(j * 3 + c) * 10 + l] * betas[l];
      Jn[j][c] = v;
      w.J[j * 3 + c] = v;
    }
    Rod r;
    rodrigues(pose + j * 3, r);
    for (int a = 0; a < 9; a++) R[j][a] = r.R[a];
    rodrigues(pose_t + j * 3, r);
    for (int a = 0; a < 9; a++) Rt[j][a] = r.R[a];
    if (j >= 1)   // pose_feature = (rot_mats[:, 1:] - I).view(-1) (lbs.py:211-213)
      for (int a = 0; a < 9; a++) w.pf[(j - 1) * 9 + a] = R[j][a] - ((a == 0 || a == 4 || a == 8) ? 1.f : 0.f);
  }
  __syncthreads();
  chain_forward(R, Jn, par, G, j);
  chain_forward(Rt, Jn, par, Gt, j);
  if (j < 24) {
    // rel_transforms = transforms - pad(transforms @ [J, 0]) (lbs.py:396-399); transl folded into A (body_models.py:353-357)
    for (int a = 0; a < 3; a++) {
      const float t = G[j][a * 4] * Jn[j][0] + G[j][a * 4 + 1] * Jn[j][1] + G[j][a * 4 + 2] * Jn[j][2];
      const float tt = Gt[j][a * 4] * Jn[j][0] + Gt[j][a * 4 + 1] * Jn[j][1] + Gt[j][a * 4 + 2] * Jn[j][2];
      for (int b = 0; b < 3; b++) { w.A[j * 12 + a * 4 + b] = G[j][a * 4 + b]; w.At[j * 12 + a * 4 + b] = Gt[j][a * 4 + b]; }
      w.A[j * 12 + a * 4 + 3] = G[j][a * 4 + 3] - t + (transl ? transl[a] : 0.f);
      w.At[j * 12 + a * 4 + 3] = Gt[j][a * 4 + 3] - tt;
    }
  }
  __syncthreads();
  if (j == 0) {
    float S[12], W[12];
    for (int a = 0; a < 12; a++) S[a] = w.A[a];
    aff_inv(S, W);
    for (int a = 0; a < 12; a++) { w.S[a] = S[a]; w.W[a] = W[a]; }
  }
}

// everything of one vertex that both directions need
struct VertexFwd { float po[3], vs[3], T[12], Tt[12], Ti[12], M[12]; };
__device__ __forceinline__ void vertex_forward(const LbsBodyDev &B, const LbsWs &w, const float *__restrict__ betas, int v, VertexFwd &o) {
  // pose-corrective blend (lbs.py:216-219): pose_feature [207] x posedirs [207, V*3]
  float p0 = 0.f, p1 = 0.f, p2 = 0.f;
  const float *pd = B.posedirs + (size_t)v * 3;
  const size_t row = (size_t)B.V * 3;
#pragma unroll 9
  for (int k = 0; k < 207; k++) {
    const float f = w.pf[k];
    p0 += f * pd[k * row]; p1 += f * pd[k * row + 1]; p2 += f * pd[k * row + 2];
  }
  o.po[0] = p0; o.po[1] = p1; o.po[2] = p2;
  for (int c = 0; c < 3; c++) {   // shape blend (lbs.py:185-187)
    float s = 0.f;
    for (int l = 0; l < 10; l++) s += betas[l] * B.shapedirs[((size_t)v * 3 + c) * 10 + l];
    o.vs[c] = B.v_template[(size_t)v * 3 + c] + s;
  }
  for (int c = 0; c < 12; c++) { o.T[c] = 0.f; o.Tt[c] = 0.f; }
  for (int jn = 0; jn < 24; jn++) {   // T = W A (lbs.py:227-230), joints in order
    const float wt = B.lbs_weights[(size_t)v * 24 + jn];
    for (int c = 0; c < 12; c++) { o.T[c] += wt * w.A[jn * 12 + c]; o.Tt[c] += wt * w.At[jn * 12 + c]; }
  }
  aff_inv(o.T, o.Ti);
  aff_mul(o.Ti, w.S, o.M);      // T^-1 . s2w (smpl_deformer.py:70)
  for (int a = 0; a < 3; a++) o.M[a * 4 + 3] += B.po_t[(size_t)v * 3 + a] - o.po[a];   // :71-73 (the shape offsets cancel: same betas)
}

__global__ __launch_bounds__(IA_LBS_THREADS) void k_lbs_vertex_fwd(LbsBodyDev B, const float *__restrict__ betas, LbsWs w,
                                                                    float *__restrict__ T_inv, float *__restrict__ verts,
                                                                    float *__restrict__ verts_t, float *__restrict__ w2s_out) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v == 0 && w2s_out) {
    for (int a = 0; a < 12; a++) w2s_out[a] = w.W[a];
    w2s_out[12] = 0.f; w2s_out[13] = 0.f; w2s_out[14] = 0.f; w2s_out[15] = 1.f;
  }
  if (v >= B.V) return;
  VertexFwd f;
  vertex_forward(B, w, betas, v, f);
  float Tv[12];
  aff_mul(f.Tt, f.M, Tv);         // T_template . T_inv (:74)
  float *o = T_inv + (size_t)v * 16;
  for (int c = 0; c < 12; c++) o[c] = Tv[c];
  o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
  float vp[3], x[3];
  for (int a = 0; a < 3; a++) vp[a] = f.vs[a] + f.po[a];
  for (int a = 0; a < 3; a++) x[a] = f.T[a * 4] * vp[0] + f.T[a * 4 + 1] * vp[1] + f.T[a * 4 + 2] * vp[2] + f.T[a * 4 + 3];   // lbs.py:232-236 (+ transl)
  if (verts)     // vertices in the SMPL-root frame (smpl_deformer.py:76)
    for (int a = 0; a < 3; a++) verts[(size_t)v * 3 + a] = w.W[a * 4] * x[0] + w.W[a * 4 + 1] * x[1] + w.W[a * 4 + 2] * x[2] + w.W[a * 4 + 3];
  if (verts_t) { // the template-pose vertices (initialize, :36-41): bbox
    for (int a = 0; a < 3; a++) vp[a] = f.vs[a] + B.po_t[(size_t)v * 3 + a];
    for (int a = 0; a < 3; a++) verts_t[(size_t)v * 3 + a] = f.Tt[a * 4] * vp[0] + f.Tt[a * 4 + 1] * vp[1] + f.Tt[a * 4 + 2] * vp[2] + f.Tt[a * 4 + 3];
  }
}

__global__ __launch_bounds__(IA_LBS_THREADS) void k_lbs_vertex_bwd(LbsBodyDev B, const float *__restrict__ betas, LbsWs w,
                                                                    const float *__restrict__ d_T_inv) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= B.V) return;
  VertexFwd f;
  vertex_forward(B, w, betas, v, f);
  const float *G = d_T_inv + (size_t)v * 16;
  // T_inv = T_t . M:  dT_t.R = G.R M.R^T + G.t M.t^T;  dT_t.t = G.t;  dM.R = T_t.R^T G.R;  dM.t = T_t.R^T G.t
  float dTt[12], dM[12];
  for (int a = 0; a < 3; a++) {
    for (int b = 0; b < 3; b++) {
      dTt[a * 4 + b] = G[a * 4] * f.M[b * 4] + G[a * 4 + 1] * f.M[b * 4 + 1] + G[a * 4 + 2] * f.M[b * 4 + 2] + G[a * 4 + 3] * f.M[b * 4 + 3];
      dM[a * 4 + b] = f.Tt[a] * G[b] + f.Tt[4 + a] * G[4 + b] + f.Tt[8 + a] * G[8 + b];
    }
    dTt[a * 4 + 3] = G[a * 4 + 3];
    dM[a * 4 + 3] = f.Tt[a] * G[3] + f.Tt[4 + a] * G[7] + f.Tt[8 + a] * G[11];
  }
  // M = T^-1 S with t += po_t - po:  d po = -dM.t;  N = Ti S:  dTi.R = dN.R S.R^T + dN.t S.t^T;  dTi.t = dN.t;  dS = Ti.R^T dN
  float dTi[12], dS[12];
  for (int a = 0; a < 3; a++) {
    for (int b = 0; b < 3; b++) {
      dTi[a * 4 + b] = dM[a * 4] * w.S[b * 4] + dM[a * 4 + 1] * w.S[b * 4 + 1] + dM[a * 4 + 2] * w.S[b * 4 + 2] + dM[a * 4 + 3] * w.S[b * 4 + 3];
      dS[a * 4 + b] = f.Ti[a] * dM[b] + f.Ti[4 + a] * dM[4 + b] + f.Ti[8 + a] * dM[8 + b];
    }
    dTi[a * 4 + 3] = dM[a * 4 + 3];
    dS[a * 4 + 3] = f.Ti[a] * dM[3] + f.Ti[4 + a] * dM[7] + f.Ti[8 + a] * dM[11];
  }
  float dT[12];
  aff_inv_bwd(f.T, f.Ti, dTi, dT);
  for (int c = 0; c < 12; c++) { w.dT[(size_t)v * 12 + c] = dT[c]; w.dTt[(size_t)v * 12 + c] = dTt[c]; w.dSv[(size_t)v * 12 + c] = dS[c]; }
  for (int a = 0; a < 3; a++) w.dpo[(size_t)v * 3 + a] = -dM[a * 4 + 3];
}

// sums over the vertices in a fixed order: block b < 24: d A_b; 24 <= b < 48: d A_t,(b-24); b == 48: d S; b >= 49: d pf[b - 49]
__global__ __launch_bounds__(IA_LBS_THREADS) void k_lbs_reduce(LbsBodyDev B, LbsWs w) {
  __shared__ float s_red[IA_LBS_THREADS][12];
  const int b = blockIdx.x, tid = threadIdx.x;
  float acc[12];
  for (int c = 0; c < 12; c++) acc[c] = 0.f;
  int n_out = 12;
  if (b < 48) {
    const int jn = b < 24 ? b : b - 24;
    const float *src = b < 24 ? w.dT : w.dTt;
    for (int v = tid; v < B.V; v += IA_LBS_THREADS) {
      const float wt = B.lbs_weights[(size_t)v * 24 + jn];
      for (int c = 0; c < 12; c++) acc[c] += wt * src[(size_t)v * 12 + c];
    }
  } else if (b == 48) {
    for (int v = tid; v < B.V; v += IA_LBS_THREADS)
      for (int c = 0; c < 12; c++) acc[c] += w.dSv[(size_t)v * 12 + c];
  } else {
    const int k = b - 49;
    n_out = 1;
    const float *pd = B.posedirs + (size_t)k * B.V * 3;
    for (int e = tid; e < B.V * 3; e += IA_LBS_THREADS) acc[0] += pd[e] * w.dpo[e];
  }
  for (int c = 0; c < 12; c++) s_red[tid][c] = acc[c];
  __syncthreads();
  for (int s = IA_LBS_THREADS / 2; s > 0; s >>= 1) {
    if (tid < s)
      for (int c = 0; c < n_out; c++) s_red[tid][c] += s_red[tid + s][c];
    __syncthreads();
  }
  if (tid < n_out) {
    if (b < 24) w.dA[b * 12 + tid] = s_red[0][tid];
    else if (b < 48) w.dAt[(b - 24) * 12 + tid] = s_red[0][tid];
    else if (b == 48) w.dS[tid] = s_red[0][tid];
    else w.dpf[b - 49] = s_red[0][tid];
  }
}

// One chain reversed: d A_j (rows 0..2) -> d R_j (local rotations), d J accumulated.  dRG / dg are scratch [24][9] / [24][3].
// children before parents (parents[i] < i); lanes 0..8 own one element of the 3x3 products, lanes 9..11 the translation part.
__device__ __forceinline__ void chain_backward(const float (*dA)[12], const float (*R)[9], const float (*G)[12], const float (*Jn)[3],
                                               const int *par, float (*dRG)[9], float (*dg)[3], float (*dRl)[9], float (*dJ)[3], int j) {
  if (j < 24) {
    for (int a = 0; a < 3; a++) {
      for (int b = 0; b < 3; b++) dRG[j][a * 3 + b] = dA[j][a * 4 + b] - dA[j][a * 4 + 3] * Jn[j][b];   // A.t = g - RG J (+ tau)
      dg[j][a] = dA[j][a * 4 + 3];
    }
    for (int b = 0; b < 3; b++)    // d J_j -= RG_j^T dA_j.t
      dJ[j][b] -= G[j][b] * dA[j][3] + G[j][4 + b] * dA[j][7] + G[j][8 + b] * dA[j][11];
  }
  __syncthreads();
  for (int i = 23; i >= 1; i--) {
    const int p = par[i];
    if (j < 9) {
      const int a = j / 3, b = j - 3 * a;
      dRl[i][j] = G[p][a] * dRG[i][b] + G[p][4 + a] * dRG[i][3 + b] + G[p][8 + a] * dRG[i][6 + b];          // dR_i = RG_p^T dRG_i
      float wv = dg[i][a] * (Jn[i][b] - Jn[p][b]);                                                          // dg_i rel_i^T
      for (int q = 0; q < 3; q++) wv += dRG[i][a * 3 + q] * R[i][b * 3 + q];                                // + dRG_i R_i^T
      dRG[p][j] += wv;
    } else if (j < 12) {
      const int c = j - 9;
      const float drel = G[p][c] * dg[i][0] + G[p][4 + c] * dg[i][1] + G[p][8 + c] * dg[i][2];              // d rel_i = RG_p^T dg_i
      dJ[i][c] += drel;
      dJ[p][c] -= drel;
    }
    __syncthreads();
    if (j >= 9 && j < 12) dg[p][j - 9] += dg[i][j - 9];
    __syncthreads();
  }
  if (j < 9) dRl[0][j] = dRG[0][j];
  if (j < 3) dJ[0][j] += dg[0][j];     // rel_0 = J_0
  __syncthreads();
}

__global__ void k_lbs_chain_bwd(LbsBodyDev B, const float *__restrict__ betas, const float *__restrict__ pose,
                                const float *__restrict__ transl, const float *__restrict__ pose_t, LbsWs w,
                                const float *__restrict__ d_w2s, float *__restrict__ d_betas, float *__restrict__ d_pose,
                                float *__restrict__ d_transl) {
  __shared__ float R[24][9], Rt[24][9], Jn[24][3], G[24][12], Gt[24][12];
  __shared__ float dA[24][12], dAt[24][12], dRG[24][9], dg[24][3], dRl[24][9], dRlt[24][9], dJ[24][3];
  __shared__ int par[24];
  const int j = threadIdx.x;
  Rod rod;
  if (j < 24) {
    par[j] = B.parents[j];
    for (int c = 0; c < 3; c++) { Jn[j][c] = w.J[j * 3 + c]; dJ[j][c] = 0.f; }
    rodrigues(pose + j * 3, rod);
    for (int a = 0; a < 9; a++) R[j][a] = rod.R[a];
    Rod rt;
    rodrigues(pose_t + j * 3, rt);
    for (int a = 0; a < 9; a++) Rt[j][a] = rt.R[a];
    for (int c = 0; c < 12; c++) { dA[j][c] = w.dA[j * 12 + c]; dAt[j][c] = w.dAt[j * 12 + c]; }
  }
  __syncthreads();
  chain_forward(R, Jn, par, G, j);
  chain_forward(Rt, Jn, par, Gt, j);
  if (j == 0) {
    // s2w = A_0: its share from the vertices (T^-1 . s2w) + the path through w2s = s2w^-1 (caller's d w2s: the ray frame)
    float dS[12];
    for (int c = 0; c < 12; c++) dS[c] = w.dS[c];
    if (d_w2s) {
      float dW[12], dSw[12];
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 4; b++) dW[a * 4 + b] = d_w2s[a * 4 + b];
      aff_inv_bwd(w.S, w.W, dW, dSw);
      for (int c = 0; c < 12; c++) dS[c] += dSw[c];
    }
    for (int c = 0; c < 12; c++) dA[0][c] += dS[c];
  }
  __syncthreads();
  if (j < 3 && d_transl) {   // tau enters every A_j.t (body_models.py:353-357)
    float v = 0.f;
    for (int i = 0; i < 24; i++) v += dA[i][j * 4 + 3];
    d_transl[j] = v;
  }
  chain_backward(dA, R, G, Jn, par, dRG, dg, dRl, dJ, j);
  chain_backward(dAt, Rt, Gt, Jn, par, dRG, dg, dRlt, dJ, j);   // template chain: constant rotations, only d J
  if (j < 24) {
    float dR[9];
    for (int a = 0; a < 9; a++) dR[a] = dRl[j][a] + (j >= 1 ? w.dpf[(j - 1) * 9 + a] : 0.f);   // + the pose feature's share
    float dth[3];
    rodrigues_bwd(pose + j * 3, rod, dR, dth);
    for (int a = 0; a < 3; a++) d_pose[j * 3 + a] = dth[a];
  }
  __syncthreads();
  if (j < 10 && d_betas) {   // J = J0 + JS beta
    float v = 0.f;
    for (int i = 0; i < 24; i++)
      for (int c = 0; c < 3; c++) v += B.JS[(i * 3 + c) * 10 + j] * dJ[i][c];
    d_betas[j] = v;
  }
}

static int lbs_make_body(const ia_smpl_body *b, const float *po_t, LbsBodyDev *o) {
  if (!b || !b->v_template || !b->shapedirs || !b->posedirs || !b->lbs_weights || !b->J0 || !b->JS || !b->parents || !po_t || b->n_verts < 1) return 1;
  o->v_template = b->v_template; o->shapedirs = b->shapedirs; o->posedirs = b->posedirs; o->lbs_weights = b->lbs_weights;
  o->J0 = b->J0; o->JS = b->JS; o->po_t = po_t; o->parents = b->parents; o->V = b->n_verts;
  return 0;
}

extern "C" int ia_smpl_lbs_fwd(const ia_smpl_body *body, const float *betas, const float *pose, const float *transl,
                               const float *pose_t, const float *po_t, float *T_inv, float *verts, float *verts_t, float *w2s,
                               void *ws, size_t ws_bytes, void *stream) {
  LbsBodyDev B;
  IA_CHECK_ARG(lbs_make_body(body, po_t, &B) == 0, "ia_smpl_lbs_fwd: incomplete body model");
  IA_CHECK_ARG(betas && pose && pose_t && T_inv && ws, "ia_smpl_lbs_fwd: null pointer");
  IA_CHECK_ARG(ws_bytes >= ia_smpl_lbs_workspace_bytes(B.V), "ia_smpl_lbs_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  LbsWs w = lbs_carve(ws, B.V);
  hipLaunchKernelGGL(k_lbs_chain_fwd, dim3(1), dim3(64), 0, s, B, betas, pose, transl, pose_t, w);
  IA_LAUNCH_CHECK("k_lbs_chain_fwd");
  hipLaunchKernelGGL(k_lbs_vertex_fwd, dim3(ia_div_up(B.V, IA_LBS_THREADS)), dim3(IA_LBS_THREADS), 0, s, B, betas, w, T_inv, verts, verts_t, w2s);
  IA_LAUNCH_CHECK("k_lbs_vertex_fwd");
  return IA_OK;
}

extern "C" int ia_smpl_lbs_bwd(const ia_smpl_body *body, const float *betas, const float *pose, const float *transl,
                               const float *pose_t, const float *po_t, const float *d_T_inv, const float *d_w2s, float *d_betas,
                               float *d_pose, float *d_transl, void *ws, size_t ws_bytes, void *stream) {
  LbsBodyDev B;
  IA_CHECK_ARG(lbs_make_body(body, po_t, &B) == 0, "ia_smpl_lbs_bwd: incomplete body model");
  IA_CHECK_ARG(betas && pose && pose_t && d_T_inv && d_pose && ws, "ia_smpl_lbs_bwd: null pointer");
  IA_CHECK_ARG(ws_bytes >= ia_smpl_lbs_workspace_bytes(B.V), "ia_smpl_lbs_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  LbsWs w = lbs_carve(ws, B.V);
  // (the forward quantities are recomputed from the same inputs: the workspace need not survive from the forward call)
  hipLaunchKernelGGL(k_lbs_chain_fwd, dim3(1), dim3(64), 0, s, B, betas, pose, transl, pose_t, w);
  IA_LAUNCH_CHECK("k_lbs_chain_fwd");
  hipLaunchKernelGGL(k_lbs_vertex_bwd, dim3(ia_div_up(B.V, IA_LBS_THREADS)), dim3(IA_LBS_THREADS), 0, s, B, betas, w, d_T_inv);
  IA_LAUNCH_CHECK("k_lbs_vertex_bwd");
  hipLaunchKernelGGL(k_lbs_reduce, dim3(49 + 207), dim3(IA_LBS_THREADS), 0, s, B, w);
  IA_LAUNCH_CHECK("k_lbs_reduce");
  hipLaunchKernelGGL(k_lbs_chain_bwd, dim3(1), dim3(64), 0, s, B, betas, pose, transl, pose_t, w, d_w2s, d_betas, d_pose, d_transl);
  IA_LAUNCH_CHECK("k_lbs_chain_bwd");
  return IA_OK;
}
