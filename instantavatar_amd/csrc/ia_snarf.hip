// ia_snarf.hip -- Fast-SNARF deformer kernels for gfx950 (wave64).
//
//   k_smpl_tfs      a1/a2  SMPL joint chain -> bone transforms (one wave)
//   k_precompute    a3     blended 3x4 transforms per skinning voxel, written
//                          channel-LAST so a trilinear corner is 48 contiguous B
//   k_search        a4+a5  Broyden root finding with lane refill (a workgroup owns
//                          64 points x n_init solves as an LDS queue), duplicate
//                          filter and ballot/prefix-sum compaction of the roots
//
// Reference semantics: fast_snarf/cuda/precompute/precompute.cu:24-71,
// fuse_kernel/fuse_cuda_kernel_fast.cu:23-55,62-108,110-248,252-413,
// filter/filter.cu:10-55, deformers/smplx/lbs.py:295-401.
#include "ia_common.h"

// ---------------------------------------------------------------------------
// a1/a2: rodrigues + kinematic chain + tfs = inv(A0) . A . inv(A_rest)
// One workgroup of 64 threads; lanes 0..23 own one joint each.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void mat4_mul(const float *a, const float *b, float *c) {
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      float s = 0.f;
      for (int k = 0; k < 4; k++) s += a[i * 4 + k] * b[k * 4 + j];
      c[i * 4 + j] = s;
    }
}

__global__ void k_smpl_tfs(const float *__restrict__ joints, const int32_t *__restrict__ parents,
                           const float *__restrict__ pose, const float *__restrict__ transl,
                           const float *__restrict__ tfs_inv_t, float *__restrict__ tfs,
                           float *__restrict__ w2s_out, float *__restrict__ A_out) {
  __shared__ float tm[24][16];     // local transforms, then chain
  __shared__ float chain[24][16];
  __shared__ float A[24][16];
  __shared__ float w2s[16];
  __shared__ int s_par[24];
  const int j = threadIdx.x;
  if (j < 24) s_par[j] = parents[j];
  if (j < 24) {
    // batch_rodrigues (lbs.py:295-329)
    float rx = pose[j * 3], ry = pose[j * 3 + 1], rz = pose[j * 3 + 2];
    float ax = rx + 1e-8f, ay = ry + 1e-8f, az = rz + 1e-8f;
    float angle = sqrtf(ax * ax + ay * ay + az * az);
    float dx = rx / angle, dy = ry / angle, dz = rz / angle;
    float c = cosf(angle), s = sinf(angle);
    float K[9] = {0, -dz, dy, dz, 0, -dx, -dy, dx, 0};
    float KK[9];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        float v = 0.f;
        for (int k = 0; k < 3; k++) v += K[a * 3 + k] * K[k * 3 + b];
        KK[a * 3 + b] = v;
      }
    int p = parents[j];
    float relx = joints[j * 3], rely = joints[j * 3 + 1], relz = joints[j * 3 + 2];
    if (j > 0) { relx -= joints[p * 3]; rely -= joints[p * 3 + 1]; relz -= joints[p * 3 + 2]; }
    float rel[3] = {relx, rely, relz};
    for (int a = 0; a < 3; a++) {
      for (int b = 0; b < 3; b++)
        tm[j][a * 4 + b] = (a == b ? 1.f : 0.f) + s * K[a * 3 + b] + (1.f - c) * KK[a * 3 + b];
      tm[j][a * 4 + 3] = rel[a];
    }
    tm[j][12] = 0; tm[j][13] = 0; tm[j][14] = 0; tm[j][15] = 1;
  }
  __syncthreads();
  // sequential chain (lbs.py:384-389): 23 dependent 4x4 products, lane (a, b) of the first 16 computes element (a, b) with
  // the summation order of mat4_mul (one thread doing all 16 elements of all 23 products: 24 -> 9 us for the launch)
  if (j < 16) chain[0][j] = tm[0][j];
  __syncthreads();
  for (int i = 1; i < 24; i++) {
    if (j < 16) {
      const int a = j >> 2, b = j & 3;
      const float *pa = chain[s_par[i]];   // (from LDS: a global load here is a ~1 us round trip per joint)
      float acc = 0.f;
      for (int k = 0; k < 4; k++) acc += pa[a * 4 + k] * tm[i][k * 4 + b];
      chain[i][j] = acc;
    }
    __syncthreads();
  }
  if (j < 24) {
    // rel_transforms = transforms - pad(transforms @ [J,0])  (lbs.py:396-399)
    float jx = joints[j * 3], jy = joints[j * 3 + 1], jz = joints[j * 3 + 2];
    for (int a = 0; a < 4; a++) {
      float t = chain[j][a * 4 + 0] * jx + chain[j][a * 4 + 1] * jy + chain[j][a * 4 + 2] * jz;
      for (int b = 0; b < 3; b++) A[j][a * 4 + b] = chain[j][a * 4 + b];
      A[j][a * 4 + 3] = chain[j][a * 4 + 3] - t;
    }
    // transl folded into A (body_models.py:353-357)
    if (transl) { A[j][3] += transl[0]; A[j][7] += transl[1]; A[j][11] += transl[2]; }
    if (A_out) for (int k = 0; k < 16; k++) A_out[j * 16 + k] = A[j][k];
  }
  __syncthreads();
  if (j == 0) {
    // w2s = inverse(A[0]) (snarf_deformer.py:83-84); general 4x4 inverse by
    // Gauss-Jordan with partial pivoting (torch.inverse = LU, same result to
    // rounding; A[0] is a rigid transform).
    float m[4][8];
    for (int a = 0; a < 4; a++)
      for (int b = 0; b < 4; b++) { m[a][b] = A[0][a * 4 + b]; m[a][4 + b] = (a == b) ? 1.f : 0.f; }
    for (int col = 0; col < 4; col++) {
      int piv = col;
      for (int r = col + 1; r < 4; r++) if (fabsf(m[r][col]) > fabsf(m[piv][col])) piv = r;
      if (piv != col) for (int b = 0; b < 8; b++) { float t = m[col][b]; m[col][b] = m[piv][b]; m[piv][b] = t; }
      float inv = 1.f / m[col][col];
      for (int b = 0; b < 8; b++) m[col][b] *= inv;
      for (int r = 0; r < 4; r++) if (r != col) {
        float f = m[r][col];
        for (int b = 0; b < 8; b++) m[r][b] -= f * m[col][b];
      }
    }
    for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) w2s[a * 4 + b] = m[a][4 + b];
    if (w2s_out) for (int k = 0; k < 16; k++) w2s_out[k] = w2s[k];
  }
  __syncthreads();
  if (j < 24) {  // tfs = w2s @ A @ tfs_inv_t  (snarf_deformer.py:86)
    float t1[16], t2[16];
    mat4_mul(w2s, A[j], t1);
    mat4_mul(t1, tfs_inv_t + j * 16, t2);
    for (int k = 0; k < 16; k++) tfs[j * 16 + k] = t2[k];
  }
}

// ---------------------------------------------------------------------------
// a1/a2 backward: dL/dtfs [24,4,4] (rows 0..2) -> dL/dpose [72], dL/dtransl [3].
// What autograd does for the reference when the SMPL parameters are optimised (DNeRF.py:113-128: prepare_deformer runs
// lbs.py under autograd -- ~120 tiny launches forward, ~250 backward): here the forward quantities are recomputed and
// the chain rule is written out, one launch of one wave:
//   tfs_j = W A_j B_j,  W = A_0^-1,  A_j = [RG_j | g_j - RG_j J_j + tau],  G_j = G_p(j) L_j,  L_j = [R(theta_j) | rel_j]
//   E_j = D_j B_j^T;  dA_j = R_W^T E_j;  dW = sum_j E_j A_j^T;  dA_0 -= W^T dW W^T   (rows 0..2 throughout)
//   dRG_j = dA_j[:, :3] - dA_j[:, 3] J_j^T;  dg_j = dA_j[:, 3];  dtau = sum_j dg_j
//   children before parents:  dRG_p += dRG_j R_j^T + dg_j rel_j^T;  dg_p += dg_j;  dR_j = RG_p^T dRG_j
//   Rodrigues (lbs.py:295-329; angle = |theta + 1e-8|, dir = theta / angle):  dR_j -> dtheta_j
// ---------------------------------------------------------------------------
__global__ void k_smpl_tfs_bwd(const float *__restrict__ joints, const int32_t *__restrict__ parents,
                               const float *__restrict__ pose, const float *__restrict__ transl,
                               const float *__restrict__ tfs_inv_t, const float *__restrict__ d_tfs,
                               float *__restrict__ d_pose, float *__restrict__ d_transl) {
  __shared__ float tm[24][16];      // L_j
  __shared__ float chain[24][16];   // G_j
  __shared__ float A[24][12];       // rows 0..2 of A_j
  __shared__ float W[16];
  __shared__ float E[24][12];       // rows 0..2 of D_j B_j^T
  __shared__ float dA[24][12];
  __shared__ float dRG[24][9], dg[24][3], dRl[24][9];
  __shared__ float dW[12];
  __shared__ int s_par[24];
  const int j = threadIdx.x;
  float ang = 1.f, sn = 0.f, cs = 1.f, dir[3] = {0, 0, 0}, K[9], KK[9];
  if (j < 24) {
    s_par[j] = parents[j];
    const float rx = pose[j * 3], ry = pose[j * 3 + 1], rz = pose[j * 3 + 2];
    const float ax = rx + 1e-8f, ay = ry + 1e-8f, az = rz + 1e-8f;
    ang = sqrtf(ax * ax + ay * ay + az * az);
    dir[0] = rx / ang; dir[1] = ry / ang; dir[2] = rz / ang;
    cs = cosf(ang); sn = sinf(ang);
    const float k[9] = {0, -dir[2], dir[1], dir[2], 0, -dir[0], -dir[1], dir[0], 0};
    for (int a = 0; a < 9; a++) K[a] = k[a];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        float v = 0.f;
        for (int q = 0; q < 3; q++) v += K[a * 3 + q] * K[q * 3 + b];
        KK[a * 3 + b] = v;
      }
    const int p = parents[j];
    float rel[3] = {joints[j * 3], joints[j * 3 + 1], joints[j * 3 + 2]};
    if (j > 0) { rel[0] -= joints[p * 3]; rel[1] -= joints[p * 3 + 1]; rel[2] -= joints[p * 3 + 2]; }
    for (int a = 0; a < 3; a++) {
      for (int b = 0; b < 3; b++) tm[j][a * 4 + b] = (a == b ? 1.f : 0.f) + sn * K[a * 3 + b] + (1.f - cs) * KK[a * 3 + b];
      tm[j][a * 4 + 3] = rel[a];
    }
    tm[j][12] = 0; tm[j][13] = 0; tm[j][14] = 0; tm[j][15] = 1;
  }
  __syncthreads();
  if (j < 16) chain[0][j] = tm[0][j];
  __syncthreads();
  for (int i = 1; i < 24; i++) {
    if (j < 16) {
      const int a = j >> 2, b = j & 3;
      const float *pa = chain[s_par[i]];
      float acc = 0.f;
      for (int q = 0; q < 4; q++) acc += pa[a * 4 + q] * tm[i][q * 4 + b];
      chain[i][j] = acc;
    }
    __syncthreads();
  }
  if (j < 24) {
    const float jx = joints[j * 3], jy = joints[j * 3 + 1], jz = joints[j * 3 + 2];
    for (int a = 0; a < 3; a++) {
      const float t = chain[j][a * 4 + 0] * jx + chain[j][a * 4 + 1] * jy + chain[j][a * 4 + 2] * jz;
      for (int b = 0; b < 3; b++) A[j][a * 4 + b] = chain[j][a * 4 + b];
      A[j][a * 4 + 3] = chain[j][a * 4 + 3] - t + (transl ? transl[a] : 0.f);
    }
  }
  __syncthreads();
  if (j == 0) {  // W = inverse(A_0): Gauss-Jordan with partial pivoting, as the forward kernel
    float m[4][8];
    for (int a = 0; a < 4; a++)
      for (int b = 0; b < 4; b++) { m[a][b] = a < 3 ? A[0][a * 4 + b] : (b == 3 ? 1.f : 0.f); m[a][4 + b] = (a == b) ? 1.f : 0.f; }
    for (int col = 0; col < 4; col++) {
      int piv = col;
      for (int r = col + 1; r < 4; r++) if (fabsf(m[r][col]) > fabsf(m[piv][col])) piv = r;
      if (piv != col) for (int b = 0; b < 8; b++) { float t = m[col][b]; m[col][b] = m[piv][b]; m[piv][b] = t; }
      const float inv = 1.f / m[col][col];
      for (int b = 0; b < 8; b++) m[col][b] *= inv;
      for (int r = 0; r < 4; r++) if (r != col) {
        const float f = m[r][col];
        for (int b = 0; b < 8; b++) m[r][b] -= f * m[col][b];
      }
    }
    for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) W[a * 4 + b] = m[a][4 + b];
  }
  __syncthreads();
  if (j < 24) {
    // E_j = D_j B_j^T (rows 0..2);  dA_j = R_W^T E_j
    const float *B = tfs_inv_t + j * 16, *D = d_tfs + j * 16;
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 4; b++) {
        float v = 0.f;
        for (int q = 0; q < 4; q++) v += D[a * 4 + q] * B[b * 4 + q];
        E[j][a * 4 + b] = v;
      }
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 4; b++) {
        float v = 0.f;
        for (int q = 0; q < 3; q++) v += W[q * 4 + a] * E[j][q * 4 + b];
        dA[j][a * 4 + b] = v;
      }
  }
  __syncthreads();
  if (j < 12) {  // dW = sum_j E_j A_j^T (rows 0..2; A_j's fourth row is (0,0,0,1)), joints in a fixed order
    const int a = j >> 2, b = j & 3;
    float v = 0.f;
    for (int i = 0; i < 24; i++) {
      float t = 0.f;
      for (int q = 0; q < 4; q++) t += E[i][a * 4 + q] * (b < 3 ? A[i][b * 4 + q] : (q == 3 ? 1.f : 0.f));
      v += t;
    }
    dW[j] = v;
  }
  __syncthreads();
  if (j < 12) {  // dA_0 -= (W^T dW W^T) rows 0..2, dW's fourth row = 0
    const int a = j >> 2, b = j & 3;
    float v = 0.f;
    for (int q = 0; q < 3; q++)       // (W^T dW)[a][r] = sum_q W[q][a] dW[q][r]
      for (int r = 0; r < 4; r++) v += W[q * 4 + a] * dW[q * 4 + r] * W[b * 4 + r];   // ... * (W^T)[r][b] = W[b][r]
    dA[0][a * 4 + b] -= v;
  }
  __syncthreads();
  if (j < 24) {
    const float jx[3] = {joints[j * 3], joints[j * 3 + 1], joints[j * 3 + 2]};
    for (int a = 0; a < 3; a++) {
      for (int b = 0; b < 3; b++) dRG[j][a * 3 + b] = dA[j][a * 4 + b] - dA[j][a * 4 + 3] * jx[b];
      dg[j][a] = dA[j][a * 4 + 3];
    }
  }
  __syncthreads();
  if (j < 3 && d_transl) {   // d tau = sum_j dg_j, joints in a fixed order
    float v = 0.f;
    for (int i = 0; i < 24; i++) v += dg[i][j];
    d_transl[j] = v;
  }
  __syncthreads();
  // children before parents (parents[i] < i in the SMPL tree): lane (a, b) of the first nine owns one element of the three
  // 3x3 products of a step, lanes 9..11 the translation gradient (one lane walking all 23 joints alone took 35 of the
  // kernel's 44 us).  Every sum keeps its order: dRG_i R_i^T first, then + dg_i rel_i^T, accumulated onto the parent.
  for (int i = 23; i >= 1; i--) {
    const int p = s_par[i];
    if (j < 9) {
      const int a = j / 3, b = j - 3 * a;
      float v = 0.f;
      for (int q = 0; q < 3; q++) v += chain[p][q * 4 + a] * dRG[i][q * 3 + b];   // dR_i = RG_p^T dRG_i
      dRl[i][j] = v;
      float w = dg[i][a] * tm[i][b * 4 + 3];                                       // dg_i rel_i^T
      for (int q = 0; q < 3; q++) w += dRG[i][a * 3 + q] * tm[i][b * 4 + q];        // + dRG_i R_i^T
      dRG[p][j] += w;
    } else if (j < 12) {
      dg[p][j - 9] += dg[i][j - 9];
    }
    __syncthreads();
  }
  if (j < 9) dRl[0][j] = dRG[0][j];
  __syncthreads();
  if (j < 24) {
    // Rodrigues backward: R = I + sin(a) K + (1 - cos a) K^2
    const float *dR = dRl[j];
    float dK[9];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        float v = sn * dR[a * 3 + b];
        for (int q = 0; q < 3; q++) v += (1.f - cs) * (dR[a * 3 + q] * K[b * 3 + q] + K[q * 3 + a] * dR[q * 3 + b]);   // dR K^T + K^T dR
        dK[a * 3 + b] = v;
      }
    float dRK = 0.f, dRKK = 0.f;
    for (int a = 0; a < 9; a++) { dRK += dR[a] * K[a]; dRKK += dR[a] * KK[a]; }
    const float d_ang = cs * dRK + sn * dRKK;
    const float d_dir[3] = {dK[7] - dK[5], dK[2] - dK[6], dK[3] - dK[1]};
    const float th[3] = {pose[j * 3], pose[j * 3 + 1], pose[j * 3 + 2]};
    const float dot = d_dir[0] * th[0] + d_dir[1] * th[1] + d_dir[2] * th[2];
    const float coef = d_ang - dot / (ang * ang);
    for (int a = 0; a < 3; a++) d_pose[j * 3 + a] = d_dir[a] / ang + coef * (th[a] + 1e-8f) / ang;
  }
}

// ---------------------------------------------------------------------------
// a3: precompute.  One thread per voxel.  voxel_w is read channel-major
// (coalesced per joint plane); tfs is wave-uniform (scalar loads).
// ---------------------------------------------------------------------------
__global__ void k_bbox_init(float *bbox) {
  if (threadIdx.x < 3) bbox[threadIdx.x] = __int_as_float(0x7f800000);
  else if (threadIdx.x < 6) bbox[threadIdx.x] = __int_as_float(0xff800000);
}

#ifndef IA_PRE_LDS_STORE
#define IA_PRE_LDS_STORE 1  // stage the channel-last transform records through LDS so that every store instruction is contiguous:
                            // 25.8 -> 18.9 us for the product call (d + bbox, incl. the reduce launch), streams alone 19.7 -> 13.2 us
                            // (profiles/r06_ab_precompute.txt)
#endif
#ifndef IA_PRE_VPT
#define IA_PRE_VPT 4  // consecutive voxels (along W) per thread: 1, 2 or 4 (measured 173 / 185 / 95 us)
#endif
template <int VPT> struct PreVec;
template <> struct PreVec<1> { typedef float type; };
template <> struct PreVec<2> { typedef float2 type; };
template <> struct PreVec<4> { typedef float4 type; };

template <int VPT, bool LDS_STORE>
__global__ __launch_bounds__(256) void k_precompute(const float *__restrict__ voxel_w,
                                                    const float *__restrict__ tfs,
                                                    float *__restrict__ voxel_J,
                                                    float *__restrict__ voxel_d,
                                                    float *__restrict__ bbox, float *__restrict__ partial,
                                                    SnarfGridDev g) {
  // VPT consecutive voxels (along W) per thread: one (4*VPT)-byte load per joint plane and 48*VPT
  // contiguous bytes of output per thread.  W % VPT == 0 is checked by the host.
  typedef typename PreVec<VPT>::type vec_t;
  const int n = g.D * g.H * g.W;
  __shared__ float4 s_stage[LDS_STORE ? 4 * 64 * (3 * VPT + 1) : 1];
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n / VPT; q += gridDim.x * blockDim.x) {
    const int index0 = q * VPT;
    float J[VPT][12];
#pragma unroll
    for (int v = 0; v < VPT; v++)
#pragma unroll
      for (int c = 0; c < 12; c++) J[v][c] = 0.f;
    // precompute.cu:51-59: J[c] accumulates over j in joint order
    // (NOTE: every variant below was timed while the bounding-box reduction at the end of the kernel still sent
    // 2 048 waves x 6 float atomics to six addresses -- 74 of the 93 us, see the end of the kernel; they compare the
    // variants under that tail, not the streams themselves.  With the reduction through per-workgroup extrema the
    // kernel takes 19-22 us for 75-82 MB, 3.7-3.9 TB/s.)
    // (requesting all 24 planes before the first use was measured: 193 VGPRs, 98 -> 115 us; one voxel per thread with
    // all 24 four-byte loads in flight and LDS-transposed, fully coalesced stores: 180 us -- 4-byte-per-lane plane
    // loads stream at half the rate of 16-byte ones; groups of 4 / 6 / 8 / 12 planes explicitly in flight (the
    // compiler otherwise waits after every load): 122 / 115 / 119 / 112 us against 93 us -- more requests in flight
    // make it slower; channel-LAST weights (96 contiguous bytes per voxel, six 16-byte loads per lane): 170 us, the
    // strided lanes are served one at a time; padding the 2 MiB plane stride: no effect.  Counters: TCP pending-stall
    // 88 % of the launch, 2.1 M L2 requests of which 1.3 M are the 16-byte pieces of the channel-last stores.)
#ifndef IA_PRE_UNROLL
#define IA_PRE_UNROLL 4  // joint planes in flight per thread: 2 / 4 / 6 / 8 measured 98.7 / 92.8 / 96.6 / 105.1 us
#endif
#pragma unroll IA_PRE_UNROLL
    for (int j = 0; j < 24; j++) {
      union { vec_t v; float f[VPT]; } w;
      w.v = *reinterpret_cast<const vec_t *>(voxel_w + (size_t)j * n + index0);
#pragma unroll
      for (int c = 0; c < 12; c++) {
        const float t = tfs[j * 16 + c];
#pragma unroll
        for (int v = 0; v < VPT; v++) J[v][c] = __builtin_fmaf(w.f[v], t, J[v][c]);
      }
    }
    if (LDS_STORE) {
    // The wave's VPT x 64 records are ONE contiguous run of 64 * VPT * 48 bytes, but lane t holds bytes [192 t, 192 t + 192) of
    // it: a direct store instruction writes 64 sixteen-byte pieces 192 bytes apart (1.3 M of the kernel's 2.1 M L2 requests).
    // Through LDS (row stride 3 VPT + 1 float4: conflict-free for both phases) every store instruction writes 1 024 contiguous
    // bytes.  Wave-private staging: no workgroup barrier (all lanes of a wave take the same trip count when n / VPT % 64 == 0,
    // which the host checks before it picks this variant).
    {
      constexpr int RS = 3 * VPT + 1;
      float4 *const sw = s_stage + (threadIdx.x >> 6) * (64 * RS);
      const int lane = threadIdx.x & 63;
#pragma unroll
      for (int v = 0; v < VPT; v++) {
        sw[lane * RS + 3 * v + 0] = make_float4(J[v][0], J[v][1], J[v][2], J[v][3]);
        sw[lane * RS + 3 * v + 1] = make_float4(J[v][4], J[v][5], J[v][6], J[v][7]);
        sw[lane * RS + 3 * v + 2] = make_float4(J[v][8], J[v][9], J[v][10], J[v][11]);
      }
      __builtin_amdgcn_wave_barrier();
      float4 *const o = reinterpret_cast<float4 *>(voxel_J + (size_t)(index0 - lane * VPT) * 12);   // the wave's first record
#pragma unroll
      for (int k = 0; k < 3 * VPT; k++) {
        const int e = k * 64 + lane;              // float4 index inside the wave's run
        o[e] = sw[(e / (3 * VPT)) * RS + e % (3 * VPT)];
      }
      __builtin_amdgcn_wave_barrier();
    }
    } else {
    float4 *o = reinterpret_cast<float4 *>(voxel_J + (size_t)index0 * 12);
#pragma unroll
    for (int v = 0; v < VPT; v++) {
      o[3 * v + 0] = make_float4(J[v][0], J[v][1], J[v][2], J[v][3]);
      o[3 * v + 1] = make_float4(J[v][4], J[v][5], J[v][6], J[v][7]);
      o[3 * v + 2] = make_float4(J[v][8], J[v][9], J[v][10], J[v][11]);
    }
    }
    const int hw = g.H * g.W;
    const int idx_d = index0 / hw, idx_h = index0 % hw / g.W, idx_w0 = index0 % hw % g.W;
    // precompute.cu:42-47
    const float cy = (((float)idx_h) / (g.H - 1) * 2 - 1) / g.scl[1] - g.off[1];
    const float cz = (((float)idx_d) / (g.D - 1) * 2 - 1) / g.scl[2] - g.off[2];
    float xi[3][VPT];
#pragma unroll
    for (int v = 0; v < VPT; v++) {
      const float cx = (((float)(idx_w0 + v)) / (g.W - 1) * 2 - 1) / g.scl[0] - g.off[0];
      // precompute.cu:66-70
#pragma unroll
      for (int i0 = 0; i0 < 3; i0++) {
        xi[i0][v] = IA_DOT3(J[v][i0 * 4 + 0], cx, J[v][i0 * 4 + 1], cy, J[v][i0 * 4 + 2], cz) + J[v][i0 * 4 + 3];
        mn[i0] = fminf(mn[i0], xi[i0][v]);
        mx[i0] = fmaxf(mx[i0], xi[i0][v]);
      }
    }
    if (voxel_d) {
#pragma unroll
      for (int i0 = 0; i0 < 3; i0++)
#pragma unroll
        for (int v = 0; v < VPT; v++) voxel_d[(size_t)i0 * n + index0 + v] = xi[i0][v];
    }
  }
  // Bounding box of the deformed voxel centres.  With a `partial` buffer every workgroup writes its six extrema
  // (k_bbox_reduce folds them): no two workgroups touch the same address.  The atomic route (partial == nullptr, kept
  // for callers without a workspace) sends 2 048 waves x 6 agent-scope loads + atomics to SIX addresses, which
  // serialise in one memory channel: measured r02 93 us for the kernel with it against 19 us without.
  if (partial) {
    __shared__ float s_red[4][6];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float a = ia_wave_min(mn[c]), b = ia_wave_max(mx[c]);
      if (ia_lane() == 0) { s_red[threadIdx.x >> 6][c] = a; s_red[threadIdx.x >> 6][3 + c] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
      float r = s_red[0][threadIdx.x];
      for (int w = 1; w < 4; w++) r = threadIdx.x < 3 ? fminf(r, s_red[w][threadIdx.x]) : fmaxf(r, s_red[w][threadIdx.x]);
      partial[(size_t)blockIdx.x * 6 + threadIdx.x] = r;
    }
  } else if (bbox) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float a = ia_wave_min(mn[c]), b = ia_wave_max(mx[c]);
      if (ia_lane() == 0) { ia_atomic_min_f(bbox + c, a); ia_atomic_max_f(bbox + 3 + c, b); }
    }
  }
}

// folds the per-workgroup extrema of k_precompute: one workgroup, component c = threadIdx.x % 6
__global__ __launch_bounds__(384) void k_bbox_reduce(const float *__restrict__ partial, int n_blocks, float *__restrict__ bbox) {
  __shared__ float s_red[64][6];
  const int c = threadIdx.x % 6, r = threadIdx.x / 6;   // 64 rows x 6 components
  const bool is_min = c < 3;
  float v = is_min ? INFINITY : -INFINITY;
  for (int b = r; b < n_blocks; b += 64) {
    const float p = partial[(size_t)b * 6 + c];
    v = is_min ? fminf(v, p) : fmaxf(v, p);
  }
  s_red[r][c] = v;
  __syncthreads();
  if (threadIdx.x < 6) {
    float o = s_red[0][threadIdx.x];
    for (int k = 1; k < 64; k++) o = threadIdx.x < 3 ? fminf(o, s_red[k][threadIdx.x]) : fmaxf(o, s_red[k][threadIdx.x]);
    bbox[threadIdx.x] = o;
  }
}


// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" int ia_smpl_tfs(const float *joints_rest, const int32_t *parents, const float *pose,
                           const float *transl, const float *tfs_inv_t, float *tfs, float *w2s, float *A,
                           void *stream) {
  IA_CHECK_ARG(joints_rest && parents && pose && tfs_inv_t && tfs, "ia_smpl_tfs: null pointer");
  hipLaunchKernelGGL(k_smpl_tfs, dim3(1), dim3(64), 0, (hipStream_t)stream, joints_rest, parents, pose,
                     transl, tfs_inv_t, tfs, w2s, A);
  IA_LAUNCH_CHECK("k_smpl_tfs");
  return IA_OK;
}

extern "C" int ia_smpl_tfs_bwd(const float *joints_rest, const int32_t *parents, const float *pose, const float *transl,
                               const float *tfs_inv_t, const float *d_tfs, float *d_pose, float *d_transl, void *stream) {
  IA_CHECK_ARG(joints_rest && parents && pose && tfs_inv_t && d_tfs && d_pose, "ia_smpl_tfs_bwd: null pointer");
  hipLaunchKernelGGL(k_smpl_tfs_bwd, dim3(1), dim3(64), 0, (hipStream_t)stream, joints_rest, parents, pose, transl, tfs_inv_t,
                     d_tfs, d_pose, d_transl);
  IA_LAUNCH_CHECK("k_smpl_tfs_bwd");
  return IA_OK;
}

static int ia_precompute_blocks(const ia_snarf_grid *grid) {
  const long nt = (long)grid->D * grid->H * grid->W / IA_PRE_VPT;
  return (int)((nt + 255) / 256 < 8192 ? (nt + 255) / 256 : 8192);
}

extern "C" size_t ia_precompute_workspace_bytes(const ia_snarf_grid *grid) {
  if (!grid || grid->D < 1 || grid->H < 1 || grid->W < 1) return 0;
  return (size_t)ia_precompute_blocks(grid) * 6 * sizeof(float);
}

extern "C" int ia_precompute_ws(const float *voxel_w, const float *tfs, float *voxel_J, float *voxel_d,
                                float *bbox, const ia_snarf_grid *grid, void *ws, size_t ws_bytes, void *stream) {
  IA_CHECK_ARG(voxel_w && tfs && voxel_J && grid, "ia_precompute: null pointer");
  IA_CHECK_ARG(grid->D > 1 && grid->H > 1 && grid->W > 1 && grid->W % 4 == 0, "ia_precompute: bad grid %d %d %d (W must be a multiple of 4)", grid->D, grid->H, grid->W);
  hipStream_t s = (hipStream_t)stream;
  const int blocks = ia_precompute_blocks(grid);
  float *partial = nullptr;
  if (bbox && ws) {
    IA_CHECK_ARG(ws_bytes >= ia_precompute_workspace_bytes(grid), "ia_precompute_ws: workspace of %zu bytes, %zu needed", ws_bytes,
                 ia_precompute_workspace_bytes(grid));
    partial = (float *)ws;
  } else if (bbox) {
    hipLaunchKernelGGL(k_bbox_init, dim3(1), dim3(64), 0, s, bbox);
    IA_LAUNCH_CHECK("k_bbox_init");
  }
  // (the LDS-staged stores need whole waves: every lane of a wave takes the same number of trips)
  const long n_thr = (long)grid->D * grid->H * grid->W / IA_PRE_VPT;
  if (IA_PRE_LDS_STORE && n_thr % 64 == 0 && n_thr % ((long)blocks * 256) == 0)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_precompute<IA_PRE_VPT, true>), dim3(blocks), dim3(256), 0, s, voxel_w, tfs, voxel_J, voxel_d, bbox,
                       partial, ia_make_grid_dev(grid));
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_precompute<IA_PRE_VPT, false>), dim3(blocks), dim3(256), 0, s, voxel_w, tfs, voxel_J, voxel_d, bbox,
                       partial, ia_make_grid_dev(grid));
  IA_LAUNCH_CHECK("k_precompute");
  if (partial) {
    hipLaunchKernelGGL(k_bbox_reduce, dim3(1), dim3(384), 0, s, partial, blocks, bbox);
    IA_LAUNCH_CHECK("k_bbox_reduce");
  }
  return IA_OK;
}

extern "C" int ia_precompute(const float *voxel_w, const float *tfs, float *voxel_J, float *voxel_d,
                             float *bbox, const ia_snarf_grid *grid, void *stream) {
  return ia_precompute_ws(voxel_w, tfs, voxel_J, voxel_d, bbox, grid, nullptr, 0, stream);
}


// ---------------------------------------------------------------------------
// a7: implicit differentiation of the roots w.r.t. the bone transforms
// (deformer_torch.py:50-67, forward_skinning :118-128, query_weights :190-202).
//   x_c <- x_c* - J_inv (d(x_c*) - sg[d(x_c*)]),   d(x) = sum_n w_n(x) (R_n x + t_n)
// has the value x_c* and the gradient  dL/dT_n[c][k] = w_n(x_c*) v_c h_k  with
// v = -J_inv^T dL/dx_c and h = (x_c*, 1).  The reference builds it from a 24-channel grid_sample
// (align_corners, border padding), an einsum, two batched mat-vecs and their autograd; here one
// kernel gathers the 8 x 24 skinning weights of every valid candidate, forms the 24 x 12 outer
// products in LDS and reduces them per workgroup (per-workgroup partials, added in a fixed order).
// ---------------------------------------------------------------------------
#define IA_ID_THREADS 256

__device__ __forceinline__ float id_border_index(float g, int size) {
  float c = ((g + 1.f) / 2) * (size - 1);          // align_corners = true
  c = fminf(fmaxf(c, 0.f), (float)(size - 1));      // padding_mode = "border"
  return c;
}

// CL: voxel_w is channel-LAST [D,H,W,24] (96 contiguous bytes per voxel: six 16-byte loads per corner instead of 24 four-byte
// loads 2 MB apart; the refine step's launch went from 176 us to the figure in DESIGN.md with it)
// query_weights (deformer_torch.py:190-202): trilinear sample of the 24 skinning weights at x, align_corners, border padding
template <bool CL>
__device__ __forceinline__ void id_sample_weights(const float *__restrict__ voxel_w, const SnarfGridDev &g, float x0, float x1, float x2,
                                                  float *__restrict__ w) {
  const long vol = (long)g.D * g.H * g.W;
  const float ix = id_border_index(g.scl[0] * (x0 + g.off[0]), g.W);
  const float iy = id_border_index(g.scl[1] * (x1 + g.off[1]), g.H);
  const float iz = id_border_index(g.scl[2] * (x2 + g.off[2]), g.D);
  const int xa = (int)floorf(ix), ya = (int)floorf(iy), za = (int)floorf(iz);
  const float fx = ix - xa, fy = iy - ya, fz = iz - za;
#pragma unroll
  for (int cz = 0; cz < 2; cz++)
#pragma unroll
    for (int cy = 0; cy < 2; cy++)
#pragma unroll
      for (int cx = 0; cx < 2; cx++) {
        const int xx = xa + cx, yy = ya + cy, zz = za + cz;
        if (xx >= g.W || yy >= g.H || zz >= g.D) continue;  // only at the clamped border, weight 0
        const float wt = (cx ? fx : 1.f - fx) * (cy ? fy : 1.f - fy) * (cz ? fz : 1.f - fz);
        if (CL) {
          const float4 *p4 = reinterpret_cast<const float4 *>(voxel_w + (((long)zz * g.H + yy) * g.W + xx) * 24);
#pragma unroll
          for (int q = 0; q < 6; q++) {
            const float4 v = p4[q];
            w[4 * q] = __builtin_fmaf(wt, v.x, w[4 * q]); w[4 * q + 1] = __builtin_fmaf(wt, v.y, w[4 * q + 1]);
            w[4 * q + 2] = __builtin_fmaf(wt, v.z, w[4 * q + 2]); w[4 * q + 3] = __builtin_fmaf(wt, v.w, w[4 * q + 3]);
          }
        } else {
          const float *p = voxel_w + ((long)zz * g.H + yy) * g.W + xx;
#pragma unroll
          for (int k = 0; k < 24; k++) w[k] = __builtin_fmaf(wt, p[(long)k * vol], w[k]);
        }
      }
}

// the blended transform T = sum_n w_n tfs_n (rows 0..2) of an entry's skinning weights; s_tfs: [24][12] in LDS
__device__ __forceinline__ void id_blend_transform(const float *__restrict__ w, const float (*__restrict__ s_tfs)[12], float *__restrict__ T) {
#pragma unroll
  for (int q = 0; q < 12; q++) T[q] = 0.f;
  for (int nb = 0; nb < 24; nb++)
#pragma unroll
    for (int q = 0; q < 12; q++) T[q] = __builtin_fmaf(w[nb], s_tfs[nb][q], T[q]);
}

// MODE 0: version 1, the implicit differentiation above (inputs J_inv, grad).
// MODE 1: version 2 (deformer_torch.py:68-75), x_c = R^T (x_d - t) with T = sum_n w_n(x_c*) tfs_n:
//         dL/dT[i][j] = (x_d - t)_i g_j (j < 3), dL/dT[i][3] = -sum_j R[i][j] g_j; the entry's target x_d is xd[pt] with
//         pt = cand_pt[e] (compact candidate lists) or e / n_init (the dense [P, n_init] layout).
template <bool CL, int MODE>
__global__ __launch_bounds__(IA_ID_THREADS) void k_implicit_bwd(
    const float *__restrict__ xc, const float *__restrict__ J_inv, const uint8_t *__restrict__ valid,
    const float *__restrict__ grad, long n, const int32_t *__restrict__ n_dev, const float *__restrict__ voxel_w,
    SnarfGridDev g, float *__restrict__ partial, const float *__restrict__ xd, const int32_t *__restrict__ cand_pt, int n_init,
    const float *__restrict__ tfs, float *__restrict__ d_xd) {
  const long n_all = n;
  if (n_dev) n = min(n, (long)*n_dev);   // compact candidate lists: device-side live count, no validity mask
  __shared__ float s_w[IA_ID_THREADS][25];   // +1: the 24-float rows start in different banks
  __shared__ float s_vh[IA_ID_THREADS][13];
  __shared__ float s_tfs[24][12];
  const int tid = threadIdx.x;
  if (MODE == 1) {
    for (int e = tid; e < 24 * 12; e += IA_ID_THREADS) s_tfs[e / 12][e % 12] = tfs[(e / 12) * 16 + e % 12];
    __syncthreads();
  }
  float acc0 = 0.f, acc1 = 0.f;              // outputs tid and tid + 256 of the 288 (bone, row, col) sums
  const long n_loop = (MODE == 1 && d_xd) ? n_all : n;   // (entries past the live count still get their zero in d_xd)
  for (long base = (long)blockIdx.x * IA_ID_THREADS; base < n_loop; base += (long)gridDim.x * IA_ID_THREADS) {
    const long i = base + tid;
    const bool ok = i < n && (!valid || valid[i]);
    float w[24], vh[12];
#pragma unroll
    for (int k = 0; k < 24; k++) w[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 12; k++) vh[k] = 0.f;
    if (ok) {
      const float x0 = xc[i * 3], x1 = xc[i * 3 + 1], x2 = xc[i * 3 + 2];
      const float g0 = grad[i * 3], g1 = grad[i * 3 + 1], g2 = grad[i * 3 + 2];
      id_sample_weights<CL>(voxel_w, g, x0, x1, x2, w);
      if (MODE == 0) {
        const float *Ji = J_inv + i * 9;
        const float v[3] = {-(Ji[0] * g0 + Ji[3] * g1 + Ji[6] * g2), -(Ji[1] * g0 + Ji[4] * g1 + Ji[7] * g2),
                            -(Ji[2] * g0 + Ji[5] * g1 + Ji[8] * g2)};
        const float h[4] = {x0, x1, x2, 1.f};
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
          for (int k = 0; k < 4; k++) vh[c * 4 + k] = v[c] * h[k];
      } else {
        float T[12];
        id_blend_transform(w, s_tfs, T);
        const long pt = cand_pt ? (long)cand_pt[i] : i / n_init;
        const float gg[3] = {g0, g1, g2};
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const float a = xd[pt * 3 + c] - T[c * 4 + 3];
#pragma unroll
          for (int k = 0; k < 3; k++) vh[c * 4 + k] = a * gg[k];
          vh[c * 4 + 3] = -(T[c * 4] * g0 + T[c * 4 + 1] * g1 + T[c * 4 + 2] * g2);
        }
      }
    }
    if (MODE == 1 && d_xd && i < n_all) {   // dL/dx_d of the entry = R g (x_d reaches the SMPL parameters through the ray frame w2s)
      d_xd[i * 3] = -vh[3]; d_xd[i * 3 + 1] = -vh[7]; d_xd[i * 3 + 2] = -vh[11];
    }
    __syncthreads();  // previous tile consumed
#pragma unroll
    for (int k = 0; k < 24; k++) s_w[tid][k] = w[k];
#pragma unroll
    for (int k = 0; k < 12; k++) s_vh[tid][k] = vh[k];
    __syncthreads();
    {
      const int o0 = tid, o1 = tid + IA_ID_THREADS;  // output o = bone * 12 + (row * 4 + col)
      const int n0 = o0 / 12, q0 = o0 - n0 * 12, n1 = o1 / 12, q1 = o1 - n1 * 12;
      float a0 = 0.f, a1 = 0.f;
      for (int c = 0; c < IA_ID_THREADS; c++) {
        a0 = __builtin_fmaf(s_w[c][n0], s_vh[c][q0], a0);
        if (o1 < 288) a1 = __builtin_fmaf(s_w[c][n1], s_vh[c][q1], a1);
      }
      acc0 += a0; acc1 += a1;
    }
  }
  partial[(size_t)blockIdx.x * 288 + tid] = acc0;
  if (tid + IA_ID_THREADS < 288) partial[(size_t)blockIdx.x * 288 + tid + IA_ID_THREADS] = acc1;
}

// version 2, forward: out[e] = R^T (x_d - t) for the live / valid entries, 0 elsewhere (the reference scatters into zeros)
template <bool CL>
__global__ __launch_bounds__(IA_ID_THREADS) void k_inverse_skinning(
    const float *__restrict__ xc, const float *__restrict__ xd, const int32_t *__restrict__ cand_pt, int n_init,
    const uint8_t *__restrict__ valid, long n, const int32_t *__restrict__ n_dev, const float *__restrict__ voxel_w, SnarfGridDev g,
    const float *__restrict__ tfs, float *__restrict__ out) {
  __shared__ float s_tfs[24][12];
  for (int e = threadIdx.x; e < 24 * 12; e += IA_ID_THREADS) s_tfs[e / 12][e % 12] = tfs[(e / 12) * 16 + e % 12];
  __syncthreads();
  const long live = n_dev ? min(n, (long)*n_dev) : n;
  for (long i = (long)blockIdx.x * IA_ID_THREADS + threadIdx.x; i < n; i += (long)gridDim.x * IA_ID_THREADS) {
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
    if (i < live && (!valid || valid[i])) {
      float w[24], T[12];
#pragma unroll
      for (int k = 0; k < 24; k++) w[k] = 0.f;
      id_sample_weights<CL>(voxel_w, g, xc[i * 3], xc[i * 3 + 1], xc[i * 3 + 2], w);
      id_blend_transform(w, s_tfs, T);
      const long pt = cand_pt ? (long)cand_pt[i] : i / n_init;
      const float a0 = xd[pt * 3] - T[3], a1 = xd[pt * 3 + 1] - T[7], a2 = xd[pt * 3 + 2] - T[11];
      o0 = a0 * T[0] + a1 * T[4] + a2 * T[8];
      o1 = a0 * T[1] + a1 * T[5] + a2 * T[9];
      o2 = a0 * T[2] + a1 * T[6] + a2 * T[10];
    }
    out[i * 3] = o0; out[i * 3 + 1] = o1; out[i * 3 + 2] = o2;
  }
}

// candidate -> sample point: cand_pt[pt_off[p] + j] = p for j < pt_cnt[p] (the compaction of k_search keeps a point's
// candidates contiguous); entries past the live candidate count are never read
__global__ __launch_bounds__(256) void k_expand_candidate_points(const int32_t *__restrict__ pt_off, const uint8_t *__restrict__ pt_cnt, int P,
                                                                 const int32_t *__restrict__ n_pts_dev, int32_t *__restrict__ cand_pt, int cap) {
  if (n_pts_dev) P = min(P, *n_pts_dev);
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int o = pt_off[p], c = pt_cnt[p];
  for (int j = 0; j < c; j++)
    if (o + j < cap) cand_pt[o + j] = p;
}

// one wave per output o = bone * 12 + row * 4 + col: lanes sum the per-workgroup partials b = lane, lane + 64, ... and the
// wave folds them in a fixed order (a single 288-thread workgroup walking up to 1 024 partials one dependent load after the
// other took 225 us in the refine step)
__global__ __launch_bounds__(64) void k_implicit_bwd_reduce(const float *__restrict__ partial, int n_blocks,
                                                            float *__restrict__ d_tfs) {
  const int o = blockIdx.x;
  float acc = 0.f;
  for (int b = threadIdx.x; b < n_blocks; b += 64) acc += partial[(size_t)b * 288 + o];
#pragma unroll
  for (int k = 32; k > 0; k >>= 1) acc += __shfl_xor(acc, k, 64);
  if (threadIdx.x == 0) {
    const int bone = o / 12, q = o - bone * 12;
    d_tfs[bone * 16 + q] += acc;  // rows 0..2 of the 4x4; row 3 has no gradient
  }
}

static int ia_implicit_blocks(long n) {
  long b = (n + IA_ID_THREADS - 1) / IA_ID_THREADS;
  if (b > 1024) b = 1024;
  return (int)(b < 1 ? 1 : b);
}

extern "C" size_t ia_snarf_implicit_bwd_workspace_bytes(long n) { return (size_t)ia_implicit_blocks(n) * 288 * sizeof(float); }

struct InvSkinArgs { const float *xd; const int32_t *cand_pt; int n_init; const float *tfs; float *d_xd; };

static int ia_implicit_bwd_impl(const char *who, const float *xc, const float *J_inv, const uint8_t *valid,
                                const float *grad_xc, long n, const int32_t *n_dev, const float *voxel_w,
                                const ia_snarf_grid *grid, float *d_tfs, void *ws, size_t ws_bytes, hipStream_t s,
                                bool channel_last = false, const InvSkinArgs *v2 = nullptr) {
  IA_CHECK_ARG(n >= 0, "%s: n < 0", who);
  if (n == 0) return IA_OK;
  IA_CHECK_ARG(xc && (J_inv || v2) && (valid || n_dev) && grad_xc && voxel_w && grid && d_tfs && ws, "%s: null pointer", who);
  IA_CHECK_ARG(!v2 || (v2->xd && v2->tfs && (v2->cand_pt || v2->n_init > 0)), "%s: version-2 arguments (xd, tfs, cand_pt or n_init)", who);
  if (ws_bytes < ia_snarf_implicit_bwd_workspace_bytes(n)) return ia_set_error(IA_ERR_WORKSPACE, "%s: workspace too small", who);
  const int blocks = ia_implicit_blocks(n);
  const SnarfGridDev g = ia_make_grid_dev(grid);
  float *part = static_cast<float *>(ws);
  const float *xd = v2 ? v2->xd : nullptr, *tfs = v2 ? v2->tfs : nullptr;
  const int32_t *cpt = v2 ? v2->cand_pt : nullptr;
  const int ni = v2 ? v2->n_init : 1;
  float *dxd = v2 ? v2->d_xd : nullptr;
#define IA_ID_LAUNCH(CL, MODE) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_implicit_bwd<CL, MODE>), dim3(blocks), dim3(IA_ID_THREADS), 0, s, xc, J_inv, \
                                                  valid, grad_xc, n, n_dev, voxel_w, g, part, xd, cpt, ni, tfs, dxd)
  if (v2) { if (channel_last) IA_ID_LAUNCH(true, 1); else IA_ID_LAUNCH(false, 1); }
  else    { if (channel_last) IA_ID_LAUNCH(true, 0); else IA_ID_LAUNCH(false, 0); }
#undef IA_ID_LAUNCH
  hipLaunchKernelGGL(k_implicit_bwd_reduce, dim3(288), dim3(64), 0, s, static_cast<const float *>(ws), blocks, d_tfs);
  IA_LAUNCH_CHECK("k_implicit_bwd");
  return IA_OK;
}

extern "C" int ia_snarf_implicit_bwd(const float *xc, const float *J_inv, const uint8_t *valid, const float *grad_xc,
                                     long n, const float *voxel_w, const ia_snarf_grid *grid, float *d_tfs, void *ws,
                                     size_t ws_bytes, void *stream) {
  return ia_implicit_bwd_impl("ia_snarf_implicit_bwd", xc, J_inv, valid, grad_xc, n, nullptr, voxel_w, grid, d_tfs, ws,
                              ws_bytes, (hipStream_t)stream);
}

extern "C" int ia_snarf_implicit_bwd_compact(const float *cand_xc, const float *cand_Jinv, const float *grad_xc, long cap,
                                             const int32_t *n_cand, const float *voxel_w, int channel_last,
                                             const ia_snarf_grid *grid, float *d_tfs, void *ws, size_t ws_bytes, void *stream) {
  IA_CHECK_ARG(n_cand, "ia_snarf_implicit_bwd_compact: n_cand is null");
  return ia_implicit_bwd_impl("ia_snarf_implicit_bwd_compact", cand_xc, cand_Jinv, nullptr, grad_xc, cap, n_cand, voxel_w,
                              grid, d_tfs, ws, ws_bytes, (hipStream_t)stream, channel_last != 0);
}

// ---- ForwardDeformer `version: 2` (deformer_torch.py:68-75): closed-form inverse skinning of the roots -----------------
extern "C" int ia_snarf_inverse_skinning(const float *xc, const float *xd, const int32_t *cand_pt, int n_init, const uint8_t *valid, long n,
                                         const int32_t *n_dev, const float *voxel_w, int channel_last, const ia_snarf_grid *grid,
                                         const float *tfs, float *out, void *stream) {
  IA_CHECK_ARG(n >= 0, "ia_snarf_inverse_skinning: n < 0");
  if (n == 0) return IA_OK;
  IA_CHECK_ARG(xc && xd && voxel_w && grid && tfs && out && (cand_pt || n_init > 0), "ia_snarf_inverse_skinning: null pointer / n_init");
  const int blocks = ia_implicit_blocks(n);
  const SnarfGridDev g = ia_make_grid_dev(grid);
  if (channel_last)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_inverse_skinning<true>), dim3(blocks), dim3(IA_ID_THREADS), 0, (hipStream_t)stream, xc, xd, cand_pt, n_init,
                       valid, n, n_dev, voxel_w, g, tfs, out);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_inverse_skinning<false>), dim3(blocks), dim3(IA_ID_THREADS), 0, (hipStream_t)stream, xc, xd, cand_pt, n_init,
                       valid, n, n_dev, voxel_w, g, tfs, out);
  IA_LAUNCH_CHECK("k_inverse_skinning");
  return IA_OK;
}

extern "C" int ia_snarf_inverse_skinning_bwd(const float *xc, const float *xd, const int32_t *cand_pt, int n_init, const uint8_t *valid,
                                             const float *grad_out, long n, const int32_t *n_dev, const float *voxel_w, int channel_last,
                                             const ia_snarf_grid *grid, const float *tfs, float *d_tfs, float *d_xd_entry, void *ws,
                                             size_t ws_bytes, void *stream) {
  const InvSkinArgs a = {xd, cand_pt, n_init, tfs, d_xd_entry};
  return ia_implicit_bwd_impl("ia_snarf_inverse_skinning_bwd", xc, nullptr, valid, grad_out, n, n_dev, voxel_w, grid, d_tfs, ws, ws_bytes,
                              (hipStream_t)stream, channel_last != 0, &a);
}

extern "C" int ia_expand_candidate_points(const int32_t *pt_off, const uint8_t *pt_cnt, int P, const int32_t *n_pts_dev, int32_t *cand_pt,
                                          int cap, void *stream) {
  IA_CHECK_ARG(P >= 0 && cap >= 0, "ia_expand_candidate_points: negative size");
  if (P == 0) return IA_OK;
  IA_CHECK_ARG(pt_off && pt_cnt && cand_pt, "ia_expand_candidate_points: null pointer");
  hipLaunchKernelGGL(k_expand_candidate_points, dim3(ia_div_up(P, 256)), dim3(256), 0, (hipStream_t)stream, pt_off, pt_cnt, P, n_pts_dev, cand_pt, cap);
  IA_LAUNCH_CHECK("k_expand_candidate_points");
  return IA_OK;
}
