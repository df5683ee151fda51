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

#ifndef IA_PRE_VPT
#define IA_PRE_VPT 4  // consecutive voxels (along W) per thread: 1, 2 or 4 (measured 173 / 185 / 95 us)
#endif
template <int VPT> struct PreVec;
template <> struct PreVec<1> { typedef float type; };
template <> struct PreVec<2> { typedef float2 type; };
template <> struct PreVec<4> { typedef float4 type; };

template <int VPT>
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
    float4 *o = reinterpret_cast<float4 *>(voxel_J + (size_t)index0 * 12);
#pragma unroll
    for (int v = 0; v < VPT; v++) {
      o[3 * v + 0] = make_float4(J[v][0], J[v][1], J[v][2], J[v][3]);
      o[3 * v + 1] = make_float4(J[v][4], J[v][5], J[v][6], J[v][7]);
      o[3 * v + 2] = make_float4(J[v][8], J[v][9], J[v][10], J[v][11]);
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
// a4: Broyden.  Trilinear fetch of the 12-channel transform grid, zero padding,
// align_corners=true (fuse_cuda_kernel_fast.cu:62-108,110-230).  J is
// channel-last: a corner is 3 x float4.
// ---------------------------------------------------------------------------

// All 24 loads of a fetch are issued before the first use: corners outside the grid (zero
// padding) read a clamped in-range address and get weight 0 -- fma(v, 0, acc) == acc for the
// finite table values, so the result equals the reference's "skip the corner" bit for bit --
// which removes the per-corner control flow that would serialise eight memory round trips.
__device__ __forceinline__ bool fetch_J(const float *__restrict__ vJ, const SnarfGridDev &g, float gx,
                                        float gy, float gz, float *__restrict__ out) {
  const float ix = src_index(gx, g.W), iy = src_index(gy, g.H), iz = src_index(gz, g.D);
  const int x0 = (int)floorf(ix), y0 = (int)floorf(iy), z0 = (int)floorf(iz);
  const int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
  const float fx1 = x1 - ix, fx0 = ix - x0, fy1 = y1 - iy, fy0 = iy - y0, fz1 = z1 - iz, fz0 = iz - z0;
  // weights in the reference order tnw,tne,tsw,tse,bnw,bne,bsw,bse (:188-195)
  const float wgt[8] = {fx1 * fy1 * fz1, fx0 * fy1 * fz1, fx1 * fy0 * fz1, fx0 * fy0 * fz1,
                        fx1 * fy1 * fz0, fx0 * fy1 * fz0, fx1 * fy0 * fz0, fx0 * fy0 * fz0};
  const bool bx0 = x0 >= 0 && x0 < g.W, bx1 = x1 >= 0 && x1 < g.W;
  const bool by0 = y0 >= 0 && y0 < g.H, by1 = y1 >= 0 && y1 < g.H;
  const bool bz0 = z0 >= 0 && z0 < g.D, bz1 = z1 >= 0 && z1 < g.D;
  const int cx0 = min(max(x0, 0), g.W - 1), cx1 = min(max(x1, 0), g.W - 1);
  const int cy0 = min(max(y0, 0), g.H - 1), cy1 = min(max(y1, 0), g.H - 1);
  const int cz0 = min(max(z0, 0), g.D - 1), cz1 = min(max(z1, 0), g.D - 1);
#ifndef IA_FETCH_SKIP_OUTSIDE
#define IA_FETCH_SKIP_OUTSIDE 1
#endif
#ifndef IA_FETCH_GROUP
#define IA_FETCH_GROUP 2  // corners whose loads are in flight together (four round trips, 24 data VGPRs): with 32 points per
#endif                    // workgroup this buys a fifth wave per SIMD (r02: 376 -> 391 frames/s; group 4 with 32 points: 325)
  // accumulators as float2 pairs: the 12 FMAs of a corner become 6 v_pk_fma_f32 (IEEE fma per half)
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 acc[6];
#pragma unroll
  for (int c = 0; c < 6; c++) acc[c] = (f2){0.f, 0.f};
#if IA_FETCH_SKIP_OUTSIDE
  // A fetch whose eight corners all lie outside the grid is zero (every weight is 0 and fma(v, 0, +0) = +0 for the finite
  // table values): the lane sits the loads out -- the L1 looks up active lanes only.  Diverging solves jump far away:
  // their second fetch is often of this kind.
  if (!((bx0 || bx1) && (by0 || by1) && (bz0 || bz1))) {
#pragma unroll
    for (int c = 0; c < 12; c++) out[c] = 0.f;
    return false;
  }
#endif
#pragma unroll
  for (int k0 = 0; k0 < 8; k0 += IA_FETCH_GROUP) {
    float4 ra[IA_FETCH_GROUP], rb[IA_FETCH_GROUP], rc[IA_FETCH_GROUP];
#pragma unroll
    for (int j = 0; j < IA_FETCH_GROUP; j++) {
      const int k = k0 + j;
      const int xx = (k & 1) ? cx1 : cx0, yy = (k & 2) ? cy1 : cy0, zz = (k & 4) ? cz1 : cz0;
      const float4 *p = reinterpret_cast<const float4 *>(vJ + (uint32_t)((zz * g.H + yy) * g.W + xx) * 12u);
      ra[j] = p[0]; rb[j] = p[1]; rc[j] = p[2];
    }
#pragma unroll
    for (int j = 0; j < IA_FETCH_GROUP; j++) {
      const int k = k0 + j;
      const bool in = ((k & 1) ? bx1 : bx0) && ((k & 2) ? by1 : by0) && ((k & 4) ? bz1 : bz0);
      const float w = in ? wgt[k] : 0.f;
      const f2 w2 = (f2){w, w};
      const f2 v[6] = {(f2){ra[j].x, ra[j].y}, (f2){ra[j].z, ra[j].w}, (f2){rb[j].x, rb[j].y},
                       (f2){rb[j].z, rb[j].w}, (f2){rc[j].x, rc[j].y}, (f2){rc[j].z, rc[j].w}};
#pragma unroll
      for (int q = 0; q < 6; q++) acc[q] = __builtin_elementwise_fma(v[q], w2, acc[q]);
    }
    if (IA_FETCH_GROUP < 8) {
      asm volatile("" ::: "memory");  // the next group's loads stay below this group's FMAs
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int c = 0; c < 6; c++) { out[2 * c] = acc[c].x; out[2 * c + 1] = acc[c].y; }
  return true;
}

// ---- quad-cooperative trilinear fetch ------------------------------------------------------------------------------
// A lane fetching its own 8 corners issues 24 divergent 16-byte loads, and the CU's vector L1 serves a divergent load one
// (lane, 64-byte segment) look-up at a time: 24 look-ups per fetch, the limit k_search ran at in round 2 (0.9 look-ups per
// clock and CU, profiles/r02_pmc_search.json).  Here the four lanes of a quad serve their four fetches one after the other:
// in round T every lane learns the target lane's corner offsets and weights (DPP quad broadcasts), lane k < 3 loads piece k
// (row k of the 3x4 transform) of each of the 8 corner records -- the three loads of a quad fall into ONE 48-byte record,
// i.e. 1-2 segments instead of 3 separate look-ups, and the L1 sees 8 load instructions per fetch instead of 24 -- and
// accumulates ITS ROW over the corners in the reference order (fuse_cuda_kernel_fast.cu:188-226: every output element is
// the same fma chain as before, on another lane), then the three rows return to the target lane by DPP.
// tools/ubench/records64.hip (B4 against A): 33.5 against 21.9 G fetches/s L2-resident, 53.5 against 35.8 L1-resident.
// All DPP traffic happens in wave-uniform control flow (a DPP read from a lane that EXEC has switched off returns
// nothing); only the loads are predicated.
// MEASURED IN THE KERNEL (round 3, MI355X; results bit-identical to the lane-per-fetch path, the parity tests pass with
// either).  First version (predicated loads with zero-filled registers, DPP broadcasts with an initialised `old`, 128 x 32
// workgroups): L1 look-ups 98.5 M -> 57.7 M per launch as predicted, but VALU instructions 47.9 M -> 110 M and 143 VGPRs
// (3 waves per SIMD): VALU-bound, 297 us against 246 us for the compact search of a frame's 213 k sample points.
// This version (mov_dpp folded into the consuming v_add / v_cndmask, unconditional clamped loads, 256 x 64 workgroups):
// 66.9 M VALU instructions, 69.6 M look-ups, 124 VGPRs (4 waves per SIMD): 218 us against 245 us in isolation (-11 %),
// 485 -> 501 frames/s for the whole frame.  IA_QUAD_GROUP-style splitting of a round, a forced fifth wave (96 VGPRs,
// spills: 380 us) and 128 x 32 / 128 x 64 / 256 x 128 / 512 x 128 / 64 x 16 workgroups (234 / 224 / 224 / 226 / 259 us)
// measured and rejected.  tools/ab_search.sh "-DIA_SEARCH_QUAD=0" gives the lane-per-fetch path.
// Later in round 3 the way the rows travel changed (IA_QUAD_LDS_DELIVER below): 199 us.
#ifndef IA_SEARCH_QUAD
#define IA_SEARCH_QUAD 1
#endif
#ifndef IA_SEARCH_QPS
#define IA_SEARCH_QPS 0   // 1: quad-per-solve state machine (see k_search); takes precedence over IA_SEARCH_QUAD
#endif
#ifndef IA_QUAD_PREDICATE
#define IA_QUAD_PREDICATE 1   // 0: every quad loads in every round (no EXEC juggling, wasted look-ups for idle lanes)
#endif
template <int S> __device__ __forceinline__ float quad_bcast(float v) {
  // (mov_dpp = update_dpp with an undefined `old` and bound_ctrl: one v_mov_b32_dpp, which the DPP combiner can fold into
  // the VOP2 instruction that consumes it; all four lanes of a quad are always enabled where this is used)
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), S * 0x55, 0xF, 0xF, true));
}
template <int S> __device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, S * 0x55, 0xF, 0xF, true);
}

// what a lane contributes to its round: BYTE offsets of the 8 corner records (clamped into the grid) and their weights
// (0 for corners outside, and for a lane that is not active) in the reference order, and whether anything is needed at all
struct FetchPlan {
  uint32_t off[8];
  float w[8];
  uint32_t load;   // 1: the lane is active and at least one corner lies inside the grid
};
#ifndef IA_PLAN_FACTOR_ZERO
#define IA_PLAN_FACTOR_ZERO 0
#endif
__device__ __forceinline__ void fetch_plan(const SnarfGridDev &g, float gx, float gy, float gz, bool active, FetchPlan &p) {
  const float ix = src_index(gx, g.W), iy = src_index(gy, g.H), iz = src_index(gz, g.D);
  const int x0 = (int)floorf(ix), y0 = (int)floorf(iy), z0 = (int)floorf(iz);
  const int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
  const float fx1 = x1 - ix, fx0 = ix - x0, fy1 = y1 - iy, fy0 = iy - y0, fz1 = z1 - iz, fz0 = iz - z0;
#if IA_PLAN_FACTOR_ZERO
  // (prepared for round 4, NOT measured and not yet run through the parity tests: validity folded into the six 1-D factors --
  // one unsigned compare and one select per axis end instead of two compares per end, three-way ANDs and eight selects on the
  // products.  A zeroed factor makes its four products exactly +0: the factors are finite and non-negative (src_index maps
  // NaN / huge coordinates to -100), and the multiplication order of the products is unchanged.)
  const bool bx0 = (uint32_t)x0 < (uint32_t)g.W, bx1 = (uint32_t)x1 < (uint32_t)g.W;
  const bool by0 = (uint32_t)y0 < (uint32_t)g.H, by1 = (uint32_t)y1 < (uint32_t)g.H;
  const bool bz0 = (uint32_t)z0 < (uint32_t)g.D, bz1 = (uint32_t)z1 < (uint32_t)g.D;
  const float qx0 = bx0 ? fx1 : 0.f, qx1 = bx1 ? fx0 : 0.f, qy0 = by0 ? fy1 : 0.f, qy1 = by1 ? fy0 : 0.f, qz0 = bz0 ? fz1 : 0.f, qz1 = bz1 ? fz0 : 0.f;
  const float wgt[8] = {qx0 * qy0 * qz0, qx1 * qy0 * qz0, qx0 * qy1 * qz0, qx1 * qy1 * qz0,
                        qx0 * qy0 * qz1, qx1 * qy0 * qz1, qx0 * qy1 * qz1, qx1 * qy1 * qz1};
#else
  const float wgt[8] = {fx1 * fy1 * fz1, fx0 * fy1 * fz1, fx1 * fy0 * fz1, fx0 * fy0 * fz1,
                        fx1 * fy1 * fz0, fx0 * fy1 * fz0, fx1 * fy0 * fz0, fx0 * fy0 * fz0};
  const bool bx0 = x0 >= 0 && x0 < g.W, bx1 = x1 >= 0 && x1 < g.W;
  const bool by0 = y0 >= 0 && y0 < g.H, by1 = y1 >= 0 && y1 < g.H;
  const bool bz0 = z0 >= 0 && z0 < g.D, bz1 = z1 >= 0 && z1 < g.D;
#endif
  const int cx0 = min(max(x0, 0), g.W - 1), cx1 = min(max(x1, 0), g.W - 1);
  const int cy0 = min(max(y0, 0), g.H - 1), cy1 = min(max(y1, 0), g.H - 1);
  const int cz0 = min(max(z0, 0), g.D - 1), cz1 = min(max(z1, 0), g.D - 1);
  // byte offsets as sums of three per-axis terms: six 24-bit multiplies (full rate) and twelve adds instead of fourteen
  // 32-bit multiplies (quarter rate) -- the clamped indices and the strides are far below 2^24 (no measurable change: 201 us)
  const uint32_t sy = (uint32_t)g.W * 48u, sz = (uint32_t)(g.W * g.H) * 48u;
  const uint32_t xo[2] = {(uint32_t)__umul24((uint32_t)cx0, 48u), (uint32_t)__umul24((uint32_t)cx1, 48u)};
  const uint32_t yo[2] = {(uint32_t)__umul24((uint32_t)cy0, sy), (uint32_t)__umul24((uint32_t)cy1, sy)};
  const uint32_t zo[2] = {(uint32_t)__umul24((uint32_t)cz0, sz), (uint32_t)__umul24((uint32_t)cz1, sz)};
  const uint32_t zy[4] = {zo[0] + yo[0], zo[0] + yo[1], zo[1] + yo[0], zo[1] + yo[1]};
#pragma unroll
  for (int k = 0; k < 8; k++) {
#if IA_PLAN_FACTOR_ZERO
    p.off[k] = zy[k >> 1] + xo[k & 1];
    p.w[k] = wgt[k];
#else
    const bool in = ((k & 1) ? bx1 : bx0) && ((k & 2) ? by1 : by0) && ((k & 4) ? bz1 : bz0);
    p.off[k] = zy[k >> 1] + xo[k & 1];
    p.w[k] = in ? wgt[k] : 0.f;
#endif
  }
  p.load = (active && (bx0 || bx1) && (by0 || by1) && (bz0 || bz1)) ? 1u : 0u;
}

// round T of the quad: serve the fetch of lane T.  Wave-uniform control flow throughout: every lane loads (addresses are
// clamped into the grid, so a load is always legal; the fourth lane of a quad repeats the third piece -- the same 16 bytes,
// no extra look-up), and a fetch with all corners outside has all weights 0: fma(v, 0, +0) = +0 for the finite table values,
// exactly the zeros the reference's skipped corners leave.
// IA_QUAD_LDS_DELIVER -- how the rows get back to the lane that needs them (all variants bit-identical, the parity tests pass
// with each; compact search of a frame's 213 k sample points, tools/bench_search.py, MI355X):
//   0  four rounds, rows return by DPP (12 moves + 12 selects per round)                                          218.6 us
//   1  four rounds, rows return through LDS: one 16-byte store per row lane and round, three 16-byte loads per
//      lane and step (12 KB per workgroup); the compiler also drops lane 3's duplicate loads                     212.4 us
//   2  as 1, and the 12 (target, row) pairs of a quad are dealt to its FOUR lanes in THREE rounds (below)         198.9 us
//   3  the deal of 2 in TWO round trips of 12 loads (128 VGPRs, still 4 waves)                                     204.0 us
// (3 against 2: the number of dependent round trips is not what bounds the step; the load instructions are -- 32 / 32 / 24 / 24.)
#ifndef IA_QUAD_LDS_DELIVER
#define IA_QUAD_LDS_DELIVER 2
#endif
#ifndef IA_QUAD_HALF_ROUNDS
#define IA_QUAD_HALF_ROUNDS 0
#endif
#ifndef IA_QUAD_ASM_DPP_ADD
#define IA_QUAD_ASM_DPP_ADD 0
#endif
#if IA_QUAD_LDS_DELIVER >= 2
// Round R, lane k serves pair 4R + k = (target (4R + k) / 3, row (4R + k) % 3): the source lane of every DPP read is a per-lane
// constant of the round -- quad_perm [0,0,0,1], [1,1,2,2], [2,3,3,3] -- and the row lands in float4 number 4R + k of the quad's
// 12-float4 block in LDS, which is exactly where target lane t reads its rows 3t .. 3t + 2.  No lane idles (the 4-round deal
// leaves lane 3 without a row), a step is three dependent load round trips instead of four, 24 load instructions instead of 32.
template <int PERM> __device__ __forceinline__ uint32_t quad_perm(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, PERM, 0xF, 0xF, true);
}
template <int PERM> __device__ __forceinline__ float quad_perm(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), PERM, 0xF, 0xF, true));
}
template <int R>
__device__ __forceinline__ void fetch_round3(const char *__restrict__ vJb, const FetchPlan &p, float4 *__restrict__ s_quad_k) {
  constexpr int PERM = R == 0 ? 0x40 : (R == 1 ? 0xA5 : 0xFE);
  const uint32_t load = quad_perm<PERM>(p.load);
  if (__ballot(load != 0) == 0) return;
  const uint32_t koff = (uint32_t)(((threadIdx.x & 3) + R) % 3) * 16u;   // row (4R + k) % 3 = (k + R) % 3
#if IA_QUAD_HALF_ROUNDS
  // (four loads in flight per half round, offsets and weights broadcast just in time: 92 VGPRs, a fifth wave per SIMD without
  // spills -- MEASURED 202.8 us against 196.5 us for whole rounds at four waves: occupancy is not what this kernel lacks.  OFF.)
  typedef float f2h __attribute__((ext_vector_type(2)));
  f2h h0 = (f2h){0.f, 0.f}, h1 = (f2h){0.f, 0.f};
#pragma unroll
  for (int c0 = 0; c0 < 8; c0 += 4) {
    uint32_t off4[4];
    float w4[4];
#pragma unroll
    for (int c = 0; c < 4; c++) { off4[c] = quad_perm<PERM>(p.off[c0 + c]); w4[c] = quad_perm<PERM>(p.w[c0 + c]); }
    if (load != 0) {
      float4 v4[4];
#pragma unroll
      for (int c = 0; c < 4; c++) v4[c] = *reinterpret_cast<const float4 *>(vJb + (size_t)(off4[c] + koff));
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const f2h w2 = (f2h){w4[c], w4[c]};
        h0 = __builtin_elementwise_fma((f2h){v4[c].x, v4[c].y}, w2, h0);
        h1 = __builtin_elementwise_fma((f2h){v4[c].z, v4[c].w}, w2, h1);
      }
    }
  }
  if (load != 0) s_quad_k[4 * R] = make_float4(h0.x, h0.y, h1.x, h1.y);
  return;
#endif
  // (all DPP reads before the divergent part: a source lane that sits out this round must still be enabled when it is read)
  uint32_t off[8];
  float w[8];
#if IA_QUAD_ASM_DPP_ADD
  // (prepared for round 4, NOT measured and not yet through the parity tests: offset broadcast and row-offset add as ONE
  // v_add_u32_dpp -- the compiler emits v_mov_b32_dpp + v_add_u32 because it sinks the add to the predicated loads.  The two
  // wait states a DPP read needs after a VALU write of its source are the s_nop: inline asm is opaque to the hazard recogniser.)
  asm volatile("s_nop 1");
#pragma unroll
  for (int c = 0; c < 8; c++) {
    asm volatile("v_add_u32_dpp %0, %1, %2 quad_perm:[%3,%4,%5,%6] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                 : "=v"(off[c]) : "v"(p.off[c]), "v"(koff), "i"(PERM & 3), "i"((PERM >> 2) & 3), "i"((PERM >> 4) & 3), "i"((PERM >> 6) & 3));
    w[c] = quad_perm<PERM>(p.w[c]);
  }
  const uint32_t kadd = 0;
#else
#pragma unroll
  for (int c = 0; c < 8; c++) { off[c] = quad_perm<PERM>(p.off[c]); w[c] = quad_perm<PERM>(p.w[c]); }
  const uint32_t kadd = koff;
#endif
  if (load != 0) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    float4 v[8];
#pragma unroll
    for (int c = 0; c < 8; c++) v[c] = *reinterpret_cast<const float4 *>(vJb + (size_t)(off[c] + kadd));
    f2 a0 = (f2){0.f, 0.f}, a1 = (f2){0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const f2 w2 = (f2){w[c], w[c]};
      a0 = __builtin_elementwise_fma((f2){v[c].x, v[c].y}, w2, a0);
      a1 = __builtin_elementwise_fma((f2){v[c].z, v[c].w}, w2, a1);
    }
    s_quad_k[4 * R] = make_float4(a0.x, a0.y, a1.x, a1.y);
  }
}
// IA_QUAD_LDS_DELIVER == 3: the same deal in TWO round trips of 12 loads -- pair k whole and the first four corners of pair 4 + k,
// then the last four corners of pair 4 + k (the fma chain of a row continues in its own registers) and pair 8 + k whole.
template <int PERM, int C0, int C1>
__device__ __forceinline__ void quad_issue(const char *__restrict__ vJb, const FetchPlan &p, uint32_t koff, float4 *__restrict__ v) {
#pragma unroll
  for (int c = C0; c < C1; c++) v[c - C0] = *reinterpret_cast<const float4 *>(vJb + (size_t)(quad_perm<PERM>(p.off[c]) + koff));
}
typedef float iaf2 __attribute__((ext_vector_type(2)));
template <int C0, int C1>
__device__ __forceinline__ void quad_chain(const float4 *__restrict__ v, const float *__restrict__ w, iaf2 &a0, iaf2 &a1) {
#pragma unroll
  for (int c = C0; c < C1; c++) {
    const iaf2 w2 = (iaf2){w[c], w[c]};
    a0 = __builtin_elementwise_fma((iaf2){v[c - C0].x, v[c - C0].y}, w2, a0);
    a1 = __builtin_elementwise_fma((iaf2){v[c - C0].z, v[c - C0].w}, w2, a1);
  }
}
__device__ __forceinline__ void fetch_rounds_2x12(const char *__restrict__ vJb, const FetchPlan &p, float4 *__restrict__ s_quad_k) {
  constexpr int P0 = 0x40, P1 = 0xA5, P2 = 0xFE;
  const uint32_t l0 = quad_perm<P0>(p.load), l1 = quad_perm<P1>(p.load), l2 = quad_perm<P2>(p.load);
  if (__ballot((l0 | l1 | l2) != 0) == 0) return;
  const int k = threadIdx.x & 3;
  const uint32_t k0 = (uint32_t)(k % 3) * 16u, k1 = (uint32_t)((k + 1) % 3) * 16u, k2 = (uint32_t)((k + 2) % 3) * 16u;
  // (every DPP read in wave-uniform control flow: the source lane must be enabled; addresses before the loads, weights after
  // their issue, so that the 12 x 4 registers of load data are the only large live set)
  float4 va[8], vb[4];
  {
    uint32_t oa[8], ob[4];
#pragma unroll
    for (int c = 0; c < 8; c++) oa[c] = quad_perm<P0>(p.off[c]) + k0;
#pragma unroll
    for (int c = 0; c < 4; c++) ob[c] = quad_perm<P1>(p.off[c]) + k1;
    if (l0 != 0) {
#pragma unroll
      for (int c = 0; c < 8; c++) va[c] = *reinterpret_cast<const float4 *>(vJb + (size_t)oa[c]);
    }
    if (l1 != 0) {
#pragma unroll
      for (int c = 0; c < 4; c++) vb[c] = *reinterpret_cast<const float4 *>(vJb + (size_t)ob[c]);
    }
  }
  iaf2 b0 = (iaf2){0.f, 0.f}, b1 = (iaf2){0.f, 0.f};
  {
    float wa[8], wb[8];
#pragma unroll
    for (int c = 0; c < 8; c++) wa[c] = quad_perm<P0>(p.w[c]);
#pragma unroll
    for (int c = 0; c < 4; c++) wb[c] = quad_perm<P1>(p.w[c]);
    if (l0 != 0) {
      iaf2 a0 = (iaf2){0.f, 0.f}, a1 = (iaf2){0.f, 0.f};
      quad_chain<0, 8>(va, wa, a0, a1);
      s_quad_k[0] = make_float4(a0.x, a0.y, a1.x, a1.y);
    }
    if (l1 != 0) quad_chain<0, 4>(vb, wb, b0, b1);
  }
  {
    uint32_t oa[8], ob[4];
#pragma unroll
    for (int c = 0; c < 4; c++) ob[c] = quad_perm<P1>(p.off[4 + c]) + k1;
#pragma unroll
    for (int c = 0; c < 8; c++) oa[c] = quad_perm<P2>(p.off[c]) + k2;
    if (l1 != 0) {
#pragma unroll
      for (int c = 0; c < 4; c++) vb[c] = *reinterpret_cast<const float4 *>(vJb + (size_t)ob[c]);
    }
    if (l2 != 0) {
#pragma unroll
      for (int c = 0; c < 8; c++) va[c] = *reinterpret_cast<const float4 *>(vJb + (size_t)oa[c]);
    }
  }
  {
    float wa[8], wb[8];
#pragma unroll
    for (int c = 4; c < 8; c++) wb[c] = quad_perm<P1>(p.w[c]);
#pragma unroll
    for (int c = 0; c < 8; c++) wa[c] = quad_perm<P2>(p.w[c]);
    if (l1 != 0) {
      quad_chain<4, 8>(vb, wb, b0, b1);
      s_quad_k[4] = make_float4(b0.x, b0.y, b1.x, b1.y);
    }
    if (l2 != 0) {
      iaf2 a0 = (iaf2){0.f, 0.f}, a1 = (iaf2){0.f, 0.f};
      quad_chain<0, 8>(va, wa, a0, a1);
      s_quad_k[8] = make_float4(a0.x, a0.y, a1.x, a1.y);
    }
  }
}
#endif
template <int T>
__device__ __forceinline__ void fetch_round(const char *__restrict__ vJb, const FetchPlan &p, uint32_t koff, bool not_mine,
                                            float *__restrict__ out, float4 *__restrict__ s_del = nullptr) {
  const uint32_t load = quad_bcast<T>(p.load);
  if (__ballot(load != 0) == 0) return;   // no lane T of this wave needs anything: `out` stays zero (cleared by the caller)
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 a0 = (f2){0.f, 0.f}, a1 = (f2){0.f, 0.f};
#if IA_QUAD_PREDICATE
  // quads whose lane T is idle (or has all 8 corners outside) sit the round out: `load` is uniform within a quad, so the
  // DPP broadcasts inside the branch read enabled lanes only
  if (load != 0)
#endif
  {
#ifndef IA_QUAD_GROUP
#define IA_QUAD_GROUP 8   // corner records in flight per load phase of a round (8: one round trip per round; 4: half the data registers)
#endif
#pragma unroll
    for (int c0 = 0; c0 < 8; c0 += IA_QUAD_GROUP) {
      float4 v[IA_QUAD_GROUP];
#pragma unroll
      for (int c = 0; c < IA_QUAD_GROUP; c++) v[c] = *reinterpret_cast<const float4 *>(vJb + (size_t)(quad_bcast<T>(p.off[c0 + c]) + koff));
#pragma unroll
      for (int c = 0; c < IA_QUAD_GROUP; c++) {
        const float w = quad_bcast<T>(p.w[c0 + c]);
        const f2 w2 = (f2){w, w};
        a0 = __builtin_elementwise_fma((f2){v[c].x, v[c].y}, w2, a0);
        a1 = __builtin_elementwise_fma((f2){v[c].z, v[c].w}, w2, a1);
      }
      if (IA_QUAD_GROUP < 8) { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
    }
  }
#if IA_QUAD_LDS_DELIVER
  // row k of target T's transform: one ds_write_b128 into the target lane's slot (the target reads its three rows after the
  // fourth round; LDS operations of one wave execute in order, the quad is inside one wave)
  if (load != 0 && (threadIdx.x & 3) < 3)
    s_del[((threadIdx.x & ~3u) | T) * 3 + (threadIdx.x & 3)] = make_float4(a0.x, a0.y, a1.x, a1.y);
  return;
#endif
  // rows 0..2 (lanes 0..2 of the quad) back to the target lane: out = not_mine ? out : row (v_cndmask with a DPP source)
  const float a[4] = {a0.x, a0.y, a1.x, a1.y};
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const float r0 = quad_bcast<0>(a[c]), r1 = quad_bcast<1>(a[c]), r2 = quad_bcast<2>(a[c]);
    out[c] = not_mine ? out[c] : r0;
    out[4 + c] = not_mine ? out[4 + c] : r1;
    out[8 + c] = not_mine ? out[8 + c] : r2;
  }
}

// two rounds with their loads in flight together (IA_QUAD_PAIR): half the round trips per step, twice the data registers
template <int T0, int T1>
__device__ __forceinline__ void fetch_round2(const char *__restrict__ vJb, const FetchPlan &p, uint32_t koff, int k,
                                             float *__restrict__ out) {
  const uint32_t l0 = quad_bcast<T0>(p.load), l1 = quad_bcast<T1>(p.load);
  if (__ballot((l0 | l1) != 0) == 0) return;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 a0 = (f2){0.f, 0.f}, a1 = (f2){0.f, 0.f}, b0 = (f2){0.f, 0.f}, b1 = (f2){0.f, 0.f};
  float4 v[8], u[8];
#pragma unroll
  for (int c = 0; c < 8; c++) v[c] = *reinterpret_cast<const float4 *>(vJb + (size_t)(quad_bcast<T0>(p.off[c]) + koff));
#pragma unroll
  for (int c = 0; c < 8; c++) u[c] = *reinterpret_cast<const float4 *>(vJb + (size_t)(quad_bcast<T1>(p.off[c]) + koff));
#pragma unroll
  for (int c = 0; c < 8; c++) {
    const float w = quad_bcast<T0>(p.w[c]);
    const f2 w2 = (f2){w, w};
    a0 = __builtin_elementwise_fma((f2){v[c].x, v[c].y}, w2, a0);
    a1 = __builtin_elementwise_fma((f2){v[c].z, v[c].w}, w2, a1);
  }
#pragma unroll
  for (int c = 0; c < 8; c++) {
    const float w = quad_bcast<T1>(p.w[c]);
    const f2 w2 = (f2){w, w};
    b0 = __builtin_elementwise_fma((f2){u[c].x, u[c].y}, w2, b0);
    b1 = __builtin_elementwise_fma((f2){u[c].z, u[c].w}, w2, b1);
  }
  const float a[4] = {a0.x, a0.y, a1.x, a1.y}, b[4] = {b0.x, b0.y, b1.x, b1.y};
  const bool m0 = k != T0, m1 = k != T1;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const float r0 = quad_bcast<0>(a[c]), r1 = quad_bcast<1>(a[c]), r2 = quad_bcast<2>(a[c]);
    out[c] = m0 ? out[c] : r0; out[4 + c] = m0 ? out[4 + c] : r1; out[8 + c] = m0 ? out[8 + c] : r2;
    const float s0 = quad_bcast<0>(b[c]), s1 = quad_bcast<1>(b[c]), s2 = quad_bcast<2>(b[c]);
    out[c] = m1 ? out[c] : s0; out[4 + c] = m1 ? out[4 + c] : s1; out[8 + c] = m1 ? out[8 + c] : s2;
  }
}

// the fetch of every lane of the wave (call in wave-uniform control flow); `loaded`: this lane's fetch touched memory
__device__ __forceinline__ void fetch_J_quad(const float *__restrict__ vJ, const SnarfGridDev &g, float gx, float gy, float gz,
                                             bool active, float *__restrict__ out, bool &loaded, float4 *__restrict__ s_del = nullptr) {
  FetchPlan p;
  fetch_plan(g, gx, gy, gz, active, p);
  loaded = p.load != 0;
  const int k = threadIdx.x & 3;
  const uint32_t koff = (uint32_t)min(k, 2) * 16u;
  const char *vJb = reinterpret_cast<const char *>(vJ);
#if IA_QUAD_LDS_DELIVER
  if (p.load == 0) {   // nobody will write this lane's slot: an active lane with all corners outside reads zeros
    s_del[threadIdx.x * 3] = make_float4(0.f, 0.f, 0.f, 0.f); s_del[threadIdx.x * 3 + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
    s_del[threadIdx.x * 3 + 2] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __builtin_amdgcn_wave_barrier();
#if IA_QUAD_LDS_DELIVER >= 2
  {
    float4 *const s_quad_k = s_del + (threadIdx.x & ~3u) * 3 + k;
#if IA_QUAD_LDS_DELIVER == 3
    fetch_rounds_2x12(vJb, p, s_quad_k);
#else
    fetch_round3<0>(vJb, p, s_quad_k);
    fetch_round3<1>(vJb, p, s_quad_k);
    fetch_round3<2>(vJb, p, s_quad_k);
#endif
  }
#else
  fetch_round<0>(vJb, p, koff, k != 0, out, s_del);
  fetch_round<1>(vJb, p, koff, k != 1, out, s_del);
  fetch_round<2>(vJb, p, koff, k != 2, out, s_del);
  fetch_round<3>(vJb, p, koff, k != 3, out, s_del);
#endif
  __builtin_amdgcn_wave_barrier();
  {
    const float4 r0 = s_del[threadIdx.x * 3], r1 = s_del[threadIdx.x * 3 + 1], r2 = s_del[threadIdx.x * 3 + 2];
    out[0] = r0.x; out[1] = r0.y; out[2] = r0.z; out[3] = r0.w; out[4] = r1.x; out[5] = r1.y; out[6] = r1.z; out[7] = r1.w;
    out[8] = r2.x; out[9] = r2.y; out[10] = r2.z; out[11] = r2.w;
  }
  __builtin_amdgcn_wave_barrier();   // the next step's zero-fill / rows must not overtake these reads
  return;
#endif
#pragma unroll
  for (int c = 0; c < 12; c++) out[c] = 0.f;
#ifndef IA_QUAD_PAIR
#define IA_QUAD_PAIR 0
#endif
#ifndef IA_QUAD_FENCE
#define IA_QUAD_FENCE 0   // 1: keep the compiler from hoisting the next round's loads above this round's arithmetic (fewer registers)
#endif
#define IA_QUAD_ROUND_END() do { if (IA_QUAD_FENCE) { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } } while (0)
#if IA_QUAD_PAIR
  fetch_round2<0, 1>(vJb, p, koff, k, out); IA_QUAD_ROUND_END();
  fetch_round2<2, 3>(vJb, p, koff, k, out);
  return;
#endif
  fetch_round<0>(vJb, p, koff, k != 0, out); IA_QUAD_ROUND_END();
  fetch_round<1>(vJb, p, koff, k != 1, out); IA_QUAD_ROUND_END();
  fetch_round<2>(vJb, p, koff, k != 2, out); IA_QUAD_ROUND_END();
  fetch_round<3>(vJb, p, koff, k != 3, out);
}

// fuse_J_inv_update (fuse_cuda_kernel_fast.cu:23-55)
#ifndef IA_SHARED_RCP
#define IA_SHARED_RCP 1
#endif
// ---- nine IEEE divisions by the same denominator ----------------------------------------------------------------------
// The compiler expands a / b into v_div_scale x 2, v_rcp, a Newton chain (fma, fma on the reciprocal; mul, fma, fma, fma on
// the quotient), v_div_fmas, v_div_fixup.  v_div_scale only rescales its operands at the edges of the exponent range (a
// denormal or huge denominator, a numerator below 2^-103, a quotient near overflow / underflow -- CDNA3 ISA, V_DIV_SCALE_F32);
// away from those it returns them unchanged with VCC = 0, and v_div_fmas is then a plain fma.  In that range the reciprocal
// half of the chain depends on the denominator alone (`rcp_refined`, once per update) and the numerator half is the same five
// instructions the compiler emits (`div_shared`); v_div_fixup keeps the zero / inf / NaN cases (it does not look at the
// quotient for those), so the quotients are bit-identical to a / b.  `ia_selftest_shared_rcp` sweeps the exponent range on
// the device (tests/test_gpu_edge_cases.py).
__device__ __forceinline__ float rcp_refined(float s) {
  const float ra = __builtin_amdgcn_rcpf(s);
  return __builtin_fmaf(__builtin_fmaf(-s, ra, 1.0f), ra, ra);
}
__device__ __forceinline__ float div_shared(float n, float s, float rb) {
  const float q0 = n * rb;
  const float q1 = __builtin_fmaf(__builtin_fmaf(-s, q0, n), rb, q0);
  const float q2 = __builtin_fmaf(__builtin_fmaf(-s, q1, n), rb, q1);
  return __builtin_amdgcn_div_fixupf(q2, s, n);
}
// The range in which `div_shared` IS the compiler's division, as a test on binary exponents (v_frexp_exp: |v| in
// [2^(e-1), 2^e); 0 for zero, inf and NaN, which pass and are v_div_fixup's cases).  A numerator is a product c_j t_i: with
// c, t zero or in [2^-49, 2^8) it is zero or in [2^-98, 2^16); with s in [2^-67, 2^22) every exponent difference stays inside
// (-126, 96), no operand is denormal and no numerator is below 2^-103.
__device__ __forceinline__ bool div_shared_range(float c0, float c1, float c2, float t0, float t1, float t2, float s) {
  const int e_hi = max(max(max(__builtin_amdgcn_frexp_expf(c0), __builtin_amdgcn_frexp_expf(c1)), __builtin_amdgcn_frexp_expf(c2)),
                       max(max(__builtin_amdgcn_frexp_expf(t0), __builtin_amdgcn_frexp_expf(t1)), __builtin_amdgcn_frexp_expf(t2)));
  const int e_lo = min(min(min(__builtin_amdgcn_frexp_expf(c0), __builtin_amdgcn_frexp_expf(c1)), __builtin_amdgcn_frexp_expf(c2)),
                       min(min(__builtin_amdgcn_frexp_expf(t0), __builtin_amdgcn_frexp_expf(t1)), __builtin_amdgcn_frexp_expf(t2)));
  const int e_s = __builtin_amdgcn_frexp_expf(s);
  return e_hi <= 8 && e_lo >= -48 && e_s >= -66 && e_s <= 22;
}
// fuse_J_inv_update (fuse_cuda_kernel_fast.cu:23-55).  SHARED: the shared reciprocal when EVERY lane of the wave is inside
// the range, the compiler's divisions otherwise; returns which one ran.  MEASURED: 202.0 -> 199.6 us (tools/bench_search.py).
template <bool SHARED>
__device__ __forceinline__ bool jinv_update_impl(float *Ji, float x0, float x1, float x2, float g0, float g1, float g2) {
  const float J00 = Ji[0], J01 = Ji[1], J02 = Ji[2], J10 = Ji[3], J11 = Ji[4], J12 = Ji[5], J20 = Ji[6],
              J21 = Ji[7], J22 = Ji[8];
  const float c0 = IA_DOT3(J00, x0, J10, x1, J20, x2);
  const float c1 = IA_DOT3(J01, x0, J11, x1, J21, x2);
  const float c2 = IA_DOT3(J02, x0, J12, x1, J22, x2);
  const float s = IA_DOT3(c0, g0, c1, g1, c2, g2);
  const float r0 = IA_DOT3(-J00, g0, -J01, g1, -J02, g2);
  const float r1 = IA_DOT3(-J10, g0, -J11, g1, -J12, g2);
  const float r2 = IA_DOT3(-J20, g0, -J21, g1, -J22, g2);
  if (SHARED) {
    const float t0 = r0 + x0, t1 = r1 + x1, t2 = r2 + x2;
    if (__ballot(!div_shared_range(c0, c1, c2, t0, t1, t2, s)) == 0) {
      const float rb = rcp_refined(s);
      const float tt[3] = {t0, t1, t2}, cc[3] = {c0, c1, c2};
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Ji[3 * i + j] += div_shared(cc[j] * tt[i], s, rb);
      return true;
    }
  }
  Ji[0] += c0 * (r0 + x0) / s; Ji[1] += c1 * (r0 + x0) / s; Ji[2] += c2 * (r0 + x0) / s;
  Ji[3] += c0 * (r1 + x1) / s; Ji[4] += c1 * (r1 + x1) / s; Ji[5] += c2 * (r1 + x1) / s;
  Ji[6] += c0 * (r2 + x2) / s; Ji[7] += c1 * (r2 + x2) / s; Ji[8] += c2 * (r2 + x2) / s;
  return false;
}
__device__ __forceinline__ void jinv_update(float *Ji, float x0, float x1, float x2, float g0, float g1, float g2) {
  (void)jinv_update_impl<IA_SHARED_RCP != 0>(Ji, x0, x1, x2, g0, g1, g2);
}



// ---------------------------------------------------------------------------
// a4 + a5 search kernel with LANE REFILL.
//
// Broyden trip counts are very uneven (most (point, init) pairs diverge at the
// first check after 2 grid fetches, roots need 3..11), so a lane-per-solve
// mapping leaves most of a wave idle while its slowest lane iterates.  Here a
// workgroup owns NP points x n_init solves as a queue in LDS; every lane runs a
// small state machine whose loop body is ONE trilinear fetch, and a lane whose
// solve terminated pulls the next item (wave ballot + one LDS atomic per wave).
// Items are ordered init-major / point-minor, so lanes refilled together start
// from neighbouring canonical positions.  Each solve executes exactly the
// arithmetic sequence of the reference kernel (fuse_cuda_kernel_fast.cu:268-412).
// Afterwards the workgroup runs the duplicate filter and either writes the dense
// reference layout (MODE 0) or compacts the surviving roots (MODE 1).
// ---------------------------------------------------------------------------
// Measured on MI355X (512^2 frame, graph mode), round 1 with 4 corners in flight (116 VGPRs, 4 waves per SIMD):
// 256 threads x 128 points 3.86 ms, 128 x 64 3.70 ms, 64 x 32 3.79 ms, 64 x 64 4.07 ms, 128 x 128 4.19 ms.
// Round 2: occupancy was bound twice at 16 waves per CU (registers AND 18 KB of LDS per workgroup); 2 corners in flight
// (91 VGPRs -> 5 waves per SIMD) together with 32 points per workgroup (9.5 KB -> the LDS allows them): 128 x 32 with
// group 2 = 2.55 ms against 2.66 ms; group 2 alone 2.71, 32 points alone (group 4) 3.08, group 1 (82 VGPRs, still 5
// waves) 2.67, forcing 6 waves per SIMD (spills) 2.86, 64 x 32 2.67, 64 x 16 2.58, 256 x 32 2.68, 256 x 64 2.62.
#ifndef IA_SEARCH_NP
#if IA_SEARCH_QUAD
#define IA_SEARCH_NP 64        // quad-cooperative fetch, round 3: 256 x 64 218 us, 128 x 32 234, 128 x 64 224, 256 x 128 224, 512 x 128 226, 64 x 16 259
                               // re-measured with the three-round deal: 256 x 64 199.6, 128 x 32 206.7, 512 x 128 206.7, 256 x 128 212.1,
                               // 128 x 64 215.3, 64 x 32 223.0, 256 x 32 228.6, 512 x 64 229.2, 1024 x 64 326.3 (threads x points)
#else
#define IA_SEARCH_NP 32        // points per workgroup (power of two, <= 128): 9.6 KB of LDS, 16 workgroups per CU
#endif
#endif
#ifndef IA_SEARCH_THREADS
#if IA_SEARCH_QUAD
#define IA_SEARCH_THREADS 256
#else
#define IA_SEARCH_THREADS 128  // multiple of IA_SEARCH_NP
#endif
#endif
#ifdef IA_SEARCH_WAVES_PER_EU
#define IA_SEARCH_ATTR __attribute__((amdgpu_waves_per_eu(IA_SEARCH_WAVES_PER_EU, IA_SEARCH_WAVES_PER_EU)))
#else
#define IA_SEARCH_ATTR
#endif

#ifndef IA_SEARCH_T_LDS
#define IA_SEARCH_T_LDS 0      // the solve's target x_d: 1 = re-read from LDS at every step (106 VGPRs, 197.7 us), 0 = three registers (108, 196.5 us)
#endif
#ifndef IA_REFILL_GROUP
// lanes refilled together; measured r02 (k_search per launch, frames/s with two frames in flight): 1: 196 us / 485,
// 2: 220 / 455, 4: 234 / 440, 8: 228 / 444 -- waiting for whole quads idles more lanes than the shared L1 look-ups save
#define IA_REFILL_GROUP 1
#endif
#if IA_REFILL_GROUP == 1
#define IA_REFILL_GROUP_HEADS 0xFFFFFFFFFFFFFFFFull
#elif IA_REFILL_GROUP == 2
#define IA_REFILL_GROUP_HEADS 0x5555555555555555ull
#elif IA_REFILL_GROUP == 4
#define IA_REFILL_GROUP_HEADS 0x1111111111111111ull
#elif IA_REFILL_GROUP == 8
#define IA_REFILL_GROUP_HEADS 0x0101010101010101ull
#elif IA_REFILL_GROUP == 16
#define IA_REFILL_GROUP_HEADS 0x0001000100010001ull
#else
#error "IA_REFILL_GROUP: 1, 2, 4, 8 or 16"
#endif

template <int MODE>
__global__ __launch_bounds__(IA_SEARCH_THREADS) IA_SEARCH_ATTR void k_search(
    const float *__restrict__ xd, int P, const int32_t *__restrict__ n_pts_dev,
    const float *__restrict__ vJ, const float *__restrict__ tfs, BoneIds bones, int n_init, SnarfGridDev g,
    float cvg2, float dvg2,
    // MODE 0
    float *__restrict__ xc, uint8_t *__restrict__ valid_out, uint8_t *__restrict__ valid_raw,
    float *__restrict__ J_inv,
    // MODE 1 / 2 (2 = 1 + the Broyden J_inv of every surviving root, compacted like cand_xc: the training route
    // with SMPL parameters under optimisation needs it for the implicit differentiation, deformer_torch.py:58-60)
    float *__restrict__ cand_xc, int cand_cap, int32_t *__restrict__ pt_off, uint8_t *__restrict__ pt_cnt,
    int32_t *__restrict__ n_cand, unsigned long long *prof, float *__restrict__ cand_Jinv, float *__restrict__ jinv_dense) {
  constexpr int NP = IA_SEARCH_NP;
  // (MODE 2: the J_inv of a converged solve goes to `jinv_dense` [P][n_init][9] in global memory -- written for valid solves
  // only, read back by the same workgroup at compaction; 30 KB of LDS for it cost the kernel a wave per SIMD: 282 us per
  // refine step)
  __shared__ float s_x[IA_N_INIT_MAX][NP][3];
  __shared__ float s_xd[NP][3];
  __shared__ uint8_t s_valid[IA_N_INIT_MAX][NP];
  __shared__ uint8_t s_keep[IA_N_INIT_MAX][NP];
  __shared__ int s_base[NP];
  __shared__ int s_wtot[IA_SEARCH_THREADS / 64];
  __shared__ int s_next;
  __shared__ int s_blockbase;
  __shared__ int s_nlive;
  __shared__ int s_prof[3];
  __shared__ uint16_t s_list[IA_N_INIT_MAX * NP];
  __shared__ float s_T[IA_N_INIT_MAX][12];  // rows 0..2 of the init bones' transforms (same indexing as the 4x4)
  __shared__ float4 s_del_store[(IA_SEARCH_QUAD != 0 && IA_QUAD_LDS_DELIVER != 0) ? IA_SEARCH_THREADS * 3 : 1];
  float4 *const s_del = s_del_store;
  if (n_pts_dev) P = min(P, *n_pts_dev);
  const int tid = threadIdx.x, lane = tid & 63;
  // (an XCD-aware remap -- XCD x takes the x-th contiguous eighth of the point list -- was measured:
  // the occupancy probes got 25 % slower, the slabs at the rim of the bounding box hold little live work)
  const int p0 = blockIdx.x * NP;
  if (p0 >= P) return;  // uniform per workgroup
  const int np = min(NP, P - p0);
  const int n_items = np * n_init;
  if (tid == 0) { s_next = 0; s_nlive = 0; s_prof[0] = 0; s_prof[1] = 0; s_prof[2] = 0; }
  for (int e = tid; e < np * 3; e += IA_SEARCH_THREADS) (&s_xd[0][0])[e] = xd[(size_t)p0 * 3 + e];
  for (int e = tid; e < n_init * 12; e += IA_SEARCH_THREADS) s_T[e / 12][e % 12] = tfs[bones.id[e / 12] * 16 + e % 12];
  __syncthreads();

  // ---- classification -------------------------------------------------------------------
  // A solve whose INITIAL fetch has all 8 corners outside the grid is invalid by construction:
  // J = 0 gives J_inv0 = 0, so the update is 0, x never moves, every later fetch is zero too
  // and the residual stays -x_d: the reference kernel ends in `diverged`, in `converged` with
  // the bounds test failing, or (1e-5 < |x_d| < 0.1) in ten iterations of NaN -- never valid.
  // Most (point, init) pairs of the occupancy probes are of this kind; they are resolved here,
  // and only the remaining items enter the queue, so the waves of the solver stay dense.
  // items are (init << 7 | point): init-major, no integer division anywhere
  static_assert((NP & (NP - 1)) == 0 && NP <= 128 && IA_SEARCH_THREADS % NP == 0, "item packing: NP = 2^k <= 128");
  for (int init0 = 0; init0 < n_init; init0 += IA_SEARCH_THREADS / NP) {
    const int init = init0 + tid / NP, pt = tid & (NP - 1);
    const int q = (init << 7) | pt;
    bool keep = false;
    if (init < n_init && pt < np) {
      keep = !ia_solve_is_trivial(g, s_T[init], s_xd[pt][0], s_xd[pt][1], s_xd[pt][2]);
      if (!keep) {
        s_x[init][pt][0] = 0.f; s_x[init][pt][1] = 0.f; s_x[init][pt][2] = 0.f;
        s_valid[init][pt] = 0;
        if (MODE == 0 && J_inv) {
          const size_t o = ((size_t)(p0 + pt) * n_init + init) * 9;
#pragma unroll
          for (int k = 0; k < 9; k++) J_inv[o + k] = 0.f;
        }
      }
    }
    const unsigned long long m = __ballot(keep);
    int base = 0;
    if (lane == 0 && m) base = atomicAdd(&s_nlive, __popcll(m));
    base = __shfl(base, 0, 64);
    if (keep) s_list[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)q;
  }
  __syncthreads();
  const int n_live = s_nlive;

#if IA_SEARCH_QPS
  // ---- QUAD state machine (quad-per-solve) -----------------------------------------------------------------------------
  // One solve per quad: lane r (r = 0..2; lane 3 shadows lane 2 and writes nothing) owns ROW r of everything with rows -- the
  // fetched 3x4 transform, J_inv, the residual / update component r -- while x, the previous residual and the scalars are
  // replicated.  A Broyden step is then ONE round of 8 loads per lane (row r of the 8 corner records) instead of the four
  // rounds of the lane-per-solve mapping, the per-lane state shrinks (J_inv row 3 instead of 9 registers, no delivery of
  // rows to a target lane), and the arithmetic is the reference's sequence element by element: a dot product whose terms
  // live in different lanes walks from lane 0 to lane 2 (multiply, fma, fma) through DPP, so every rounding happens where
  // and in the order it did before.
  // MEASURED (round 3, MI355X; bit-identical results, the parity tests pass with it): 88 VGPRs, 5 waves per SIMD, one round
  // trip per step -- and slower: 269 us against 219 us for the lane-per-solve kernel with the quad-cooperative fetch (256 x 64;
  // 256 x 128: 310 us).  A wave-step costs about the same ~450 VALU instructions whether it advances 16 solves or 64 (the fetch
  // plan, the replicated scalars, the selects and the queue logic do not shrink with the number of solves), so the VALU work
  // per solve doubles and the kernel, at 62 % VALU issue before, turns VALU bound.  OFF; kept as the measured alternative.
  const int r_own = min(lane & 3, 2);
  const bool leader = (lane & 3) == 0, writer = (lane & 3) < 3;
  const uint32_t koff = (uint32_t)r_own * 16u;
  const char *vJb = reinterpret_cast<const char *>(vJ);
  bool active = false, first = false;
  int item = 0, iter = 0, fetches = 0, solves = 0, loaded = 0;
  float t_own = 0, xl0 = 0, xl1 = 0, xl2 = 0, gx0 = 0, gx1 = 0, gx2 = 0, u0 = 0, u1 = 0, u2 = 0;
  float Jr[3] = {0.f, 0.f, 0.f};   // row r_own of J_inv
  bool queue_empty = false;
  while (true) {
    if (!queue_empty) {
      const unsigned long long need = __ballot(!active && leader);
      if (need) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_next, __popcll(need));
        base = __shfl(base, 0, 64);
        if (base >= n_live) queue_empty = true;
        int my = base + __popcll(need & ((1ull << lane) - 1ull));
        my = (int)quad_bcast<0>((uint32_t)my);          // the leader's slot, for the whole quad
        if (!active && my < n_live) {
          item = s_list[my];
          const int init = item >> 7, pt = item & (NP - 1);
          const float ta = s_xd[pt][0], tb = s_xd[pt][1], tc = s_xd[pt][2];
          t_own = r_own == 0 ? ta : (r_own == 1 ? tb : tc);
          const float *T = s_T[init];
          const float ixd = ta - T[3], iyd = tb - T[7], izd = tc - T[11];
          xl0 = IA_DOT3(ixd, T[0], iyd, T[4], izd, T[8]);
          xl1 = IA_DOT3(ixd, T[1], iyd, T[5], izd, T[9]);
          xl2 = IA_DOT3(ixd, T[2], iyd, T[6], izd, T[10]);
          active = true; first = true; iter = 0;
          if (leader) solves++;
        }
      }
    }
    if (!__any(active)) break;
    const float ix = g.scl[0] * (xl0 + g.off[0]);
    const float iy = g.scl[1] * (xl1 + g.off[1]);
    const float iz = g.scl[2] * (xl2 + g.off[2]);
    // ---- fetch: row r_own of the trilinear blend (weights / offsets computed by every lane of the quad: same inputs, same bits)
    FetchPlan p;
    fetch_plan(g, ix, iy, iz, active, p);
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = (f2){0.f, 0.f}, a1 = (f2){0.f, 0.f};
    if (p.load != 0) {     // uniform within a quad; a fetch with all corners outside is zero without a load
      float4 v[8];
#pragma unroll
      for (int c = 0; c < 8; c++) v[c] = *reinterpret_cast<const float4 *>(vJb + (size_t)(p.off[c] + koff));
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const f2 w2 = (f2){p.w[c], p.w[c]};
        a0 = __builtin_elementwise_fma((f2){v[c].x, v[c].y}, w2, a0);
        a1 = __builtin_elementwise_fma((f2){v[c].z, v[c].w}, w2, a1);
      }
    }
    const float Jrow[4] = {a0.x, a0.y, a1.x, a1.y};   // J[r][0..3]
    // ---- everything below runs in wave-uniform control flow (DPP inside): idle quads compute on stale values, results unused
    // residual component r: g_r(x) = J[r] . x + d_r - xd_r  (:325-332 / :356-367)
    const float n_own = IA_DOT3(Jrow[0], xl0, Jrow[1], xl1, Jrow[2], xl2) + Jrow[3] - t_own;
    const float n0 = quad_bcast<0>(n_own), n1 = quad_bcast<1>(n_own), n2 = quad_bcast<2>(n_own);
    bool done = false, ok = false;
    const unsigned long long any_first = __ballot(active && first);
    if (any_first) {
      // :302-311 J_inv0 = (J_3x3)^T: row r of J_inv = column r of J = element r of the three row lanes
      const float b00 = quad_bcast<0>(Jrow[0]), b01 = quad_bcast<0>(Jrow[1]), b02 = quad_bcast<0>(Jrow[2]);
      const float b10 = quad_bcast<1>(Jrow[0]), b11 = quad_bcast<1>(Jrow[1]), b12 = quad_bcast<1>(Jrow[2]);
      const float b20 = quad_bcast<2>(Jrow[0]), b21 = quad_bcast<2>(Jrow[1]), b22 = quad_bcast<2>(Jrow[2]);
      if (active && first) {
        Jr[0] = r_own == 0 ? b00 : (r_own == 1 ? b01 : b02);
        Jr[1] = r_own == 0 ? b10 : (r_own == 1 ? b11 : b12);
        Jr[2] = r_own == 0 ? b20 : (r_own == 1 ? b21 : b22);
      }
    }
    const bool was_first = first;
    // Broyden update of J_inv (fuse_J_inv_update, :23-55) with x-arguments u and g-arguments dg = n - gx:
    //   c_j = J0j u0 + J1j u1 + J2j u2 walks down the rows (lane 0 multiplies, lanes 1 and 2 fma), s and r_r follow
    const float dg0 = n0 - gx0, dg1 = n1 - gx1, dg2 = n2 - gx2;
    const float u_own = r_own == 0 ? u0 : (r_own == 1 ? u1 : u2);
    float c[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const float m = Jr[j] * u_own;                                   // valid in lane 0: J0j * u0
      const float q = __builtin_fmaf(Jr[j], u_own, quad_bcast<0>(m));   // valid in lane 1: fma(J1j, u1, .)
      const float cc = __builtin_fmaf(Jr[j], u_own, quad_bcast<1>(q));  // valid in lane 2: fma(J2j, u2, .)
      c[j] = quad_bcast<2>(cc);
    }
    const float sden = IA_DOT3(c[0], dg0, c[1], dg1, c[2], dg2);
    const float r_r = IA_DOT3(-Jr[0], dg0, -Jr[1], dg1, -Jr[2], dg2);
    if (active) {
      if (was_first) {
        gx0 = n0; gx1 = n1; gx2 = n2;
        first = false;
      } else {
        const float norm = IA_DOT3(n0, n0, n1, n1, n2, n2);
        if (norm < cvg2) {
          done = true;
          ok = ix >= -1 && ix <= 1 && iy >= -1 && iy <= 1 && iz >= -1 && iz <= 1;
        } else if (norm > dvg2) {
          done = true;
        } else {
          Jr[0] += c[0] * (r_r + u_own) / sden; Jr[1] += c[1] * (r_r + u_own) / sden; Jr[2] += c[2] * (r_r + u_own) / sden;  // :400-411
          gx0 = n0; gx1 = n1; gx2 = n2;
          if (++iter == 10) done = true;  // Q1
        }
      }
      if (leader) { fetches++; loaded += p.load ? 1 : 0; }
    }
    if (active && done) {
      const int init = item >> 7, pt = item & (NP - 1);
      if (leader) {
        s_x[init][pt][0] = ok ? xl0 : 0.f; s_x[init][pt][1] = ok ? xl1 : 0.f; s_x[init][pt][2] = ok ? xl2 : 0.f;
        s_valid[init][pt] = ok;
      }
      if (writer) {   // Q4: J_inv as it was BEFORE the last rank-1 update (a converged / diverged step does not update it)
        if (MODE == 0 && J_inv) {
          const size_t o = ((size_t)(p0 + pt) * n_init + init) * 9 + 3 * r_own;
          J_inv[o] = ok ? Jr[0] : 0.f; J_inv[o + 1] = ok ? Jr[1] : 0.f; J_inv[o + 2] = ok ? Jr[2] : 0.f;
        }
        if (MODE == 2 && ok) {
          float *o = jinv_dense + ((size_t)(p0 + pt) * n_init + init) * 9 + 3 * r_own;
          o[0] = Jr[0]; o[1] = Jr[1]; o[2] = Jr[2];
        }
      }
      active = false;
    }
    // :340-351 update = -J_inv g ; x += update (start of the next iteration): component r in lane r, then to the whole quad
    const float un = IA_DOT3(-Jr[0], gx0, -Jr[1], gx1, -Jr[2], gx2);
    const float v0 = quad_bcast<0>(un), v1 = quad_bcast<1>(un), v2 = quad_bcast<2>(un);
    if (active) { u0 = v0; u1 = v1; u2 = v2; xl0 += u0; xl1 += u1; xl2 += u2; }
  }
#else
  // ---- lane state machine ----
  bool active = false, first = false;
  // `solves` counts the queued (non-trivial) ones; `fetches` every trilinear fetch of the reference's algorithm, `loaded` those
  // that touched memory (a fetch with all 8 corners outside the grid is zero by construction and loads nothing)
  // (packed: registers are what bounds the waves per SIMD here -- `counts` = fetches | loaded << 16, a lane sees at most
  // 13 * NP * 11 < 2^16 fetches per launch; `it_solves` = iter | solves << 8; IA_SEARCH_T_LDS: the target x_d re-read from LDS)
  int item = 0;
  uint32_t counts = 0, it_solves = 0;
#if !IA_SEARCH_T_LDS
  float t0 = 0, t1 = 0, t2 = 0;
#endif
  float xl0 = 0, xl1 = 0, xl2 = 0, gx0 = 0, gx1 = 0, gx2 = 0, u0 = 0, u1 = 0, u2 = 0;
  float Ji[9];
#pragma unroll
  for (int k = 0; k < 9; k++) Ji[k] = 0.f;
  bool queue_empty = false;
  while (true) {
    if (!queue_empty) {
      // IA_REFILL_GROUP > 1 (experiment, rejected): refill in aligned groups of lanes, a group taking CONSECUTIVE items
      // (neighbouring samples of a ray under the same init bone) only when all of its lanes are idle, so that the lanes
      // the L1 looks up together (one access per quad and cache line) keep fetching from the same or adjacent cells.
      unsigned long long need = __ballot(!active);
#pragma unroll
      for (int sft = 1; sft < IA_REFILL_GROUP; sft *= 2) need &= need >> sft;
      need &= IA_REFILL_GROUP_HEADS;
#pragma unroll
      for (int sft = 1; sft < IA_REFILL_GROUP; sft *= 2) need |= need << sft;
      if (need) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_next, __popcll(need));
        base = __shfl(base, 0, 64);
        if (base >= n_live) queue_empty = true;
        const int my = base + __popcll(need & ((1ull << lane) - 1ull));
        if (!active && my < n_live) {
          item = s_list[my];
          const int init = item >> 7, pt = item & (NP - 1);
#if IA_SEARCH_T_LDS
          const float t0 = s_xd[pt][0], t1 = s_xd[pt][1], t2 = s_xd[pt][2];
#else
          t0 = s_xd[pt][0]; t1 = s_xd[pt][1]; t2 = s_xd[pt][2];
#endif
          const float *T = s_T[init];
          // :287-293  x0 = R^T (xd - t)
          const float ixd = t0 - T[3], iyd = t1 - T[7], izd = t2 - T[11];
          xl0 = IA_DOT3(ixd, T[0], iyd, T[4], izd, T[8]);
          xl1 = IA_DOT3(ixd, T[1], iyd, T[5], izd, T[9]);
          xl2 = IA_DOT3(ixd, T[2], iyd, T[6], izd, T[10]);
          active = true; first = true; it_solves = (it_solves & ~0xFFu) + 0x100u;
        }
      }
    }
    if (!__any(active)) break;
    const float ix = g.scl[0] * (xl0 + g.off[0]);
    const float iy = g.scl[1] * (xl1 + g.off[1]);
    const float iz = g.scl[2] * (xl2 + g.off[2]);
    float Jl[12];
    bool ld = false;
#if IA_SEARCH_QUAD
    fetch_J_quad(vJ, g, ix, iy, iz, active, Jl, ld, s_del);   // all lanes: the quad serves its four fetches together
#else
    if (active) ld = fetch_J(vJ, g, ix, iy, iz, Jl);
#endif
    if (active) {
      counts += ld ? 0x10001u : 1u;
      bool done = false, ok = false;
#if IA_SEARCH_T_LDS
      const float *txd = s_xd[item & (NP - 1)];
      const float t0 = txd[0], t1 = txd[1], t2 = txd[2];
#endif
      // residual g(x) = J x + d - x_d at the current point (:325-332 initial, :356-367 updated)
      const float n0 = IA_DOT3(Jl[0], xl0, Jl[1], xl1, Jl[2], xl2) + Jl[3] - t0;
      const float n1 = IA_DOT3(Jl[4], xl0, Jl[5], xl1, Jl[6], xl2) + Jl[7] - t1;
      const float n2 = IA_DOT3(Jl[8], xl0, Jl[9], xl1, Jl[10], xl2) + Jl[11] - t2;
      if (first) {
        // :302-311 J_inv0 = (J_3x3)^T
        Ji[0] = Jl[0]; Ji[1] = Jl[4]; Ji[2] = Jl[8]; Ji[3] = Jl[1]; Ji[4] = Jl[5]; Ji[5] = Jl[9];
        Ji[6] = Jl[2]; Ji[7] = Jl[6]; Ji[8] = Jl[10];
        gx0 = n0; gx1 = n1; gx2 = n2;
        first = false;
      } else {
        // :368-398 convergence / divergence tests
        const float norm = IA_DOT3(n0, n0, n1, n1, n2, n2);
        if (norm < cvg2) {
          done = true;
          ok = ix >= -1 && ix <= 1 && iy >= -1 && iy <= 1 && iz >= -1 && iz <= 1;
        } else if (norm > dvg2) {
          done = true;
        } else {
          jinv_update(Ji, u0, u1, u2, n0 - gx0, n1 - gx1, n2 - gx2);  // :400-411
          gx0 = n0; gx1 = n1; gx2 = n2;
          if ((++it_solves & 0xFFu) == 10u) done = true;  // Q1: not converged after 10 iterations -> invalid
        }
      }
      if (done) {
        const int init = item >> 7, pt = item & (NP - 1);
        s_x[init][pt][0] = ok ? xl0 : 0.f; s_x[init][pt][1] = ok ? xl1 : 0.f; s_x[init][pt][2] = ok ? xl2 : 0.f;
        s_valid[init][pt] = ok;
        if (MODE == 0 && J_inv) {
          // Q4: the stored J_inv is the matrix BEFORE the last rank-1 update (:383-391)
          const size_t o = ((size_t)(p0 + pt) * n_init + init) * 9;
#pragma unroll
          for (int k = 0; k < 9; k++) J_inv[o + k] = ok ? Ji[k] : 0.f;
        }
        if (MODE == 2 && ok) {
#pragma unroll
          for (int k = 0; k < 9; k++) jinv_dense[((size_t)(p0 + pt) * n_init + init) * 9 + k] = Ji[k];  // Q4 as above
        }
        active = false;
      } else {
        // :340-351 update = -J_inv g ; x += update (start of the next iteration)
        u0 = IA_DOT3(-Ji[0], gx0, -Ji[1], gx1, -Ji[2], gx2);
        u1 = IA_DOT3(-Ji[3], gx0, -Ji[4], gx1, -Ji[5], gx2);
        u2 = IA_DOT3(-Ji[6], gx0, -Ji[7], gx1, -Ji[8], gx2);
        xl0 += u0; xl1 += u1; xl2 += u2;
      }
    }
  }
#endif
  if (prof) {  // bench-only accounting: solves and trilinear fetches
#if IA_SEARCH_QPS
    int f = fetches, n = solves, l = loaded;
#else
    int f = (int)(counts & 0xFFFFu), n = (int)(it_solves >> 8), l = (int)(counts >> 16);
#endif
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { f += __shfl_xor(f, o, 64); n += __shfl_xor(n, o, 64); l += __shfl_xor(l, o, 64); }
    if (lane == 0) { atomicAdd(&s_prof[0], n); atomicAdd(&s_prof[1], f); atomicAdd(&s_prof[2], l); }
  }
  __syncthreads();
  if (prof && tid == 0) {  // one pair of global atomics per workgroup, on a per-shard line
    unsigned long long *ps = prof + (size_t)(blockIdx.x & (IA_PROF_SHARDS - 1)) * 8;
    atomicAdd(ps, (unsigned long long)s_prof[0]);
    atomicAdd(ps + 1, (unsigned long long)s_prof[1]);
    atomicAdd(ps + 2, (unsigned long long)s_prof[2]);
  }
  // ---- a5 filter (filter.cu:27-51): drop i if a LATER valid candidate lies within 1e-4 ----
  for (int init0 = 0; init0 < n_init; init0 += IA_SEARCH_THREADS / NP) {
    const int init = init0 + tid / NP, pt = tid & (NP - 1);
    if (init >= n_init || pt >= np) continue;
    bool keep = s_valid[init][pt];
    if (keep) {
      const float x0 = s_x[init][pt][0], x1 = s_x[init][pt][1], x2 = s_x[init][pt][2];
      for (int j = init + 1; j < n_init; j++) {
        if (!s_valid[j][pt]) continue;
        const float d0 = x0 - s_x[j][pt][0], d1 = x1 - s_x[j][pt][1], d2 = x2 - s_x[j][pt][2];
        const float dist = IA_DOT3(d0, d0, d1, d1, d2, d2);
        if ((double)dist < 0.0001 * 0.0001) { keep = false; break; }
      }
    }
    s_keep[init][pt] = keep;
    if (MODE == 0) {
      const size_t o = (size_t)(p0 + pt) * n_init + init;
      xc[o * 3] = s_x[init][pt][0]; xc[o * 3 + 1] = s_x[init][pt][1]; xc[o * 3 + 2] = s_x[init][pt][2];
      valid_out[o] = keep;
      if (valid_raw) valid_raw[o] = s_valid[init][pt];
    }
  }
  if (MODE == 0) return;
  __syncthreads();
  // ---- compaction: per-point counts, workgroup scan, ONE global atomic ----
  int cnt = 0;
  if (tid < np)
    for (int j = 0; j < n_init; j++) cnt += s_keep[j][tid];
  int wtot;
  const int excl = ia_wave_excl_scan(cnt, wtot);
  if (lane == 0) s_wtot[tid >> 6] = wtot;
  __syncthreads();
  if (tid == 0) {
    int tot = 0;
    for (int w = 0; w < IA_SEARCH_THREADS / 64; w++) { const int c = s_wtot[w]; s_wtot[w] = tot; tot += c; }
    // (one same-address atomic per workgroup; r02: spreading it over 8 / 64 counters 128 bytes apart changes nothing --
    // 246.6 / 246.6 / 246.2 us on the 213 k sample points of a frame, tools/bench_search.py -- while a SECOND dependent
    // atomic per workgroup doubles the launch time: its latency sits on every workgroup's critical path)
    s_blockbase = tot > 0 ? atomicAdd(n_cand, tot) : 0;
  }
  __syncthreads();
  if (tid < np) {
    const int b = s_blockbase + s_wtot[tid >> 6] + excl;
    s_base[tid] = b;
    pt_off[p0 + tid] = b;
    pt_cnt[p0 + tid] = (uint8_t)cnt;
  }
  __syncthreads();
  for (int init0 = 0; init0 < n_init; init0 += IA_SEARCH_THREADS / NP) {
    const int init = init0 + tid / NP, pt = tid & (NP - 1);
    if (init >= n_init || pt >= np || !s_keep[init][pt]) continue;
    int rank = 0;
    for (int j = 0; j < init; j++) rank += s_keep[j][pt];
    const int o = s_base[pt] + rank;
    if (o < cand_cap) {
      cand_xc[(size_t)o * 3] = s_x[init][pt][0]; cand_xc[(size_t)o * 3 + 1] = s_x[init][pt][1];
      cand_xc[(size_t)o * 3 + 2] = s_x[init][pt][2];
      if (MODE == 2) {
#pragma unroll
        for (int k = 0; k < 9; k++) cand_Jinv[(size_t)o * 9 + k] = jinv_dense[((size_t)(p0 + pt) * n_init + init) * 9 + k];
      }
    }
  }
}

__global__ void k_zero_i32(int32_t *p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0;
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
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_precompute<IA_PRE_VPT>), dim3(blocks), dim3(256), 0, s, voxel_w, tfs, voxel_J, voxel_d, bbox,
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


extern "C" int ia_snarf_search(const float *xd, int P, const float *voxel_J, const float *tfs,
                               const int32_t *bone_ids, int n_init, const ia_snarf_grid *grid,
                               float cvg_thresh, float dvg_thresh, float *xc, uint8_t *valid,
                               uint8_t *valid_raw, float *J_inv, void *stream) {
  IA_CHECK_ARG(P >= 0, "ia_snarf_search: P < 0");
  if (P == 0) return IA_OK;
  IA_CHECK_ARG(xd && voxel_J && tfs && grid && xc && valid, "ia_snarf_search: null pointer");
  BoneIds b;
  IA_CHECK_ARG(ia_make_bones(bone_ids, n_init, &b) == 0, "ia_snarf_search: bad bone ids / n_init=%d", n_init);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_search<0>), dim3(ia_div_up(P, IA_SEARCH_NP)), dim3(IA_SEARCH_THREADS), 0,
                     (hipStream_t)stream, xd, P, (const int32_t *)nullptr, voxel_J, tfs, b, n_init,
                     ia_make_grid_dev(grid), cvg_thresh * cvg_thresh, dvg_thresh * dvg_thresh, xc, valid,
                     valid_raw, J_inv, (float *)nullptr, 0, (int32_t *)nullptr, (uint8_t *)nullptr,
                     (int32_t *)nullptr, (unsigned long long *)nullptr, (float *)nullptr, (float *)nullptr);
  IA_LAUNCH_CHECK("k_search<0>");
  return IA_OK;
}

static int ia_search_compact_impl(const char *who, const float *xd, int P, const int32_t *n_pts_dev, const float *voxel_J,
                                  const float *tfs, const int32_t *bone_ids, int n_init, const ia_snarf_grid *grid,
                                  float cvg_thresh, float dvg_thresh, float *cand_xc, float *cand_Jinv, int32_t cand_cap,
                                  int32_t *pt_off, uint8_t *pt_cnt, int32_t *n_cand, int zero_counter, bool with_jinv,
                                  hipStream_t s, float *jinv_dense = nullptr) {
  IA_CHECK_ARG(P >= 0, "%s: P < 0", who);
  IA_CHECK_ARG(n_cand, "%s: n_cand is null", who);
  if (zero_counter) { hipLaunchKernelGGL(k_zero_i32, dim3(1), dim3(64), 0, s, n_cand, 1); IA_LAUNCH_CHECK("k_zero_i32"); }
  if (P == 0) return IA_OK;
  IA_CHECK_ARG(xd && voxel_J && tfs && grid && cand_xc && pt_off && pt_cnt && (cand_Jinv || !with_jinv), "%s: null pointer", who);
  BoneIds b;
  IA_CHECK_ARG(ia_make_bones(bone_ids, n_init, &b) == 0, "%s: bad bone ids / n_init=%d", who, n_init);
  const dim3 grd(ia_div_up(P, IA_SEARCH_NP)), blk(IA_SEARCH_THREADS);
  const SnarfGridDev g = ia_make_grid_dev(grid);
  ia_prof_begin(IA_PROF_SEARCH, s);
  if (with_jinv)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_search<2>), grd, blk, 0, s, xd, P, n_pts_dev, voxel_J, tfs, b, n_init, g,
                       cvg_thresh * cvg_thresh, dvg_thresh * dvg_thresh, (float *)nullptr, (uint8_t *)nullptr,
                       (uint8_t *)nullptr, (float *)nullptr, cand_xc, cand_cap, pt_off, pt_cnt, n_cand,
                       ia_prof_units(IA_PROF_SEARCH), cand_Jinv, jinv_dense);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_search<1>), grd, blk, 0, s, xd, P, n_pts_dev, voxel_J, tfs, b, n_init, g,
                       cvg_thresh * cvg_thresh, dvg_thresh * dvg_thresh, (float *)nullptr, (uint8_t *)nullptr,
                       (uint8_t *)nullptr, (float *)nullptr, cand_xc, cand_cap, pt_off, pt_cnt, n_cand,
                       ia_prof_units(IA_PROF_SEARCH), (float *)nullptr, (float *)nullptr);
  ia_prof_end(IA_PROF_SEARCH, s);
  IA_LAUNCH_CHECK("k_search<compact>");
  return IA_OK;
}

extern "C" int ia_snarf_search_compact(const float *xd, int P, const int32_t *n_pts_dev,
                                       const float *voxel_J, const float *tfs, const int32_t *bone_ids,
                                       int n_init, const ia_snarf_grid *grid, float cvg_thresh,
                                       float dvg_thresh, float *cand_xc, int32_t cand_cap, int32_t *pt_off,
                                       uint8_t *pt_cnt, int32_t *n_cand, int zero_counter, void *stream) {
  return ia_search_compact_impl("ia_snarf_search_compact", xd, P, n_pts_dev, voxel_J, tfs, bone_ids, n_init, grid, cvg_thresh,
                                dvg_thresh, cand_xc, nullptr, cand_cap, pt_off, pt_cnt, n_cand, zero_counter, false,
                                (hipStream_t)stream);
}

extern "C" size_t ia_snarf_search_jinv_workspace_bytes(int P, int n_init) {
  return P > 0 && n_init > 0 ? (size_t)P * (size_t)n_init * 9 * sizeof(float) : 0;
}

extern "C" int ia_snarf_search_compact_jinv(const float *xd, int P, const int32_t *n_pts_dev,
                                            const float *voxel_J, const float *tfs, const int32_t *bone_ids,
                                            int n_init, const ia_snarf_grid *grid, float cvg_thresh,
                                            float dvg_thresh, float *cand_xc, float *cand_Jinv, int32_t cand_cap,
                                            int32_t *pt_off, uint8_t *pt_cnt, int32_t *n_cand, int zero_counter,
                                            void *ws, size_t ws_bytes, void *stream) {
  IA_CHECK_ARG(P <= 0 || (ws && ws_bytes >= ia_snarf_search_jinv_workspace_bytes(P, n_init)),
               "ia_snarf_search_compact_jinv: workspace of %zu bytes, %zu needed", ws_bytes, ia_snarf_search_jinv_workspace_bytes(P, n_init));
  return ia_search_compact_impl("ia_snarf_search_compact_jinv", xd, P, n_pts_dev, voxel_J, tfs, bone_ids, n_init, grid,
                                cvg_thresh, dvg_thresh, cand_xc, cand_Jinv, cand_cap, pt_off, pt_cnt, n_cand, zero_counter,
                                true, (hipStream_t)stream, static_cast<float *>(ws));
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
template <bool CL>
__global__ __launch_bounds__(IA_ID_THREADS) void k_implicit_bwd(
    const float *__restrict__ xc, const float *__restrict__ J_inv, const uint8_t *__restrict__ valid,
    const float *__restrict__ grad, long n, const int32_t *__restrict__ n_dev, const float *__restrict__ voxel_w,
    SnarfGridDev g, float *__restrict__ partial) {
  if (n_dev) n = min(n, (long)*n_dev);   // compact candidate lists: device-side live count, no validity mask
  __shared__ float s_w[IA_ID_THREADS][25];   // +1: the 24-float rows start in different banks
  __shared__ float s_vh[IA_ID_THREADS][13];
  const int tid = threadIdx.x;
  const long vol = (long)g.D * g.H * g.W;
  float acc0 = 0.f, acc1 = 0.f;              // outputs tid and tid + 256 of the 288 (bone, row, col) sums
  for (long base = (long)blockIdx.x * IA_ID_THREADS; base < n; base += (long)gridDim.x * IA_ID_THREADS) {
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
      const float *Ji = J_inv + i * 9;
      const float v[3] = {-(Ji[0] * g0 + Ji[3] * g1 + Ji[6] * g2), -(Ji[1] * g0 + Ji[4] * g1 + Ji[7] * g2),
                          -(Ji[2] * g0 + Ji[5] * g1 + Ji[8] * g2)};
      const float h[4] = {x0, x1, x2, 1.f};
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
        for (int k = 0; k < 4; k++) vh[c * 4 + k] = v[c] * h[k];
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

static int ia_implicit_bwd_impl(const char *who, const float *xc, const float *J_inv, const uint8_t *valid,
                                const float *grad_xc, long n, const int32_t *n_dev, const float *voxel_w,
                                const ia_snarf_grid *grid, float *d_tfs, void *ws, size_t ws_bytes, hipStream_t s,
                                bool channel_last = false) {
  IA_CHECK_ARG(n >= 0, "%s: n < 0", who);
  if (n == 0) return IA_OK;
  IA_CHECK_ARG(xc && J_inv && (valid || n_dev) && grad_xc && voxel_w && grid && d_tfs && ws, "%s: null pointer", who);
  if (ws_bytes < ia_snarf_implicit_bwd_workspace_bytes(n)) return ia_set_error(IA_ERR_WORKSPACE, "%s: workspace too small", who);
  const int blocks = ia_implicit_blocks(n);
  if (channel_last)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_implicit_bwd<true>), dim3(blocks), dim3(IA_ID_THREADS), 0, s, xc, J_inv, valid, grad_xc, n, n_dev,
                       voxel_w, ia_make_grid_dev(grid), static_cast<float *>(ws));
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_implicit_bwd<false>), dim3(blocks), dim3(IA_ID_THREADS), 0, s, xc, J_inv, valid, grad_xc, n, n_dev,
                       voxel_w, ia_make_grid_dev(grid), static_cast<float *>(ws));
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

// ---- device self-tests of the shared-reciprocal division (called by tests/ only; they launch the SAME device functions
// k_search uses) ----
__global__ void k_selftest_shared_rcp(const float *__restrict__ num, const float *__restrict__ den, int n,
                                      float *__restrict__ q_shared, float *__restrict__ q_ieee) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = num[i], b = den[i];
  q_shared[i] = div_shared(a, b, rcp_refined(b));
  q_ieee[i] = a / b;
}
__global__ void k_selftest_jinv_update(const float *__restrict__ Ji, const float *__restrict__ x, const float *__restrict__ g, int n,
                                       float *__restrict__ out_shared, float *__restrict__ out_plain, uint8_t *__restrict__ took_shared) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // n is a multiple of the wave size: whole waves run the update
  if (i >= n) return;
  float A[9], B[9];
#pragma unroll
  for (int k = 0; k < 9; k++) { A[k] = Ji[(size_t)i * 9 + k]; B[k] = A[k]; }
  const bool sh = jinv_update_impl<true>(A, x[i * 3], x[i * 3 + 1], x[i * 3 + 2], g[i * 3], g[i * 3 + 1], g[i * 3 + 2]);
  (void)jinv_update_impl<false>(B, x[i * 3], x[i * 3 + 1], x[i * 3 + 2], g[i * 3], g[i * 3 + 1], g[i * 3 + 2]);
#pragma unroll
  for (int k = 0; k < 9; k++) { out_shared[(size_t)i * 9 + k] = A[k]; out_plain[(size_t)i * 9 + k] = B[k]; }
  took_shared[i] = sh;
}
// q_shared[i] = the shared-reciprocal quotient num[i] / den[i], q_ieee[i] = the compiler's division (device pointers)
extern "C" int ia_selftest_shared_rcp(const float *num, const float *den, int n, float *q_shared, float *q_ieee, void *stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  IA_CHECK_ARG(n >= 0 && (n == 0 || (num && den && q_shared && q_ieee)), "ia_selftest_shared_rcp: bad arguments");
  if (n == 0) return IA_OK;
  hipLaunchKernelGGL(k_selftest_shared_rcp, dim3((n + 255) / 256), dim3(256), 0, stream, num, den, n, q_shared, q_ieee);
  IA_LAUNCH_CHECK("ia_selftest_shared_rcp");
  return IA_OK;
}
// the Broyden rank-1 update of n J_inv matrices [n][9] with steps x [n][3] and residual differences g [n][3], once with the
// shared reciprocal (as k_search runs it: per wave, only when all 64 lanes are in range -> took_shared[i]) and once with the
// compiler's divisions; n must be a multiple of 64
extern "C" int ia_selftest_jinv_update(const float *Ji, const float *x, const float *g, int n, float *out_shared, float *out_plain,
                                       uint8_t *took_shared, void *stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  IA_CHECK_ARG(n >= 0 && n % 64 == 0 && (n == 0 || (Ji && x && g && out_shared && out_plain && took_shared)), "ia_selftest_jinv_update: bad arguments (n % 64 == 0)");
  if (n == 0) return IA_OK;
  hipLaunchKernelGGL(k_selftest_jinv_update, dim3(n / 256 + 1), dim3(256), 0, stream, Ji, x, g, n, out_shared, out_plain, took_shared);
  IA_LAUNCH_CHECK("ia_selftest_jinv_update");
  return IA_OK;
}

// Resource usage of the search kernel as compiled into THIS library (bench.py reports it next to the counters instead
// of quoting numbers from a build log): VGPRs per lane, static LDS per workgroup, threads per workgroup and the
// resident workgroups per CU the runtime computes from them.
extern "C" int ia_search_kernel_info(int *vgprs, int *lds_bytes, int *threads, int *workgroups_per_cu) {
  hipFuncAttributes a;
  const void *fn = reinterpret_cast<const void *>(&k_search<1>);
  if (hipFuncGetAttributes(&a, fn) != hipSuccess) return ia_set_error(IA_ERR_LAUNCH, "ia_search_kernel_info: hipFuncGetAttributes failed");
  int nb = 0;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, IA_SEARCH_THREADS, 0);
  if (vgprs) *vgprs = a.numRegs;
  if (lds_bytes) *lds_bytes = (int)a.sharedSizeBytes;
  if (threads) *threads = IA_SEARCH_THREADS;
  if (workgroups_per_cu) *workgroups_per_cu = nb;
  return IA_OK;
}
