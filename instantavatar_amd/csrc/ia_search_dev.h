// ia_search_dev.h -- device functions of the Broyden search (a4): the quad-cooperative trilinear fetch of the 12-channel
// transform grid and the rank-1 update of J_inv.  Reference semantics: fuse_cuda_kernel_fast.cu:23-55 (fuse_J_inv_update),
// :62-108 (grid_sampler_compute_source_index, zero padding, align_corners), :110-248 (the trilinear blend, corner order
// tnw,tne,tsw,tse,bnw,bne,bsw,bse :188-195).  Only the shipped path lives here; the variants that were measured and
// rejected in rounds 2-3 (lane-per-fetch, four-round DPP delivery, 2x12 round trips, half rounds, quad-per-solve, group
// refill) are archived with their numbers in tools/variants/.
#pragma once
#include "ia_common.h"

// ---- quad-cooperative trilinear fetch ------------------------------------------------------------------------------
// J is channel-last: a corner record is 3 x float4 = 48 contiguous bytes.  A lane fetching its own 8 corners would issue 24
// divergent 16-byte loads, each its own look-up in the CU's vector L1.  Here the four lanes of a quad serve their four
// fetches together: the quad's 12 (target lane, row) pairs are dealt to its FOUR lanes in THREE rounds -- round R, lane k
// serves pair 4R + k = (target (4R + k) / 3, row (4R + k) % 3): it learns the target's corner offsets and weights by DPP
// (per-lane constant sources: quad_perm [0,0,0,1], [1,1,2,2], [2,3,3,3]), loads ITS ROW of the 8 corner records (the lanes of
// a quad that serve the same target read one 48-byte record: 1-2 L1 segments instead of 3 look-ups) and accumulates it
// over the corners in the reference order -- every output element is the same fma chain as in a lane-per-fetch kernel, on
// another lane -- then stores the row as float4 number 4R + k of the quad's 12-float4 block in LDS, which is exactly where
// target lane t reads its rows 3t .. 3t + 2.  24 load instructions per step, three dependent round trips, no lane idles.
// All DPP traffic happens in wave-uniform control flow (a DPP read from a lane that EXEC has switched off returns nothing);
// only the loads are predicated.  Measured (round 3, compact search of a frame's 213 k sample points): lane-per-fetch 246 us,
// four DPP-delivered rounds 218.6, LDS delivery 212.4, this three-round deal 198.9.

// what a lane contributes to the rounds: BYTE offsets of the 8 corner records (clamped into the grid) and their weights
// (0 for corners outside, and for a lane that is not active) in the reference order, and whether anything is needed at all
struct FetchPlan {
  uint32_t off[8];
  float w[8];
  uint32_t load;   // 1: the lane is active and at least one corner lies inside the grid
};

// Validity is folded into the six 1-D interpolation factors -- one unsigned compare and one select per axis end instead of
// two compares per end, three-way ANDs and eight selects on the products (round 3's form).  A zeroed factor makes its four
// products exactly +0: the factors are finite and non-negative (src_index maps NaN / huge coordinates to -100), and the
// multiplication order of the products is unchanged, so the weights are bit-identical to select-after-multiply
// (test_broyden_search_and_filter, test_ref_pin.py; round 4: adopted, 196.3 -> 193.4 us together with the DPP add below).
__device__ __forceinline__ void fetch_plan(const SnarfGridDev &g, float gx, float gy, float gz, bool active, FetchPlan &p) {
  const float ix = src_index(gx, g.W), iy = src_index(gy, g.H), iz = src_index(gz, g.D);
  const int x0 = (int)floorf(ix), y0 = (int)floorf(iy), z0 = (int)floorf(iz);
  const int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
  const float fx1 = x1 - ix, fx0 = ix - x0, fy1 = y1 - iy, fy0 = iy - y0, fz1 = z1 - iz, fz0 = iz - z0;
  const bool bx0 = (uint32_t)x0 < (uint32_t)g.W, bx1 = (uint32_t)x1 < (uint32_t)g.W;
  const bool by0 = (uint32_t)y0 < (uint32_t)g.H, by1 = (uint32_t)y1 < (uint32_t)g.H;
  const bool bz0 = (uint32_t)z0 < (uint32_t)g.D, bz1 = (uint32_t)z1 < (uint32_t)g.D;
  // (reference weights :188-195: tnw = (x1-ix)(y1-iy)(z1-iz) belongs to corner (x0,y0,z0), and so on)
  const float qx0 = bx0 ? fx1 : 0.f, qx1 = bx1 ? fx0 : 0.f, qy0 = by0 ? fy1 : 0.f, qy1 = by1 ? fy0 : 0.f, qz0 = bz0 ? fz1 : 0.f, qz1 = bz1 ? fz0 : 0.f;
  const float wgt[8] = {qx0 * qy0 * qz0, qx1 * qy0 * qz0, qx0 * qy1 * qz0, qx1 * qy1 * qz0,
                        qx0 * qy0 * qz1, qx1 * qy0 * qz1, qx0 * qy1 * qz1, qx1 * qy1 * qz1};
  const int cx0 = min(max(x0, 0), g.W - 1), cx1 = min(max(x1, 0), g.W - 1);
  const int cy0 = min(max(y0, 0), g.H - 1), cy1 = min(max(y1, 0), g.H - 1);
  const int cz0 = min(max(z0, 0), g.D - 1), cz1 = min(max(z1, 0), g.D - 1);
  // byte offsets as sums of three per-axis terms: six 24-bit multiplies (full rate) and twelve adds -- the clamped indices
  // and the strides are far below 2^24
  const uint32_t sy = (uint32_t)g.W * 48u, sz = (uint32_t)(g.W * g.H) * 48u;
  const uint32_t xo[2] = {(uint32_t)__umul24((uint32_t)cx0, 48u), (uint32_t)__umul24((uint32_t)cx1, 48u)};
  const uint32_t yo[2] = {(uint32_t)__umul24((uint32_t)cy0, sy), (uint32_t)__umul24((uint32_t)cy1, sy)};
  const uint32_t zo[2] = {(uint32_t)__umul24((uint32_t)cz0, sz), (uint32_t)__umul24((uint32_t)cz1, sz)};
  const uint32_t zy[4] = {zo[0] + yo[0], zo[0] + yo[1], zo[1] + yo[0], zo[1] + yo[1]};
#pragma unroll
  for (int k = 0; k < 8; k++) {
    p.off[k] = zy[k >> 1] + xo[k & 1];
    p.w[k] = wgt[k];
  }
  p.load = (active && (bx0 || bx1) && (by0 || by1) && (bz0 || bz1)) ? 1u : 0u;
}

// (mov_dpp = update_dpp with an undefined `old` and bound_ctrl: one v_mov_b32_dpp, which the DPP combiner can fold into the
// VOP2 instruction that consumes it; all four lanes of a quad are always enabled where this is used)
template <int PERM> __device__ __forceinline__ uint32_t quad_perm(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, PERM, 0xF, 0xF, true);
}
template <int PERM> __device__ __forceinline__ float quad_perm(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), PERM, 0xF, 0xF, true));
}

// The offset broadcast and the row-offset add are ONE v_add_u32_dpp (inline asm: left to itself the compiler emits
// v_mov_b32_dpp + v_add_u32, because it sinks the add to the predicated loads).  The two wait states a DPP read needs after a
// VALU write of its source are the s_nop: inline asm is opaque to the hazard recogniser.
#ifndef IA_SEARCH_POL
#define IA_SEARCH_POL 0
#endif
template <int R>
__device__ __forceinline__ void fetch_round3(const char *__restrict__ vJb, const FetchPlan &p, float4 *__restrict__ s_quad_k) {
  constexpr int PERM = R == 0 ? 0x40 : (R == 1 ? 0xA5 : 0xFE);
  const uint32_t load = quad_perm<PERM>(p.load);
  if (__ballot(load != 0) == 0) return;
  const uint32_t koff = (uint32_t)(((threadIdx.x & 3) + R) % 3) * 16u;   // row (4R + k) % 3 = (k + R) % 3
  // (all DPP reads before the divergent part: a source lane that sits out this round must still be enabled when it is read)
  uint32_t off[8];
  float w[8];
  // ONE asm statement: the s_nop and the eight DPP adds cannot be separated -- volatile asm is ordered only against other
  // volatile asm, so with separate statements the compiler was free to sink a VALU producer of p.off[c] between the s_nop
  // and a DPP read (the build checked clean, nothing enforced it).  Early-clobber outputs: all eight are
  // written before the last input is read.
#define IA_DPP_ADD(o, i) "v_add_u32_dpp %" #o ", %" #i ", %16 quad_perm:[%17,%18,%19,%20] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
  asm volatile("s_nop 1\n\t" IA_DPP_ADD(0, 8) IA_DPP_ADD(1, 9) IA_DPP_ADD(2, 10) IA_DPP_ADD(3, 11) IA_DPP_ADD(4, 12) IA_DPP_ADD(5, 13)
               IA_DPP_ADD(6, 14) IA_DPP_ADD(7, 15)
               : "=&v"(off[0]), "=&v"(off[1]), "=&v"(off[2]), "=&v"(off[3]), "=&v"(off[4]), "=&v"(off[5]), "=&v"(off[6]), "=&v"(off[7])
               : "v"(p.off[0]), "v"(p.off[1]), "v"(p.off[2]), "v"(p.off[3]), "v"(p.off[4]), "v"(p.off[5]), "v"(p.off[6]), "v"(p.off[7]),
                 "v"(koff), "i"(PERM & 3), "i"((PERM >> 2) & 3), "i"((PERM >> 4) & 3), "i"((PERM >> 6) & 3));
#undef IA_DPP_ADD
#pragma unroll
  for (int c = 0; c < 8; c++) w[c] = quad_perm<PERM>(p.w[c]);
  if (load != 0) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    float4 v[8];
#pragma unroll
    for (int c = 0; c < 8; c++) {
#if IA_SEARCH_POL == 1   // nt: the record bypasses the CU's vector L1 (A/B switch, profiles/r06_ab_search_policy.txt)
      typedef float f4v __attribute__((ext_vector_type(4)));
      const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(vJb + (size_t)off[c]));
      v[c] = make_float4(t.x, t.y, t.z, t.w);
#else
      v[c] = *reinterpret_cast<const float4 *>(vJb + (size_t)off[c]);
#endif
    }
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

// the fetch of every lane of the wave (call in wave-uniform control flow); `loaded`: this lane's fetch touched memory.
// s_del: 3 float4 per lane of the workgroup (rows of the blended transform on their way to the lane that needs them).
// Corners outside the grid read a clamped in-range address with weight 0 -- fma(v, 0, acc) == acc for the finite table values,
// so the result equals the reference's "skip the corner" bit for bit; a fetch with all 8 corners outside loads nothing and
// reads zeros.
__device__ __forceinline__ void fetch_J_quad(const float *__restrict__ vJ, const SnarfGridDev &g, float gx, float gy, float gz,
                                             bool active, float *__restrict__ out, bool &loaded, float4 *__restrict__ s_del) {
  FetchPlan p;
  fetch_plan(g, gx, gy, gz, active, p);
  loaded = p.load != 0;
  const int k = threadIdx.x & 3;
  const char *vJb = reinterpret_cast<const char *>(vJ);
  if (p.load == 0) {   // nobody will write this lane's slot: an active lane with all corners outside reads zeros
    s_del[threadIdx.x * 3] = make_float4(0.f, 0.f, 0.f, 0.f); s_del[threadIdx.x * 3 + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
    s_del[threadIdx.x * 3 + 2] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __builtin_amdgcn_wave_barrier();
  {
    float4 *const s_quad_k = s_del + (threadIdx.x & ~3u) * 3 + k;
    fetch_round3<0>(vJb, p, s_quad_k);
    fetch_round3<1>(vJb, p, s_quad_k);
    fetch_round3<2>(vJb, p, s_quad_k);
  }
  __builtin_amdgcn_wave_barrier();
  {
    const float4 r0 = s_del[threadIdx.x * 3], r1 = s_del[threadIdx.x * 3 + 1], r2 = s_del[threadIdx.x * 3 + 2];
    out[0] = r0.x; out[1] = r0.y; out[2] = r0.z; out[3] = r0.w; out[4] = r1.x; out[5] = r1.y; out[6] = r1.z; out[7] = r1.w;
    out[8] = r2.x; out[9] = r2.y; out[10] = r2.z; out[11] = r2.w;
  }
  __builtin_amdgcn_wave_barrier();   // the next step's zero-fill / rows must not overtake these reads
}

// ---- nine IEEE divisions by the same denominator ----------------------------------------------------------------------
// The compiler expands a / b into v_div_scale x 2, v_rcp, a Newton chain (fma, fma on the reciprocal; mul, fma, fma, fma on
// the quotient), v_div_fmas, v_div_fixup.  v_div_scale only rescales its operands at the edges of the exponent range (a
// denormal or huge denominator, a numerator below 2^-103, a quotient near overflow / underflow -- CDNA3 ISA, V_DIV_SCALE_F32);
// away from those it returns them unchanged with VCC = 0, and v_div_fmas is then a plain fma.  In that range the reciprocal
// half of the chain depends on the denominator alone (`rcp_refined`, once per update) and the numerator half is the same five
// instructions the compiler emits (`div_shared`); v_div_fixup keeps the zero / inf / NaN cases (it does not look at the
// quotient for those), so the quotients are bit-identical to a / b.  `ia_selftest_shared_rcp` sweeps the exponent range on
// the device (tests/test_gpu_edge_cases.py).  Measured: 202.0 -> 199.6 us.
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
// the range, the compiler's divisions otherwise; returns which one ran.
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
  (void)jinv_update_impl<true>(Ji, x0, x1, x2, g0, g1, g2);
}
