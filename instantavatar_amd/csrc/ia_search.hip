// ia_search.hip -- a4 + a5: Broyden root finding of the Fast-SNARF deformer, duplicate filter, compaction (gfx950, wave64).
//
//   k_search  a workgroup owns 64 points x n_init solves as an LDS queue; every lane runs a state machine whose loop body is
//             ONE quad-cooperative trilinear fetch + the Broyden update (ia_search_dev.h); finished lanes pull the next item;
//             then the workgroup runs the duplicate filter and either writes the dense reference layout or compacts the roots.
//
// Reference semantics: fuse_kernel/fuse_cuda_kernel_fast.cu:252-413 (broyden_kernel; every solve executes exactly its
// arithmetic sequence), filter/filter.cu:10-55, deformer_torch.py:85-116.
//
// What bounds it (round 4; profiles/r04_search_phase_cycles.txt, r04_ab_search_occupancy_cellcache.txt): the
// solver loop is a dependent chain -- fetch plan, three load round trips, row delivery through LDS, the Broyden update -- whose
// latency the four waves per SIMD that 106 VGPRs allow hide only in part (3 / 2 workgroups per CU: 205 / 253 us against 191).  A
// wave-step moves 24.6 KB through the CU's vector L1 and issues 380-480 VALU instructions; at one wave-step per ~575 CU-cycles
// both pipes are two-thirds busy and neither is the limit (a register cell cache that removed 35 % of the loads gained 1.4 %
// and lost 31 % through its registers).  A persistent-wave rewrite (no workgroup barrier, global work supply, lanes packed 15 %
// tighter, bit-identical) lost by 12 %: more waves inside the loop only lengthen every round trip.  All archived with their
// numbers under tools/variants/.
#include "ia_search_dev.h"

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
#define IA_SEARCH_NP 64        // points per workgroup (threads x points, round 3): 256 x 64 199.6 us, 128 x 32 206.7, 512 x 128 206.7,
#endif
#define IA_SEARCH_THREADS 256  // 256 x 128 212.1, 128 x 64 215.3, 64 x 32 223.0, 256 x 32 228.6, 512 x 64 229.2
#define IA_SEARCH_ATTR

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
  __shared__ float4 s_del_store[IA_SEARCH_THREADS * 3];
  float4 *const s_del = s_del_store;
  if (n_pts_dev) P = min(P, *n_pts_dev);
  const int tid = threadIdx.x, lane = tid & 63;
  // (an XCD-aware remap -- XCD x takes the x-th contiguous eighth of the point list -- was measured:
  // the occupancy probes got 25 % slower, the slabs at the rim of the bounding box hold little live work;
  // round 6, on the render launches only: 612-613 frames/s with and without it)
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

  // ---- lane state machine ----
  bool active = false, first = false;
  // `solves` counts the queued (non-trivial) ones; `fetches` every trilinear fetch of the reference's algorithm, `loaded` those
  // that touched memory (a fetch with all 8 corners outside the grid is zero by construction and loads nothing)
  // (packed: registers are what bounds the waves per SIMD here -- `counts` = fetches | loaded << 16, a lane sees at most
  // 13 * NP * 11 < 2^16 fetches per launch; `it_solves` = iter | solves << 8; IA_SEARCH_T_LDS: the target x_d re-read from LDS)
  int item = 0;
  uint32_t counts = 0, it_solves = 0;
  float t0 = 0, t1 = 0, t2 = 0;
  float xl0 = 0, xl1 = 0, xl2 = 0, gx0 = 0, gx1 = 0, gx2 = 0, u0 = 0, u1 = 0, u2 = 0;
  float Ji[9];
#pragma unroll
  for (int k = 0; k < 9; k++) Ji[k] = 0.f;
  bool queue_empty = false;
  while (true) {
    if (!queue_empty) {
      const unsigned long long need = __ballot(!active);
      if (need) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_next, __popcll(need));
        base = __shfl(base, 0, 64);
        if (base >= n_live) queue_empty = true;
        const int my = base + __popcll(need & ((1ull << lane) - 1ull));
        if (!active && my < n_live) {
          item = s_list[my];
          const int init = item >> 7, pt = item & (NP - 1);
          t0 = s_xd[pt][0]; t1 = s_xd[pt][1]; t2 = s_xd[pt][2];
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
    fetch_J_quad(vJ, g, ix, iy, iz, active, Jl, ld, s_del);   // all lanes: the quad serves its four fetches together
    if (active) {
      counts += ld ? 0x10001u : 1u;
      bool done = false, ok = false;
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
        if (ok) it_solves |= 0x80000000u;   // this lane found a root (bit 31: `solves` stays below 2^16)
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
  if (prof) {  // bench-only accounting: solves and trilinear fetches
    int f = (int)(counts & 0xFFFFu), n = (int)((it_solves >> 8) & 0xFFFFu), l = (int)(counts >> 16);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { f += __shfl_xor(f, o, 64); n += __shfl_xor(n, o, 64); l += __shfl_xor(l, o, 64); }
    if (lane == 0) { atomicAdd(&s_prof[0], n); atomicAdd(&s_prof[1], f); atomicAdd(&s_prof[2], l); }
  }
  const int any_root = __syncthreads_or((int)(it_solves >> 31));
  if (prof && tid == 0) {  // one pair of global atomics per workgroup, on a per-shard line
    unsigned long long *ps = prof + (size_t)(blockIdx.x & (IA_PROF_SHARDS - 1)) * 8;
    atomicAdd(ps, (unsigned long long)s_prof[0]);
    atomicAdd(ps + 1, (unsigned long long)s_prof[1]);
    atomicAdd(ps + 2, (unsigned long long)s_prof[2]);
  }
  // No root in the whole workgroup (most workgroups of the occupancy probes, whose points lie in empty space around the
  // body): nothing to filter, nothing to compact -- every point gets count 0 at offset 0, which is what the compaction
  // below writes for a workgroup without candidates (s_blockbase = 0, all prefix sums 0).  Four barriers and the filter's
  // and the compaction's LDS sweeps less; measured neutral on the launch time (profiles/r05_ab_probe.txt: those cycles are
  // waves waiting, not a resource the solver loops of the other waves compete for).
#ifndef IA_SEARCH_EARLY_OUT
#define IA_SEARCH_EARLY_OUT 1
#endif
  if (IA_SEARCH_EARLY_OUT && MODE != 0 && !any_root) {
    if (tid < np) {
      pt_off[p0 + tid] = 0;
      pt_cnt[p0 + tid] = 0;
    }
    return;
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
