/*
 * instantavatar_hip.h -- C ABI of libinstantavatar_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the volumetric-rendering hot path of InstantAvatar.
 * Every entry point replaces one pybind11/ATen op (or one torch-level fused
 * sequence) of the reference; the reference interface is cited per function
 * (paths relative to the reference tree).
 *
 * Conventions
 *  - All pointers are DEVICE pointers owned by the caller unless marked HOST.
 *  - `stream` is a hipStream_t passed as void*.  No call synchronises the
 *    device, none touches the default stream, none allocates device memory:
 *    scratch comes from the caller through `ws`/`ws_bytes` (query the size
 *    with the matching *_workspace_bytes function).
 *  - Return value: 0 = IA_OK, negative = error (ia_last_error() gives text).
 *  - All float data is fp32 unless a parameter says fp16 (IEEE binary16,
 *    passed as uint16_t*).
 *  - Batch is 1 frame (the reference kernels are only correct for B = 1:
 *    fast_snarf/cuda/filter/filter.cu:21-22, precompute.cu:56).
 */
#ifndef INSTANTAVATAR_HIP_H
#define INSTANTAVATAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IA_OK 0
#define IA_ERR_ARG (-1)
#define IA_ERR_LAUNCH (-2)
#define IA_ERR_WORKSPACE (-3)

#define IA_N_JOINTS 24      /* SMPL joints                                   */
#define IA_N_INIT_MAX 16    /* >= 13 init bones (deformer_torch.py:28)        */
#define IA_MAX_LEVELS 16    /* hash-grid levels (ngp.py:30-37)                */

/* ---- descriptors (HOST structs, passed by pointer, copied at call time) -- */

/* Fast-SNARF skinning voxel grid: deformer_torch.py:130-169
 * (offset_kernel = -centre, scale_kernel = 1/scale with z * ratio).          */
typedef struct ia_snarf_grid {
  int D, H, W;          /* 32,128,128 for resolution 128                      */
  float offset[3];      /* offset_kernel                                      */
  float scale[3];       /* scale_kernel                                       */
} ia_snarf_grid;

/* tcnn-v1.6 HashGrid level table (restated; see oracle/ia_oracle.c header).  */
typedef struct ia_hash_desc {
  int n_levels;                       /* 16                                   */
  float scale[IA_MAX_LEVELS];         /* exp2f(l*log2f(pls))*base - 1         */
  uint32_t res[IA_MAX_LEVELS];        /* ceilf(scale)+1                       */
  uint32_t offset[IA_MAX_LEVELS + 1]; /* entry offsets (x2 features each)     */
} ia_hash_desc;

/* NeRFNGPNet (models/networks/ngp.py:23-83): normalisation + weights.
 * table: fp16 [n_entries][2].  MLP weights fp16 row-major [out][in]:
 * sig_w1[64][32], sig_w2[16][64]; col_w1[64][16], col_w2[64][64],
 * col_w3[16][64] (tcnn FullyFusedMLP layout, no biases).                     */
typedef struct ia_field {
  float center[3];
  float scale[3];
  ia_hash_desc hash;
  const uint16_t *table;
  const uint16_t *sig_w1, *sig_w2;
  const uint16_t *col_w1, *col_w2, *col_w3;
  /* optional: MFMA weight-fragment image built by ia_field_prepare from the five
   * matrices above (ia_field_frags_bytes() bytes); NULL = built inside every
   * field kernel launch (slower).  Must be rebuilt whenever the weights change. */
  const uint16_t *mlp_frags;
  /* optional: scratch for the XCD-sharded encoding, n_levels planes of enc_ws_samples
   * packed half2 (4 B) each.  When non-NULL the field kernels encode a call of
   * V <= enc_ws_samples samples level-by-level, each hashed level pinned to one XCD's
   * L2 (same results, higher gather rate); NULL = single fused kernel.            */
  uint32_t *enc_ws;
  size_t enc_ws_samples;
  /* hint for the sharded encoding, 16-level tables: tiles of every four of its second level group (hashed 12-15 + dense 0-3) that
   * XCDs 0-3 take.  3 for spatially coherent samples (ray-ordered candidates of a frame or a training batch: XCDs 0-3 own the
   * cheap hashed levels 4-7 and finish them early), 2 for incoherent ones; 0 = 2.  Same results whatever the value.            */
  int32_t enc_split;
} ia_field;

/* Occupancy grid (models/structures/density_grid.py): G^3 cells over aabb.   */
typedef struct ia_occ_grid {
  int G;                 /* 64                                                */
  float aabb_min[3];
  float aabb_max[3];
} ia_occ_grid;

/* ---- library ------------------------------------------------------------- */
int ia_version(void);
const char *ia_last_error(void);
/* "<translation unit>=<16 hex digits>;..." -- a hash per translation unit over (compiler flags, shared headers,
 * source) of what this library was built from.  No counterpart in the reference (its extensions are JIT-built at
 * import, deformer_torch.py:10-19): bench.py uses it to accept a committed counter summary only for the very
 * kernel source it is running.                                                 */
const char *ia_source_manifest(void);

/* Level table as tcnn computes it on the host (grid.h: grid_scale /
 * grid_resolution / offset table; call site ngp.py:30-37).                   */
int ia_hash_desc_init(ia_hash_desc *out, int n_levels, int log2_hashmap_size,
                      int base_resolution, float per_level_scale);

/* ---- a1/a2: SMPL pose -> bone transforms --------------------------------
 * Replaces smplx lbs() joint chain (deformers/smplx/lbs.py:152-250,295-401),
 * transl folding (body_models.py:353-360) and SNARFDeformer.prepare_deformer
 * (snarf_deformer.py:71-93): tfs = inv(A[0]) . A . inv(A_rest).
 * joints_rest: [24,3] shaped rest joints (J_regressor . v_shaped), parents:
 * [24] int32, pose: [72] axis-angle, transl: [3], tfs_inv_t: [24,4,4]
 * = inv(A) of the canonical pose.  Outputs: tfs [24,4,4], w2s [4,4], A[24,4,4]
 * (A may be NULL).                                                           */
int ia_smpl_tfs(const float *joints_rest, const int32_t *parents,
                const float *pose, const float *transl, const float *tfs_inv_t,
                float *tfs, float *w2s, float *A, void *stream);

/* Backward of ia_smpl_tfs for SMPL-parameter optimisation (DNeRF.py:113-128 -> lbs.py under autograd in the
 * reference): d_tfs [24,4,4] (rows 0..2 are read) -> d_pose [72] and, optionally, d_transl [3] (analytically zero:
 * the bone transforms are expressed in the SMPL-root frame; what comes out is rounding noise, as in the reference).
 * The forward quantities are recomputed from the same inputs as ia_smpl_tfs.                                      */
int ia_smpl_tfs_bwd(const float *joints_rest, const int32_t *parents, const float *pose,
                    const float *transl, const float *tfs_inv_t, const float *d_tfs,
                    float *d_pose, float *d_transl, void *stream);

/* ---- SMPLDeformer (nearest-vertex deformer plugin) ------------------
 * Replaces SMPLDeformer.deform (deformers/smpl_deformer.py:86-110) incl. the pytorch3d
 * knn_points call (K = 1): for every point the nearest of the n_verts posed SMPL vertices
 * (verts [n_verts,3], SMPL-root frame), valid = dist^2 < threshold^2, pts_cano = T_inv[nearest]
 * applied to the point (T_inv [n_verts,4,4], smpl_deformer.py:66-75).  idx: optional.       */
int ia_smpl_nn_deform(const float *pts, int P, const int32_t *n_pts_dev, const float *verts,
                      const float *T_inv, int n_verts, float threshold, float *pts_cano,
                      uint8_t *valid, int32_t *idx, void *stream);
/* Fused deform_test / deform_train (smpl_deformer.py:112-131): the field is evaluated on the
 * valid points only; invalid points get sigma = fill (0 at test, -1e5 in training), rgb = 0;
 * nan_to_num != 0 zeroes non-finite field outputs.  rgb may be NULL.                        */
size_t ia_smpl_query_workspace_bytes(int P);
int ia_smpl_deform_query(const float *pts, int P, const int32_t *n_pts_dev, const float *verts,
                         const float *T_inv, int n_verts, float threshold, const ia_field *field,
                         float fill, int nan_to_num, float *rgb, float *sigma, void *ws,
                         size_t ws_bytes, const void *nn_grid, void *stream);
/* Vertex grid for the fused SMPLDeformer queries (`nn_grid` above and in ia_smpl_nn_compact; NULL = brute force over all
 * vertices).  The fused queries use a point's nearest vertex only when it is closer than `threshold` (smpl_deformer.py:102-104),
 * and every such vertex lies in the 3 x 3 x 3 cells around the point when the cells are at least `threshold` wide: the posed
 * vertices are binned once per frame (device-side counting sort; bounding box, dimensions and cell size are computed on the
 * device, no host read) and a point is tested against the vertices of its 27 cells -- the same winner (smallest distance,
 * lowest index among equals) for every valid point, nothing for the others.  grid: ia_smpl_nn_grid_bytes(n_verts) bytes,
 * valid for the `verts` it was built from.  ia_smpl_nn_deform (every point, exact index) stays brute force.               */
size_t ia_smpl_nn_grid_bytes(int n_verts);
int ia_smpl_nn_grid_build(const float *verts, int n_verts, float threshold, void *grid, size_t grid_bytes, void *stream);

/* ---- a20: skinning-weight voxelisation (one-time) ---------------------------
 * Replaces query_weights_smpl (fast_snarf/deformer_torch.py:225-244) including the
 * pytorch3d knn_points call (third_parties/pytorch3d/ops.py:123): for every voxel
 * centre pts[i] (i in (d,h,w) raster order) the 30 nearest of the n_verts SMPL
 * vertices, inverse-distance blend of their weights [n_verts,24], then n_smooth
 * (30) 6-neighbour smoothing + renormalisation passes.  voxel_w: OUT [24,d,h,w]. */
size_t ia_voxelise_workspace_bytes(int d, int h, int w);
int ia_voxelise_weights(const float *pts, const float *verts, int n_verts,
                        const float *vert_weights, int d, int h, int w, int n_smooth,
                        float *voxel_w, void *ws, size_t ws_bytes, void *stream);

/* ---- a3: precompute -------------------------------------------------------
 * Replaces precompute(voxel_w, tfs, voxel_d, voxel_J, offset, scale)
 * (fast_snarf/cuda/precompute/precompute.cpp:7-13, precompute.cu:24-71).
 * voxel_w: [24,D,H,W].  voxel_J: OUT, channel-LAST [D,H,W,12] (native layout:
 * one trilinear corner = 48 contiguous bytes).  voxel_d: OUT [3,D,H,W] or
 * NULL.  bbox: OUT [6] = min xyz, max xyz of voxel_d (what
 * SNARFDeformer.get_bbox_deformed, snarf_deformer.py:105-107, reduces) or
 * NULL.  No device synchronise (the reference's cudaDeviceSynchronize at
 * precompute.cu:102 is dropped).                                             */
int ia_precompute(const float *voxel_w, const float *tfs, float *voxel_J,
                  float *voxel_d, float *bbox, const ia_snarf_grid *grid,
                  void *stream);
/* Same, with a scratch buffer of ia_precompute_workspace_bytes(grid) bytes (24 B
 * per workgroup): the bounding box is reduced through per-workgroup extrema and
 * one small second launch instead of float atomics on six addresses, which
 * serialise in a single memory channel (measured on MI355X, 32x128x128 grid:
 * 93 us -> 22 us).  Concurrent calls on different streams need their own ws.
 * ws == NULL behaves like ia_precompute.                                       */
size_t ia_precompute_workspace_bytes(const ia_snarf_grid *grid);
int ia_precompute_ws(const float *voxel_w, const float *tfs, float *voxel_J,
                     float *voxel_d, float *bbox, const ia_snarf_grid *grid,
                     void *ws, size_t ws_bytes, void *stream);

/* ---- a4 + a5: Broyden search + duplicate filter ----------------------------
 * Replaces fuse_broyden(...) + filter(x, mask)
 * (fast_snarf/cuda/fuse_kernel/fuse_cuda.cpp:14-28,
 *  fuse_cuda_kernel_fast.cu:252-413, filter/filter.cpp:12-21, filter.cu:10-55)
 * as called by ForwardDeformer.broyden_cuda (deformer_torch.py:100-116).
 * xd: [P,3].  bone_ids: HOST int[n_init].  Outputs (fully written, caller need
 * not zero them): xc [P,n_init,3] (0 where not converged&valid),
 * valid [P,n_init] uint8 AFTER the duplicate filter, valid_raw [P,n_init] (before
 * the filter; may be NULL), J_inv [P,n_init,3,3] or NULL.                    */
int ia_snarf_search(const float *xd, int P, const float *voxel_J,
                    const float *tfs, const int32_t *bone_ids, int n_init,
                    const ia_snarf_grid *grid, float cvg_thresh,
                    float dvg_thresh, float *xc, uint8_t *valid,
                    uint8_t *valid_raw, float *J_inv, void *stream);

/* Fused form used by the fast path: same search + filter, but the surviving
 * candidates are compacted on device (wavefront ballot + prefix sum):
 *   cand_xc   [cap,3]  canonical positions of valid candidates
 *   pt_off    [P]      first candidate of point p   (int32)
 *   pt_cnt    [P]      number of valid candidates   (uint8, <= n_init)
 *   n_cand    [1]      total (int32, device; must be zeroed by the caller or
 *                      by zero_counter != 0)
 * n_pts_dev: optional DEVICE int32* holding the live point count (<= P); P is
 * then only the launch upper bound.  Candidates of one point are contiguous
 * and ordered by init index (snarf_deformer.py:139 takes the first maximum). */
int ia_snarf_search_compact(const float *xd, int P, const int32_t *n_pts_dev,
                            const float *voxel_J, const float *tfs,
                            const int32_t *bone_ids, int n_init,
                            const ia_snarf_grid *grid, float cvg_thresh,
                            float dvg_thresh, float *cand_xc, int32_t cand_cap,
                            int32_t *pt_off, uint8_t *pt_cnt, int32_t *n_cand,
                            int zero_counter, void *stream);
/* The same call for the training route with SMPL parameters under optimisation (DNeRF.py:113-128 ->
 * deformer_torch.py:50-67): additionally cand_Jinv [cap,3,3], the Broyden J_inv of every surviving root
 * (the matrix before the last rank-1 update, fuse_cuda_kernel_fast.cu:383-391), compacted exactly like
 * cand_xc -- what the reference gathers with `others['J_inv'][others['valid_ids']]` from a dense
 * [1,P,13,3,3] tensor.  Input of ia_snarf_implicit_bwd_compact.                                    */
size_t ia_snarf_search_jinv_workspace_bytes(int P, int n_init);   /* P x n_init x 9 floats: J_inv of the valid solves before compaction */
int ia_snarf_search_compact_jinv(const float *xd, int P, const int32_t *n_pts_dev,
                                 const float *voxel_J, const float *tfs,
                                 const int32_t *bone_ids, int n_init,
                                 const ia_snarf_grid *grid, float cvg_thresh,
                                 float dvg_thresh, float *cand_xc, float *cand_Jinv,
                                 int32_t cand_cap, int32_t *pt_off, uint8_t *pt_cnt,
                                 int32_t *n_cand, int zero_counter, void *ws, size_t ws_bytes,
                                 void *stream);

/* ---- a7: implicit differentiation of the roots ----------------------------------------
 * Backward of ForwardDeformer.forward's training branch (deformer_torch.py:50-67 with
 * forward_skinning :118-128 and query_weights :190-202): x_c <- x_c* - J_inv (d(x_c*) - sg[d(x_c*)])
 * has the value x_c* and the gradient dL/dtfs[n][c][k] = w_n(x_c*) v_c h_k, v = -J_inv^T dL/dx_c,
 * h = (x_c*, 1), w = trilinear sample (align_corners, border padding) of the skinning-weight volume
 * voxel_w [24,D,H,W].  xc, grad_xc: [n,3]; J_inv: [n,3,3]; valid: [n]; d_tfs [24,4,4] is ACCUMULATED
 * (rows 0..2).  Per-workgroup partial sums in ws, added in a fixed order.                       */
size_t ia_snarf_implicit_bwd_workspace_bytes(long n);
int ia_snarf_implicit_bwd(const float *xc, const float *J_inv, const uint8_t *valid,
                          const float *grad_xc, long n, const float *voxel_w,
                          const ia_snarf_grid *grid, float *d_tfs, void *ws, size_t ws_bytes,
                          void *stream);
/* The same backward over a COMPACT candidate list (ia_snarf_search_compact_jinv): no validity mask, the
 * first min(cap, *n_cand) rows are live (n_cand: device int32, no host read).  Workspace as above for n = cap. */
int ia_snarf_implicit_bwd_compact(const float *cand_xc, const float *cand_Jinv, const float *grad_xc,
                                  long cap, const int32_t *n_cand, const float *voxel_w,
                                  int channel_last /* voxel_w is [D,H,W,24] instead of [24,D,H,W] */,
                                  const ia_snarf_grid *grid, float *d_tfs, void *ws, size_t ws_bytes,
                                  void *stream);

/* ---- a7, `version: 2` of ForwardDeformer (deformer_torch.py:68-75, confs/deformer/fast_snarf_debug.yaml) ----------------
 * In training the reference replaces every valid root x_c* by the closed-form inverse skinning  x_c = R^T (x_d - t),
 * T = sum_n w_n(x_c*) tfs_n  (query_weights at the detached root, einsum "pn,nij->pij", `(pts - T[:, :3, 3]) @ T[:, :3, :3]`)
 * and differentiates it w.r.t. tfs.  Entries e = 0..n-1: root xc[e], target xd[pt] with pt = cand_pt[e] when cand_pt is
 * given (compact candidate lists) or e / n_init (the dense [P, n_init] layout); live entries: e < *n_dev and / or valid[e].
 *   ia_snarf_inverse_skinning      out [n,3] = the value (0 for entries that are not live)
 *   ia_snarf_inverse_skinning_bwd  d_tfs [24,4,4] += the gradient (rows 0..2); d_xd_entry [n,3] (optional) = R g of every entry, the
 *                                  gradient w.r.t. ITS target (sum the entries of a point: x_d reaches the SMPL parameters through
 *                                  the ray frame, snarf_deformer.py:95-103); ws as for ia_snarf_implicit_bwd
 *   ia_expand_candidate_points     cand_pt[pt_off[p] + j] = p, j < pt_cnt[p] (the point of every compact candidate)       */
int ia_snarf_inverse_skinning(const float *xc, const float *xd, const int32_t *cand_pt, int n_init,
                              const uint8_t *valid, long n, const int32_t *n_dev, const float *voxel_w,
                              int channel_last, const ia_snarf_grid *grid, const float *tfs, float *out,
                              void *stream);
int ia_snarf_inverse_skinning_bwd(const float *xc, const float *xd, const int32_t *cand_pt, int n_init,
                                  const uint8_t *valid, const float *grad_out, long n, const int32_t *n_dev,
                                  const float *voxel_w, int channel_last, const ia_snarf_grid *grid,
                                  const float *tfs, float *d_tfs, float *d_xd_entry, void *ws, size_t ws_bytes,
                                  void *stream);
int ia_expand_candidate_points(const int32_t *pt_off, const uint8_t *pt_cnt, int P, const int32_t *n_pts_dev,
                               int32_t *cand_pt, int cap, void *stream);

/* ---- a9 + a10 + a11: canonical field ---------------------------------------
 * Replaces NeRFNGPNet.forward (ngp.py:73-83) = tcnn NetworkWithInputEncoding
 * (HashGrid -> FullyFusedMLP 32-64-16) + tcnn Network (16-64-64-16, sigmoid).
 * x: [V,3] canonical points (un-normalised).  rgb [V,3], sigma [V] fp32.
 * n_dev: optional DEVICE int32* live count (V = upper bound).                */
int ia_field_fwd(const float *x, int V, const int32_t *n_dev,
                 const ia_field *field, float *rgb, float *sigma,
                 void *stream);
/* Training-mode forward (what tcnn does under autograd, ngp.py:78,81): same
 * outputs plus the fp16 activation record per sample, ia_field_act_stride()
 * halves: [features 2L | h1 64 | sigma-net out 16 | c1 64 | c2 64].          */
int ia_field_act_stride(int n_levels);
int ia_field_fwd_train(const float *x, int V, const int32_t *n_dev, const ia_field *field,
                       float *rgb, float *sigma, uint16_t *acts, void *stream);
/* NeRFLoss (instant_avatar/utils/loss.py:53-77), value and gradient in one pass:
 * out5 (zero-filled by the caller) = {loss, mse_loss, loss_alpha_coarse, reg_alpha, reg_density};
 * d_rgb [n_rays,3], d_alpha [n_rays], d_weight [n_weights] = d loss / d input.
 * weight: the dense weight_coarse tensor [n_rays x MAX_SAMPLES] (raymarcher_acc.py:181-186).
 * poison: optional device scalar; > 0 multiplies the loss and all gradients by NaN (the step of a training render that
 * dropped candidates is then skipped by the optimiser's non-finite check, the way GradScaler skips a step: DNeRF.py:151-154).
 * overflow_count / overflow_cap: optional device counter + capacity; *overflow_count > overflow_cap poisons the step the same
 * way (the render's candidate count against the capacity of its buffers, decided inside this kernel), and out5 then has a SIXTH
 * value: out5[5] = 1 if the step was poisoned by the counter, else 0.                                                          */
int ia_nerf_loss(const float *rgb, const float *tgt_rgb, const float *alpha,
                 const float *tgt_alpha, const float *weight, int n_rays, long long n_weights,
                 float w_rgb, float w_alpha, float w_reg, const float *poison, const int32_t *overflow_count,
                 int overflow_cap, float *out5, float *d_rgb, float *d_alpha, float *d_weight, void *stream);

/* Fused backward of both tiny MLPs (tcnn FullyFusedMLP backward; reached in the reference
 * through autograd of ngp.py:78,81).  acts: the activation record of ia_field_fwd_train;
 * rgb [V,3]: its colour output; d_rgb [V,3], d_sigma [V]: incoming gradients; *scale
 * (device scalar): factor applied before gradients are rounded to half (tcnn: the fixed
 * 1024x loss scale of DNeRF.py:58), divided out of all results.  Outputs: dfeat fp32 [V,2L]
 * (input of ia_hashgrid_bwd) and the five weight gradients, ACCUMULATED in fp32 into
 * g_* (tcnn layouts [out][in]; caller zero-fills) -- per-workgroup partial sums in `ws`, added
 * up in a fixed order: no atomics, bitwise reproducible.  Needs field->mlp_frags.          */
/* The scale for ia_field_bwd: *scale = 1024 / max(|d_rgb * rgb (1 - rgb)|, |d_sigma|) over the live samples, in one
 * launch without a host read.  state2: DEVICE uint32[2], zero on first use, left zero by every call.            */
int ia_field_grad_scale(const float *rgb, const float *d_rgb, const float *d_sigma, int V,
                        const int32_t *n_dev, uint32_t *state2, float *scale, void *stream);
size_t ia_field_bwd_workspace_bytes(int V, int n_levels);
int ia_field_bwd(const uint16_t *acts, const float *rgb, const float *d_rgb,
                 const float *d_sigma, int V, const int32_t *n_dev, const float *scale,
                 const ia_field *field, float *dfeat, float *g_sig_w1, float *g_sig_w2,
                 float *g_col_w1, float *g_col_w2, float *g_col_w3, void *ws, size_t ws_bytes,
                 void *stream);
/* Hash-grid backward (tcnn kernel_grid_backward): dtable fp32 [n_entries,2]
 * += interpolation weight * dfeat [V,2L] (fp32 atomics; caller zero-fills).
 * dx: optional [V,3] gradient w.r.t. the (un-normalised) input positions
 * (tcnn kernel_grid_backward_input), needed when SMPL poses are optimised.   */
int ia_hashgrid_bwd(const float *x, int V, const int32_t *n_dev, const ia_field *field,
                    const float *dfeat, float *dtable, float *dx, void *stream);
/* The same scatter for levels [l_begin, l_end) only (no dx): a finished slice of the table
 * gradient can go to the data-parallel all-reduce while the other levels are still being
 * scattered (no counterpart in the single-GPU reference).                        */
int ia_hashgrid_bwd_levels(const float *x, int V, const int32_t *n_dev, const ia_field *field,
                           const float *dfeat, float *dtable, int l_begin, int l_end,
                           void *stream);
size_t ia_field_frags_bytes(void);
int ia_field_prepare(const ia_field *field, uint16_t *frags_out, void *stream);
/* Encoding only (the roofline kernel in isolation): feat fp16 [V,32].        */
int ia_hashgrid_fwd(const float *x, int V, const ia_field *field,
                    uint16_t *feat, void *stream);
/* Same encoding, XCD-sharded (one hashed level per XCD L2, see ia_field.enc_ws):
 * planes [n_levels][stride] of packed half2, stride >= V.  Needs the tcnn default
 * table shape (4 dense + 4 or 12 equal hashed levels).                          */
int ia_hashgrid_fwd_planes(const float *x, int V, const ia_field *field,
                           uint32_t *planes, size_t stride, void *stream);

/* ---- a6: candidate reduction ----------------------------------------------
 * Replaces SNARFDeformer.deform_test tail (snarf_deformer.py:130-141):
 * nan_to_num, sigma = max over candidates (invalid contribute 0 in test mode,
 * `fill` = -1e5 in train mode, snarf_deformer.py:147), rgb of the arg-max.   */
int ia_candidate_max(const float *cand_rgb, const float *cand_sigma,
                     const int32_t *pt_off, const uint8_t *pt_cnt, int P,
                     const int32_t *n_pts_dev, int n_init, float fill,
                     int nan_to_num, float *rgb, float *sigma, void *stream);

/* ---- a13: raymarch_test ----------------------------------------------------
 * Replaces raymarch_test(rays_o, rays_d, nears, fars, alive, grid, scale,
 * offset, step, N_steps) (renderers/cuda/raymarcher.cpp:16-36,
 * raymarcher.cu:13-112).  occ_bits: bit-packed G^3 grid, bit index
 * (x*G+y)*G+z.  nears is updated in place (raymarcher.cu:72).  Outputs are
 * fully written (zeros where no sample): pts [n_alive,N_steps,3],
 * deltas/depths [n_alive,N_steps].                                           */
int ia_raymarch_test(const float *rays_o, const float *rays_d, float *nears,
                     const float *fars, const int64_t *alive, int n_alive,
                     const uint32_t *occ_bits, const ia_occ_grid *occ,
                     const float *step_size, int N_steps, float *pts,
                     float *deltas, float *depths, void *stream);

/* ---- a14: composite_test ---------------------------------------------------
 * Replaces composite_test(rgb, sigma, delta, depth, alive, color, depth_out,
 * no_hit, thresh) (raymarcher.cpp:57-75, raymarcher.cu:200-262).             */
int ia_composite_test(const float *rgb, const float *sigma, const float *delta,
                      const float *depth, const int64_t *alive, int n_alive,
                      int N_steps, float *color, float *depth_out,
                      float *no_hit, float thresh, void *stream);

/* ---- a15: raymarch_train ---------------------------------------------------
 * Replaces raymarch_train(...) (raymarcher.cpp:38-55, raymarcher.cu:116-198):
 * depths [n_rays,N_steps] (zeros where unused).                              */
int ia_raymarch_train(const float *rays_o, const float *rays_d,
                      const float *nears, const float *fars, int n_rays,
                      const uint32_t *occ_bits, const ia_occ_grid *occ,
                      const float *step_size, int N_steps, float *depths,
                      void *stream);

/* ---- a16 + a18: occupancy-grid post-processing -----------------------------
 * Replaces the tail of DensityGrid.initialize (density_grid.py:104-110) and
 * max_connected_component (density_grid.py:118-125):
 *   f = 1-exp(-0.01*density); maxpool3; f > min(mean(f), 0.01);
 *   keep the largest 26-connected component.
 * density: [G,G,G] indexed [x][y][z].  Outputs: occ_bits (bit-packed, G^3/32 words + EIGHT tail words: [0] flag, 1 = no border
 * cell is occupied; [1..6] = x_min, x_max, y_min, y_max, z_min, z_max of the occupied cells (min > max: none); [7] spare -- read
 * by ia_render_test's marcher for its exact empty-space skip; occ_bits must be 16-byte aligned), occ_bool [G^3] uint8 or NULL.    */
size_t ia_occupancy_workspace_bytes(int G);
int ia_occupancy_from_density(const float *density, int G, uint32_t *occ_bits,
                              uint8_t *occ_bool, void *ws, size_t ws_bytes,
                              void *stream);
/* Pack a bool grid (e.g. loaded from a checkpoint) into occ_bits (G^3/32 + 8 words: the bits, the border flag and the bounds of
 * the occupied cells, as above).                                             */
int ia_occupancy_pack(const uint8_t *occ_bool, int G, uint32_t *occ_bits,
                      void *stream);

/* ---- a6+a9..a11 fused: observation-space field query ------------------------
 * deformer(pts, net) of the reference (snarf_deformer.py:161-165 eval branch,
 * §3.2 of SURVEY.md) as one call with no host sync: search+filter+compact ->
 * field on valid candidates -> max over candidates.
 * dmax: optional [P] running maximum that sigma is max-ed into (used by the
 * occupancy probes, density_grid.py:99-102); rgb may be NULL.                */
size_t ia_query_workspace_bytes(int P, int n_init);
int ia_deform_query(const float *pts, int P, const int32_t *n_pts_dev,
                    const float *voxel_J, const float *tfs,
                    const int32_t *bone_ids, int n_init,
                    const ia_snarf_grid *grid, const ia_field *field,
                    float *rgb, float *sigma, float *dmax, void *ws,
                    size_t ws_bytes, void *stream);

/* ---- a16 fused: DensityGrid.initialize -------------------------------------
 * (density_grid.py:95-110).  jitter: [iters,G^3,3] uniform [0,1) numbers (what
 * torch.rand_like draws at density_grid.py:100).  aabb: DEVICE [6] bbox of
 * the deformed voxels (ia_precompute's bbox).  Outputs occ_bits / occ_bool /
 * density [G^3].                                                             */
size_t ia_density_init_workspace_bytes(int G, int n_init);
/* Larger workspace that lets ia_density_grid_init probe all `iters` jittered sets in one
 * search / field launch (same result; 0 = not available for these sizes).                  */
size_t ia_density_init_workspace_bytes_batched(int G, int n_init, int iters);
int ia_density_grid_init(const float *jitter, int iters, int G,
                         const float *aabb, const float *voxel_J,
                         const float *tfs, const int32_t *bone_ids, int n_init,
                         const ia_snarf_grid *grid, const ia_field *field,
                         float *density, uint32_t *occ_bits, uint8_t *occ_bool,
                         void *ws, size_t ws_bytes, void *stream);

/* ---- a12 fused: Raymarcher.render_test --------------------------------------
 * (raymarcher_acc.py:83-138) with the model closure specialised to
 * (SNARFDeformer, NeRFNGPNet).  Same N_step schedule
 * max(min(MAX_BATCH // n_alive, MAX_SAMPLES), 1), computed on device; alive
 * compaction by ballot/prefix sum; no host synchronisation: `n_iters` loop
 * iterations are enqueued and iterations with no alive ray exit immediately.
 * n_alive_out (DEVICE int32[2]) holds [0] the alive count after the last enqueued
 * iteration (0 = frame complete; >0 = call again with resume = the number of iterations
 * enqueued so far for this frame, same workspace) and [1] the number of iterations that had
 * rays to process so far.  resume = 0 starts a new frame.  At most 1023 iterations per frame.
 * rays_o/rays_d: [R,3] (SMPL-root frame, after transform_rays_w2s), near/far
 * [R].  aabb: DEVICE [6] occupancy aabb.  bg: [R,3] or NULL (white).
 * Outputs: rgb [R,3], depth [R], alpha [R], counter [R].                     */
size_t ia_render_workspace_bytes(int R, int max_batch, int n_init);
int ia_render_test(const float *rays_o, const float *rays_d, const float *near,
                   const float *far, int R, const float *bg,
                   const uint32_t *occ_bits, int G, const float *aabb,
                   const float *voxel_J, const float *tfs,
                   const int32_t *bone_ids, int n_init,
                   const ia_snarf_grid *grid, const ia_field *field,
                   int max_samples, int max_batch, int n_iters, int resume,
                   float *rgb, float *depth, float *alpha, float *counter,
                   int32_t *n_alive_out, void *ws, size_t ws_bytes,
                   void *stream);

/* transform_rays_w2s (snarf_deformer.py:95-103): o' = R o + t, d' = R d,
 * near = |o'| - 1, far = |o'| + 1.  w2s: DEVICE [4,4].                        */
int ia_transform_rays_w2s(const float *rays_o, const float *rays_d,
                          const float *w2s, int R, float *o_out, float *d_out,
                          float *near, float *far, void *stream);

/* ---- a15 + a8 fused: Raymarcher.render_train over compact samples ----------
 * (raymarcher_acc.py:140-186, composite :25-36, snarf_deformer.py:143-159).
 * ia_march_train_compact: raymarch_train + jitter (z = t + jitter*dt, jitter
 *   [n_rays,max_samples] as torch.rand_like draws it at :156, NULL = 0.5) + compaction:
 *   s_pts [cap,3], s_z [cap], s_slot [cap] (slot in the dense layout), ray_off/ray_cnt
 *   [n_rays], n_samples (device int32, zeroed inside).
 * ia_composite_train_fwd: per sample max over its candidates (invalid = -1e5; cand_cap =
 *   length of the candidate arrays, candidates past it were dropped by the search),
 *   optional sigma noise (noise [n_rays,max_samples] as torch.randn_like draws it at :167:
 *   the sample in slot k of ray n gets noise[n][k], NULL = none),
 *   alpha = 1-exp(-relu(sigma)*dt), T = cumprod(1-alpha+1e-10);
 *   outputs color [n,3] (+T*bg), depth, alpha (= sum w), weights_dense [n,max_samples]
 *   (zero-filled by the caller; only occupied slots are written);
 *   saves s_arg (winning candidate or -1), s_sigma, s_alpha, s_T for the backward.
 * ia_composite_train_bwd: gradients w.r.t. the candidates' rgb [n_cand,3] / sigma
 *   [n_cand] (buffers zero-filled by the caller).                              */
int ia_march_train_compact(const float *rays_o, const float *rays_d, const float *nears,
                           const float *fars, int n_rays, const uint32_t *occ_bits,
                           const ia_occ_grid *occ, int max_samples, const float *jitter,
                           float *s_pts, float *s_z, int32_t *s_slot, int32_t *ray_off,
                           int32_t *ray_cnt, int32_t *n_samples, int sample_cap, void *stream);
int ia_composite_train_fwd(const float *cand_rgb, const float *cand_sigma, int cand_cap,
                           const int32_t *pt_off, const uint8_t *pt_cnt, int n_init, const int32_t *ray_off,
                           const int32_t *ray_cnt, const float *s_z, const float *nears,
                           const float *fars, int n_rays, int max_samples, const float *noise,
                           float noise_scale, const float *bg, float *color, float *depth,
                           float *alpha, float *weights_dense, const int32_t *s_slot,
                           int32_t *s_arg, float *s_sigma, float *s_alpha, float *s_T, void *stream);
int ia_composite_train_bwd(const float *d_color, const float *d_depth, const float *d_alpha,
                           const float *d_weights, const float *cand_rgb, const int32_t *ray_off,
                           const int32_t *ray_cnt, const float *s_z, const float *nears,
                           const float *fars, int n_rays, int max_samples, const float *bg,
                           const int32_t *s_slot, const int32_t *s_arg, const float *s_sigma,
                           const float *s_alpha, const float *s_T, float *d_cand_rgb,
                           float *d_cand_sigma, void *stream);

/* deform_train's max over candidates (snarf_deformer.py:147-158): index of the
 * winning candidate per point, -1 when an invalid slot (sigma = -1e5) wins;
 * cand_cap = length of cand_sigma (candidates past it were dropped).          */
int ia_candidate_argmax(const float *cand_sigma, int cand_cap, const int32_t *pt_off,
                        const uint8_t *pt_cnt, int P, int n_init, int32_t *arg, void *stream);
/* The gather that follows it (snarf_deformer.py:150-158, torch.gather on the arg-max) and
 * its backward: rgb/sigma [P] <- candidate arg[p] (arg < 0: rgb 0, sigma = fill); the
 * backward is a UNIQUE scatter (a candidate belongs to one point): plain stores into the
 * caller-zeroed d_cand_* arrays.  d_rgb / d_sigma may be NULL.                             */
int ia_candidate_gather_fwd(const float *cand_rgb, const float *cand_sigma, const int32_t *arg,
                            int P, float fill, float *rgb, float *sigma, void *stream);
int ia_candidate_gather_bwd(const float *d_rgb, const float *d_sigma, const int32_t *arg, int P,
                            float *d_cand_rgb, float *d_cand_sigma, void *stream);

/* ---- Raymarcher(smpl_init=True): mesh bootstrap of the training occupancy grid ----------------
 * (density_grid.py:53-75).  sdf[i] = (inside ? -1 : +1) * distance of pts[i] to the triangle mesh
 * (verts [V,3], faces int32 [F,3], watertight); replaces kaolin's point_to_mesh_distance + check_sign
 * (absent: restated from their definitions, floating-point parity with kaolin unpinned).                 */
int ia_mesh_signed_distance(const float *pts, long N, const float *verts, const int32_t *faces,
                            int n_faces, float *sdf, void *stream);
/* cell centres denormalize(coords + 0.5 / G, aabb) of a G^3 grid (density_grid.py:55); aabb: DEVICE [6]. */
int ia_grid_cell_centres(int G, const float *aabb, float *pts, void *stream);

/* ---- f4: the data side of a training step, on the device ----------------------
 * make_rays (instant_avatar/datasets/peoplesnapshot.py:12-25): rays_o / rays_d [H*W,3] fp32 of a pinhole
 * camera, evaluated in fp64 like the reference's numpy code.  K_inv [9], c2w_R [9] (row-major), c2w_t [3]:
 * HOST arrays (three tiny matrices; inv(K) is the caller's np.linalg.inv as in the reference).          */
int ia_make_rays(const double *K_inv, const double *c2w_R, const double *c2w_t, int H, int W,
                 float *rays_o, float *rays_d, void *stream);
/* EdgeSampler's band (instant_avatar/utils/sampler.py:25-28): edge = cv2.dilate(mask, ones(k,k)) -
 * cv2.erode(mask, ones(k,k)), anchor (k/2, k/2), pixels outside the image ignored.  mask, edge: [H,W].    */
size_t ia_mask_edge_workspace_bytes(int H, int W);
int ia_mask_edge(const float *mask, int H, int W, int kernel_size, float *edge, void *ws,
                 size_t ws_bytes, void *stream);
/* cv2.dilate(mask, ones(k,k)) alone: PatchSampler(dilate=k) (sampler.py:62-65).  Same workspace as ia_mask_edge. */
int ia_mask_dilate(const float *mask, int H, int W, int kernel_size, float *dilated, void *ws,
                   size_t ws_bytes, void *stream);
/* np.where(mask[y0:y1, x0:x1])[rank] for n ranks derived from uniform draws u [n] in [0,1) on the device:
 * with replacement rank = floor(u * count) (np.random.randint, sampler.py:33-35); without replacement the
 * floor(u_i * (count - i))-th element not chosen before (np.random.choice(replace=False), sampler.py:69).
 * out_row / out_col [n]: coordinates RELATIVE to the window, row-major order of np.where; -1 when the window
 * holds no nonzero element.  count_out (optional, DEVICE int32): number of nonzeros.  No host sync.        */
size_t ia_nonzero_select_workspace_bytes(int rows, int n);
int ia_nonzero_select(const float *mask, int H, int W, int y0, int y1, int x0, int x1, const float *u,
                      int n, int without_replacement, int32_t *out_row, int32_t *out_col,
                      int32_t *count_out, void *ws, size_t ws_bytes, void *stream);
/* The batch of one training step (peoplesnapshot.py:99-151) for n sampled pixels, given either as flat pixel
 * indices (EdgeSampler) or as n_patch patch corners (PatchSampler; n = n_patch * patch^2, sample (p,i,j) =
 * corner_p + (i,j)): alpha = mask, rgb = img * mask + (1 - mask) * bg, rays gathered.  img: uint8 [H*W,3]
 * (divided by 255 as at :107) or float [H*W,3] (give one, the other NULL).  bg [n,3]: uniform draws
 * (training, :111) or NULL = white (:114).  bg_out / idx_out optional.                                     */
int ia_sample_batch(const uint8_t *img_u8, const float *img_f, const float *mask, const float *rays_o,
                    const float *rays_d, int H, int W, const int32_t *flat_idx,
                    const int32_t *corner_row, const int32_t *corner_col, int n_patch, int patch, int n,
                    const float *bg, float *rgb, float *alpha, float *o_out, float *d_out,
                    float *bg_out, int32_t *idx_out, void *stream);

/* PatchSampler's corners (sampler.py:58-75): row_mask / col_mask [n] = the mask branch (ia_nonzero_select without
 * replacement on the window cropped by patch/2), draws [1 + 2 n] = branch coin, then the anchor draws; the uniform
 * branch is floor(u * (H - patch)) (np.random.randint(0, H - patch)), chosen when coin >= ratio_mask.  rows / cols [n]. */
int ia_patch_corners(const int32_t *row_mask, const int32_t *col_mask, const float *draws, int n, int H, int W,
                     int patch, float ratio_mask, int32_t *rows, int32_t *cols, void *stream);
/* EdgeSampler's flat pixel indices (sampler.py:33-41) from the two ia_nonzero_select results and the draws [n_mask + n_edge + n_rand]:
 * out = [row_mask W + col_mask | row_edge W + col_edge | floor(u H W)], row-major; a pick from an empty mask / band (-1) falls back
 * to the uniform pixel of its own draw (the reference raises there: np.random.randint(0, 0)).                            */
int ia_edge_indices(const int32_t *row_mask, const int32_t *col_mask, const int32_t *row_edge, const int32_t *col_edge,
                    const float *draws, int n_mask, int n_edge, int n_rand, int H, int W, int32_t *out, void *stream);
/* near / far of a frame's n rays (peoplesnapshot.py:146-150): |transl| -/+ 1, transl: DEVICE float[3].          */
int ia_near_far(const float *transl, int n, float *near_out, float *far_out, void *stream);

/* ---- measurement hooks (bench.py only) --------------------------------------
 * When enabled, every launch of the Broyden-search kernel (id 0) and of the
 * field kernel (id 1) is bracketed by HIP events on the caller's stream and the
 * kernels count the units they processed.  ia_profile_get synchronises.
 * units: id 0 -> {(point,init) solves, trilinear grid fetches};
 *        id 1 -> {samples evaluated, 0}.                                      */
int ia_profile_enable(int on);
int ia_profile_reset(void);
int ia_profile_get(int kernel_id, double *total_ms, int64_t *launches,
                   uint64_t *units);
/* measurement helper: acc2[0] += mean(counter), acc2[1] += mean(alpha > 0.5) over the R rays of a rendered frame
 * (what a caller logging samples per ray / coverage of a sequence accumulates), one launch.                     */
int ia_frame_stats(const float *counter, const float *alpha, int R, float *acc2, void *stream);
/* 8-bit RGBA image of a rendered frame: rgba[4 i + c] = (uint8)(clamp(v, 0, 1) * 255), v = rgb[3 i + c] (c < 3) or
 * alpha[i] (c = 3) -- replaces `torch.cat([rgb, alpha[..., None]], -1)` + `(img.cpu().numpy() * 255).astype(np.uint8)` of
 * the frame loop (animate.py:107-113, novel_view.py:120-125).  rgba: R x 4 bytes, 4-byte aligned, channel order as given. */
int ia_pack_rgba8(const float *rgb, const float *alpha, int R, uint8_t *rgba, void *stream);
/* all n (<= 8) counters of a kernel: id 0 -> {solves, trilinear fetches of the algorithm (fuse_cuda_kernel_fast.cu:
 * one per Broyden evaluation), fetches that loaded memory (a fetch whose 8 corners all lie outside the grid is zero
 * without a load)}; id 1 -> {samples evaluated}.  Synchronises.                                                  */
int ia_profile_get_units(int kernel_id, uint64_t *units, int n);
/* registers per lane, static LDS bytes and threads per workgroup, resident workgroups per CU of the Broyden-search
 * kernel as compiled into this library (measurement only).                                                        */
int ia_search_kernel_info(int *vgprs, int *lds_bytes, int *threads, int *workgroups_per_cu);
/* Device self-tests of the Broyden update's shared-reciprocal division (fuse_cuda_kernel_fast.cu:23-55 divides nine
 * numerators by the same scalar; k_search computes the reciprocal chain of IEEE division once).  They launch the same
 * device functions the search kernel uses; called by tests/ only.
 *   ia_selftest_shared_rcp:  q_shared[i] = shared-reciprocal quotient num[i] / den[i], q_ieee[i] = the compiler's
 *                            division of the same operands (all device pointers, n floats each).
 *   ia_selftest_jinv_update: the rank-1 update of n J_inv matrices Ji [n][9] with steps x [n][3] and residual
 *                            differences g [n][3], once as k_search runs it (the shared reciprocal per wave of 64
 *                            consecutive rows when all of them are inside its exponent range -> took_shared[i] = 1,
 *                            the compiler's divisions otherwise) and once with the compiler's divisions only.
 *                            n % 64 == 0.                                                                              */
int ia_selftest_shared_rcp(const float *num, const float *den, int n, float *q_shared, float *q_ieee, void *stream);
int ia_selftest_jinv_update(const float *Ji, const float *x, const float *g, int n, float *out_shared, float *out_plain,
                            uint8_t *took_shared, void *stream);


/* ---- SMPLDeformer training query on compact samples (smpl_deformer.py:88-120 under autograd; fit stage) ----------------
 * ia_smpl_nn_compact:     nearest posed vertex of every sample point (K = 1 knn_points), pts_cano = T_inv[nearest] [pts, 1],
 *                         valid = dist^2 < threshold^2; the valid points compacted: cand_xc [<= P,3], cand_pt [<= P] (the point of
 *                         a candidate), idx [P] (vertex of every point; with `nn_grid`: of every VALID point, -1 elsewhere), pt_off [P] / pt_cnt [P] (0 | 1: the compositor's
 *                         candidate lists with n_init = 1), *n_cand (device counter, zeroed by the call).
 * ia_smpl_nn_compact_bwd: d_cand_xc [cap,3] -> d_T_inv [V,4,4] (zero-filled by the call; d T_inv[idx] += g [x, 1]^T, rows 0..2)
 *                         and d_pts [P,3] (zero-filled; R(T_inv[idx])^T g for the points that have a candidate).  Either may be NULL.
 * ia_ray_samples_bwd:     the sample points are pts = o + z d (raymarcher_acc.py:158, z from the marcher: not differentiated):
 *                         d_o [n_rays,3] = sum over the ray's compact samples of d_pts, d_d = the same weighted with z.    */
int ia_smpl_nn_compact(const float *pts, int P, const int32_t *n_pts_dev, const float *verts, const float *T_inv,
                       int n_verts, float threshold, float *cand_xc, int32_t *cand_pt, int32_t *idx, int32_t *pt_off,
                       uint8_t *pt_cnt, int32_t *n_cand, const void *nn_grid, void *stream);
int ia_smpl_nn_compact_bwd(const float *pts, int P, const int32_t *cand_pt, const int32_t *idx, const int32_t *n_cand, int cap,
                           const float *T_inv, int n_verts, const float *d_cand_xc, float *d_T_inv, float *d_pts,
                           void *stream);
int ia_ray_samples_bwd(const int32_t *ray_off, const int32_t *ray_cnt, const float *s_z, const float *d_pts, int n_rays,
                       float *d_o, float *d_d, void *stream);

/* ---- SMPLDeformer's body model, forward and backward (deformers/smpl_deformer.py:32-77) ---------------------------
 * Replaces, per frame / per fit step, the two smplx `SMPL.forward` evaluations of `SMPLDeformer.initialize` +
 * `prepare_deformer` (body_models.py:289-372 -> lbs.py:152-250) and their two batched `torch.inverse` calls, and -- in the
 * fit stage, where betas / pose / translation are optimised (DNeRF.py:113-128) -- their autograd:
 *   fwd: betas [10], pose [72] (global_orient + body_pose, axis-angle), transl [3] (or NULL), pose_t [72] (the template pose
 *        of :33-35), po_t [V,3] (pose-corrective offsets of the template pose: constant per subject)
 *        -> T_inv [V,4,4] (:66-75), verts [V,3] posed vertices in the SMPL-root frame (:76), verts_t [V,3] template-pose
 *        vertices (optional: `initialize`'s bounding box), w2s [4,4] (optional)
 *   bwd: d_T_inv [V,4,4] (rows 0..2 are read), d_w2s [4,4] or NULL -> d_betas [10] (optional), d_pose [72], d_transl [3] (optional)
 * ia_smpl_body: device pointers, constant per subject; J0 = J_regressor @ v_template [24,3] and JS = J_regressor @ shapedirs
 * [24,3,10] fold the joint regression of lbs.py:190.  ws: ia_smpl_lbs_workspace_bytes(n_verts) bytes (scratch; the backward
 * recomputes what it needs, nothing has to survive from the forward call).                                             */
typedef struct ia_smpl_body {
  const float *v_template;   /* [V,3]        */
  const float *shapedirs;    /* [V,3,10]     */
  const float *posedirs;     /* [207, V*3]   */
  const float *lbs_weights;  /* [V,24]       */
  const float *J0;           /* [24,3]       */
  const float *JS;           /* [24,3,10]    */
  const int32_t *parents;    /* [24]         */
  int n_verts;
} ia_smpl_body;
size_t ia_smpl_lbs_workspace_bytes(int n_verts);
int ia_smpl_lbs_fwd(const ia_smpl_body *body, const float *betas, const float *pose, const float *transl,
                    const float *pose_t, const float *po_t, float *T_inv, float *verts, float *verts_t, float *w2s,
                    void *ws, size_t ws_bytes, void *stream);
int ia_smpl_lbs_bwd(const ia_smpl_body *body, const float *betas, const float *pose, const float *transl,
                    const float *pose_t, const float *po_t, const float *d_T_inv, const float *d_w2s, float *d_betas,
                    float *d_pose, float *d_transl, void *ws, size_t ws_bytes, void *stream);

/* ---- optimiser step (DNeRF.py:46-50, :151-159) ---------------------------------------------------------------------
 * One call replaces what the reference's `self.scaler.unscale_(optimizer); self.scaler.step(optimizer);
 * optimizer.zero_grad()` launch per step through torch: GradScaler's inf / NaN check over ALL gradients (any non-finite
 * element skips the whole update: parameters, moments and step counters stay as they are), `torch.optim.Adam` (betas
 * (0.9, 0.99), eps 1e-15 in every shipped config; no weight decay, no amsgrad) over up to IA_ADAM_MAX_TENSORS parameter
 * tensors with their own learning rates (the three parameter groups of DNeRF.py:46-50), the refresh of the half-precision
 * copy of the parameters that the field kernels read (tcnn holds fp32 master weights next to its half parameters), and,
 * with `zero_grad`, the gradient zero-fill for the next step.  Arithmetic: torch/optim/adam.py `_single_tensor_adam`
 * (see csrc/ia_optim.hip); `step` is the device-resident counter of torch's capturable state, so the call can be captured
 * in a HIP graph; `lr_dev` (device scalar, optional) lets an lr scheduler reach a captured step.
 *   skip_in   optional device scalar: non-zero = skip this step whatever the gradients hold (candidate overflow)
 *   found_inf optional device scalar, written: 1.0 when the step was skipped, else 0.0
 *   ws        ia_adam_workspace_bytes() bytes, zero-filled ONCE by the caller, private to one stream                     */
#define IA_ADAM_MAX_TENSORS 8
typedef struct ia_adam_tensor {
  float *param, *grad, *exp_avg, *exp_avg_sq; /* device fp32 [numel], 16-byte aligned */
  uint16_t *shadow;                           /* device fp16 [numel] copy of param (8-byte aligned) or NULL */
  float *step;                                /* device scalar: steps taken so far */
  const float *lr_dev;                        /* device scalar or NULL -> lr */
  double lr, beta1, beta2, eps;
  long long numel;
} ia_adam_tensor;
size_t ia_adam_workspace_bytes(void);
int ia_adam_step(const ia_adam_tensor *tensors, int n_tensors, const float *skip_in, float *found_inf,
                 int zero_grad, void *ws, size_t ws_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* INSTANTAVATAR_HIP_H */
