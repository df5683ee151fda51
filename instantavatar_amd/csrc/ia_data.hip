// ia_data.hip -- the data side of a training step on the device (row f4 of SURVEY.md section 8):
// camera rays, the Patch / Edge ray samplers and the batch gather, so that a 1.2 ms training step is
// not fed by a host loop of numpy / cv2 calls per frame.
//
// Reference semantics: instant_avatar/datasets/peoplesnapshot.py:12-25 (get_ray_directions, make_rays),
// :99-151 (__getitem__: masked compositing with a random background, sampler call, near / far),
// instant_avatar/utils/sampler.py:5-46 (EdgeSampler: cv2.erode / cv2.dilate with a k x k box, np.where,
// np.random.randint), :48-82 (PatchSampler: np.where on the cropped mask, np.random.choice without
// replacement, patch slicing).  Random numbers are INPUTS (uniform draws in [0,1) made by the caller),
// so that the CPU checker can be fed the same draws.
#include "ia_common.h"

// ---------------------------------------------------------------------------
// make_rays (peoplesnapshot.py:17-25).  The reference evaluates it in numpy float64 (camera matrices come
// out of np.load / np.linalg.inv as float64) and casts the result to float32: same here, in double.
//   d_c = [x, y, 1] @ inv(K)^T ; d_w = d_c @ R^T ; d_w /= |d_w| ; o_w = t
// ---------------------------------------------------------------------------
struct RayCam { double Kinv[9]; double R[9]; double t[3]; };

__global__ __launch_bounds__(256) void k_make_rays(RayCam cam, int H, int W, float *__restrict__ rays_o,
                                                   float *__restrict__ rays_d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * W) return;
  const double xy[3] = {(double)(float)(i % W), (double)(float)(i / W), 1.0};
  double dc[3], dw[3];
#pragma unroll
  for (int a = 0; a < 3; a++) dc[a] = xy[0] * cam.Kinv[a * 3] + xy[1] * cam.Kinv[a * 3 + 1] + xy[2] * cam.Kinv[a * 3 + 2];
#pragma unroll
  for (int a = 0; a < 3; a++) dw[a] = dc[0] * cam.R[a * 3] + dc[1] * cam.R[a * 3 + 1] + dc[2] * cam.R[a * 3 + 2];
  const double nrm = sqrt(dw[0] * dw[0] + dw[1] * dw[1] + dw[2] * dw[2]);
#pragma unroll
  for (int a = 0; a < 3; a++) {
    rays_d[(size_t)i * 3 + a] = (float)(dw[a] / nrm);
    rays_o[(size_t)i * 3 + a] = (float)cam.t[a];
  }
}

extern "C" int ia_make_rays(const double *K_inv, const double *c2w_R, const double *c2w_t, int H, int W, float *rays_o,
                            float *rays_d, void *stream) {
  IA_CHECK_ARG(K_inv && c2w_R && c2w_t && rays_o && rays_d && H > 0 && W > 0, "ia_make_rays: bad arguments");
  RayCam cam;
  for (int i = 0; i < 9; i++) { cam.Kinv[i] = K_inv[i]; cam.R[i] = c2w_R[i]; }
  for (int i = 0; i < 3; i++) cam.t[i] = c2w_t[i];
  hipLaunchKernelGGL(k_make_rays, dim3(ia_div_up((long)H * W, 256)), dim3(256), 0, (hipStream_t)stream, cam, H, W, rays_o, rays_d);
  IA_LAUNCH_CHECK("k_make_rays");
  return IA_OK;
}

// ---------------------------------------------------------------------------
// EdgeSampler's band: mask_e = dilate(mask, ones(k,k)) - erode(mask, ones(k,k))  (sampler.py:25-28).
// cv2 semantics: anchor at (k/2, k/2) -> window [-k/2, k-1-k/2] in both axes; pixels outside the image do
// not take part (cv2's default border value is +max for erode, -max for dilate).  Separable: rows, then columns.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_minmax_rows(const float *__restrict__ m, int H, int W, int k,
                                                     float *__restrict__ mn, float *__restrict__ mx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * W) return;
  const int y = i / W, x = i - y * W, a = k / 2;
  float lo = INFINITY, hi = -INFINITY;
  for (int d = -a; d <= k - 1 - a; d++) {
    const int xx = x + d;
    if (xx < 0 || xx >= W) continue;
    const float v = m[y * W + xx];
    lo = fminf(lo, v); hi = fmaxf(hi, v);
  }
  mn[i] = lo; mx[i] = hi;
}

__global__ __launch_bounds__(256) void k_minmax_cols(const float *__restrict__ mn, const float *__restrict__ mx, int H, int W,
                                                     int k, float *__restrict__ edge, float *__restrict__ dilated) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * W) return;
  const int y = i / W, x = i - y * W, a = k / 2;
  float lo = INFINITY, hi = -INFINITY;
  for (int d = -a; d <= k - 1 - a; d++) {
    const int yy = y + d;
    if (yy < 0 || yy >= H) continue;
    lo = fminf(lo, mn[yy * W + x]); hi = fmaxf(hi, mx[yy * W + x]);
  }
  if (edge) edge[i] = hi - lo;  // mask_o - mask_i
  if (dilated) dilated[i] = hi;
}

extern "C" size_t ia_mask_edge_workspace_bytes(int H, int W) { return 2 * ia_align((size_t)H * W * 4) + 256; }

extern "C" int ia_mask_edge(const float *mask, int H, int W, int kernel_size, float *edge, void *ws, size_t ws_bytes,
                            void *stream) {
  IA_CHECK_ARG(mask && edge && ws && H > 0 && W > 0 && kernel_size > 0, "ia_mask_edge: bad arguments");
  if (ws_bytes < ia_mask_edge_workspace_bytes(H, W)) return ia_set_error(IA_ERR_WORKSPACE, "ia_mask_edge: workspace too small");
  WsCarver w(ws, ws_bytes);
  float *mn = w.take<float>((size_t)H * W), *mx = w.take<float>((size_t)H * W);
  const dim3 g(ia_div_up((long)H * W, 256)), b(256);
  hipLaunchKernelGGL(k_minmax_rows, g, b, 0, (hipStream_t)stream, mask, H, W, kernel_size, mn, mx);
  hipLaunchKernelGGL(k_minmax_cols, g, b, 0, (hipStream_t)stream, mn, mx, H, W, kernel_size, edge, (float *)nullptr);
  IA_LAUNCH_CHECK("ia_mask_edge");
  return IA_OK;
}

// cv2.dilate(mask, ones(k, k)) alone (PatchSampler(dilate = k), sampler.py:62-65; `sampler.dilate=8` in bash/run-neuman-demo.sh)
extern "C" int ia_mask_dilate(const float *mask, int H, int W, int kernel_size, float *dilated, void *ws, size_t ws_bytes,
                              void *stream) {
  IA_CHECK_ARG(mask && dilated && ws && H > 0 && W > 0 && kernel_size > 0, "ia_mask_dilate: bad arguments");
  if (ws_bytes < ia_mask_edge_workspace_bytes(H, W)) return ia_set_error(IA_ERR_WORKSPACE, "ia_mask_dilate: workspace too small");
  WsCarver w(ws, ws_bytes);
  float *mn = w.take<float>((size_t)H * W), *mx = w.take<float>((size_t)H * W);
  const dim3 g(ia_div_up((long)H * W, 256)), b(256);
  hipLaunchKernelGGL(k_minmax_rows, g, b, 0, (hipStream_t)stream, mask, H, W, kernel_size, mn, mx);
  hipLaunchKernelGGL(k_minmax_cols, g, b, 0, (hipStream_t)stream, mn, mx, H, W, kernel_size, (float *)nullptr, dilated);
  IA_LAUNCH_CHECK("ia_mask_dilate");
  return IA_OK;
}

// ---------------------------------------------------------------------------
// "np.where(mask)[rank]" without a host round trip: the r-th nonzero element (row-major order, as np.where
// returns them) of a rectangular window of `mask`, for n uniform draws u in [0,1):
//   with replacement    (np.random.randint(0, count, n)):          rank_i = floor(u_i * count)
//   without replacement (np.random.choice(count, n, replace=False)): rank_i = the floor(u_i * (count - i))-th
//                        of the elements not chosen by draws 0..i-1
// Three launches: nonzeros per row (one wave per row, ballot), prefix over the rows + ranks (one workgroup),
// selection (one wave per draw: binary search over the row prefix, ballot scan of the row).
// count == 0 -> index -1 for every draw (the reference would raise in np.random.randint / choice).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_row_counts(const float *__restrict__ mask, int W, int y0, int rows, int x0, int cols,
                                                    int32_t *__restrict__ rowcnt) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float *p = mask + (size_t)(y0 + row) * W + x0;
  int c = 0;
  for (int b = 0; b < cols; b += 64) {
    const int x = b + lane;
    c += __popcll(__ballot(x < cols && p[x] != 0.f));
  }
  if (lane == 0) rowcnt[row] = c;
}

__global__ __launch_bounds__(256) void k_row_prefix_ranks(const int32_t *__restrict__ rowcnt, int rows, const float *__restrict__ u,
                                                          int n, int without_replacement, int32_t *__restrict__ rowstart,
                                                          int32_t *__restrict__ ranks, int32_t *__restrict__ count_out) {
  __shared__ int s_part[256];
  __shared__ int s_total;
  const int t = threadIdx.x;
  const int per = (rows + 255) / 256, r0 = min(t * per, rows), r1 = min(r0 + per, rows);
  int sum = 0;
  for (int r = r0; r < r1; r++) sum += rowcnt[r];
  s_part[t] = sum;
  __syncthreads();
  if (t == 0) {
    int acc = 0;
    for (int k = 0; k < 256; k++) { const int c = s_part[k]; s_part[k] = acc; acc += c; }
    s_total = acc;
    rowstart[rows] = acc;
    if (count_out) *count_out = acc;
  }
  __syncthreads();
  int acc = s_part[t];
  for (int r = r0; r < r1; r++) { rowstart[r] = acc; acc += rowcnt[r]; }
  const int total = s_total;
  if (!without_replacement) {
    for (int i = t; i < n; i += 256) {
      int r = total > 0 ? (int)floorf(u[i] * (float)total) : -1;
      if (r >= total) r = total - 1;  // u * total can round up to total in float
      ranks[i] = r;
    }
  } else if (t == 0) {
    // sequential draw from the remaining elements (n is tiny: patches per frame)
    for (int i = 0; i < n; i++) {
      const int left = total - i;
      if (left <= 0) { ranks[i] = -1; continue; }
      int r = (int)floorf(u[i] * (float)left);
      if (r >= left) r = left - 1;
      // r is a rank among the NOT yet chosen elements: the fixed point of r = r' + #{chosen <= r}
      const int rp = r;
      while (true) {
        int cnt = 0;
        for (int j = 0; j < i; j++) cnt += (ranks[j] >= 0 && ranks[j] <= r) ? 1 : 0;
        if (rp + cnt == r) break;
        r = rp + cnt;
      }
      ranks[i] = r;
    }
  }
}

__global__ __launch_bounds__(256) void k_select_ranked(const float *__restrict__ mask, int W, int y0, int rows, int x0, int cols,
                                                       const int32_t *__restrict__ rowstart, const int32_t *__restrict__ ranks,
                                                       int n, int32_t *__restrict__ out_row, int32_t *__restrict__ out_col) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  const int r = ranks[i];
  if (r < 0) { if (lane == 0) { out_row[i] = -1; out_col[i] = -1; } return; }
  int lo = 0, hi = rows - 1;  // last row whose start is <= r
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (rowstart[mid] <= r) lo = mid; else hi = mid - 1;
  }
  const float *p = mask + (size_t)(y0 + lo) * W + x0;
  int need = r - rowstart[lo];  // the need-th nonzero of this row
  int col = -1;
  for (int b = 0; b < cols; b += 64) {
    const int x = b + lane;
    const unsigned long long m = __ballot(x < cols && p[x] != 0.f);
    const int c = __popcll(m);
    if (need < c) {
      unsigned long long mm = m;
      for (int k = 0; k < need; k++) mm &= mm - 1;  // drop the `need` lowest set bits (uniform across the wave)
      col = b + __ffsll((long long)mm) - 1;
      break;
    }
    need -= c;
  }
  if (lane == 0) { out_row[i] = lo; out_col[i] = col; }
}

extern "C" size_t ia_nonzero_select_workspace_bytes(int rows, int n) {
  return ia_align((size_t)rows * 4) + ia_align((size_t)(rows + 1) * 4) + ia_align((size_t)(n > 0 ? n : 1) * 4) + 256;
}

extern "C" int ia_nonzero_select(const float *mask, int H, int W, int y0, int y1, int x0, int x1, const float *u, int n,
                                 int without_replacement, int32_t *out_row, int32_t *out_col, int32_t *count_out, void *ws,
                                 size_t ws_bytes, void *stream) {
  IA_CHECK_ARG(mask && H > 0 && W > 0 && 0 <= y0 && y0 < y1 && y1 <= H && 0 <= x0 && x0 < x1 && x1 <= W, "ia_nonzero_select: bad window [%d:%d, %d:%d] of %d x %d", y0, y1, x0, x1, H, W);
  IA_CHECK_ARG(n >= 0 && (n == 0 || (u && out_row && out_col)) && ws, "ia_nonzero_select: null pointer");
  IA_CHECK_ARG(!without_replacement || n <= 256, "ia_nonzero_select: at most 256 draws without replacement");
  const int rows = y1 - y0, cols = x1 - x0;
  if (ws_bytes < ia_nonzero_select_workspace_bytes(rows, n)) return ia_set_error(IA_ERR_WORKSPACE, "ia_nonzero_select: workspace too small");
  WsCarver w(ws, ws_bytes);
  int32_t *rowcnt = w.take<int32_t>(rows), *rowstart = w.take<int32_t>(rows + 1), *ranks = w.take<int32_t>(n > 0 ? n : 1);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_row_counts, dim3(ia_div_up(rows, 4)), dim3(256), 0, s, mask, W, y0, rows, x0, cols, rowcnt);
  hipLaunchKernelGGL(k_row_prefix_ranks, dim3(1), dim3(256), 0, s, rowcnt, rows, u, n, without_replacement, rowstart, ranks, count_out);
  if (n > 0)
    hipLaunchKernelGGL(k_select_ranked, dim3(ia_div_up(n, 4)), dim3(256), 0, s, mask, W, y0, rows, x0, cols, rowstart, ranks, n,
                       out_row, out_col);
  IA_LAUNCH_CHECK("ia_nonzero_select");
  return IA_OK;
}

// ---------------------------------------------------------------------------
// The batch of a training step (peoplesnapshot.py:99-151) for n sampled pixels given by flat indices
// (EdgeSampler) or by n_patch patch corners (PatchSampler: pixel (p, i, j) = corner_p + (i, j)):
//   alpha = msk ; rgb = img * msk + (1 - msk) * bg ; rays_o / rays_d gathered.
// img is the frame as stored (uint8 BGR->as given, /255 like :107) or float; bg: caller-provided uniform
// draws [n,3] (np.random.rand at :111, drawn for the sampled pixels only) or NULL = white (:114-115).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sample_batch(const uint8_t *__restrict__ img_u8, const float *__restrict__ img_f,
                                                      const float *__restrict__ mask, const float *__restrict__ rays_o,
                                                      const float *__restrict__ rays_d, int W, const int32_t *__restrict__ flat_idx,
                                                      const int32_t *__restrict__ corner_row, const int32_t *__restrict__ corner_col,
                                                      int P, int n, const float *__restrict__ bg, float *__restrict__ rgb,
                                                      float *__restrict__ alpha, float *__restrict__ o_out, float *__restrict__ d_out,
                                                      float *__restrict__ bg_out, int32_t *__restrict__ idx_out) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  int pix;
  if (flat_idx) pix = flat_idx[s];
  else {
    const int p = s / (P * P), r = s % (P * P);
    pix = (corner_row[p] + r / P) * W + corner_col[p] + r % P;
  }
  if (idx_out) idx_out[s] = pix;
  const float m = mask[pix];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float v = img_u8 ? (float)((double)img_u8[(size_t)pix * 3 + c] / 255.0) : img_f[(size_t)pix * 3 + c];  // (img / 255).astype(float32)
    const float b = bg ? bg[(size_t)s * 3 + c] : 1.f;
    rgb[(size_t)s * 3 + c] = v * m + (1.f - m) * b;
    if (bg_out) bg_out[(size_t)s * 3 + c] = b;
    o_out[(size_t)s * 3 + c] = rays_o[(size_t)pix * 3 + c];
    d_out[(size_t)s * 3 + c] = rays_d[(size_t)pix * 3 + c];
  }
  alpha[s] = m;
}

extern "C" int ia_sample_batch(const uint8_t *img_u8, const float *img_f, const float *mask, const float *rays_o,
                               const float *rays_d, int H, int W, const int32_t *flat_idx, const int32_t *corner_row,
                               const int32_t *corner_col, int n_patch, int patch, int n, const float *bg, float *rgb,
                               float *alpha, float *o_out, float *d_out, float *bg_out, int32_t *idx_out, void *stream) {
  IA_CHECK_ARG(n >= 0, "ia_sample_batch: n < 0");
  if (n == 0) return IA_OK;
  IA_CHECK_ARG((img_u8 || img_f) && mask && rays_o && rays_d && rgb && alpha && o_out && d_out && H > 0 && W > 0, "ia_sample_batch: null pointer");
  IA_CHECK_ARG(flat_idx || (corner_row && corner_col && patch > 0 && n == n_patch * patch * patch), "ia_sample_batch: give flat indices or patch corners with n = n_patch * patch^2");
  hipLaunchKernelGGL(k_sample_batch, dim3(ia_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, img_u8, img_f, mask, rays_o, rays_d, W,
                     flat_idx, corner_row, corner_col, patch, n, bg, rgb, alpha, o_out, d_out, bg_out, idx_out);
  IA_LAUNCH_CHECK("k_sample_batch");
  return IA_OK;
}


// ---------------------------------------------------------------------------
// PatchSampler's corner choice (instant_avatar/utils/sampler.py:58-75) after the mask branch has been evaluated by
// ia_nonzero_select: the uniform branch np.random.randint(0, H - P) from the same draws and the blend by the branch
// coin, in one launch instead of a dozen element-wise ones.  draws [1 + 2 n]: coin, then the anchor draws.
// ---------------------------------------------------------------------------
__global__ void k_patch_corners(const int32_t *__restrict__ row_mask, const int32_t *__restrict__ col_mask,
                                const float *__restrict__ draws, int n, int H, int W, int P, float p_mask,
                                int32_t *__restrict__ rows, int32_t *__restrict__ cols) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool use_mask = draws[0] < p_mask;
  // floor(u * (H - P)) clamped to H - P - 1: np.random.randint(0, H - P) (sampler.py:72)
  const float fr = fminf(floorf(draws[1 + i] * (float)(H - P)), (float)(H - P - 1));
  const float fc = fminf(floorf(draws[1 + n + i] * (float)(W - P)), (float)(W - P - 1));
  // an empty (cropped) mask gives -1 from ia_nonzero_select -- np.random.choice raises there; here the patch falls back to
  // the uniform branch instead of handing a negative pixel index to the gather
  const bool from_mask = use_mask && row_mask[i] >= 0 && col_mask[i] >= 0;
  rows[i] = from_mask ? row_mask[i] : (int32_t)fr;
  cols[i] = from_mask ? col_mask[i] : (int32_t)fc;
}

extern "C" int ia_patch_corners(const int32_t *row_mask, const int32_t *col_mask, const float *draws, int n, int H, int W,
                                int patch, float ratio_mask, int32_t *rows, int32_t *cols, void *stream) {
  IA_CHECK_ARG(n >= 0, "ia_patch_corners: n < 0");
  if (n == 0) return IA_OK;
  IA_CHECK_ARG(row_mask && col_mask && draws && rows && cols && patch > 0 && H > patch && W > patch, "ia_patch_corners: bad arguments");
  hipLaunchKernelGGL(k_patch_corners, dim3(ia_div_up(n, 64)), dim3(64), 0, (hipStream_t)stream, row_mask, col_mask, draws, n, H, W, patch,
                     ratio_mask, rows, cols);
  IA_LAUNCH_CHECK("k_patch_corners");
  return IA_OK;
}

// near / far of a frame's rays (peoplesnapshot.py:146-150): distance of the camera (at the origin) to the mid-hip
// translation -/+ 1, the three squares summed in numpy's order for float32.
__global__ void k_near_far(const float *__restrict__ transl, int n, float *__restrict__ near_out, float *__restrict__ far_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = transl[0], y = transl[1], z = transl[2];
  const float d = sqrtf((x * x + y * y) + z * z);
  near_out[i] = d - 1.f;
  far_out[i] = d + 1.f;
}

// EdgeSampler's pixel indices (sampler.py:33-41): [mask picks | edge-band picks | uniform picks], flat row-major.  A pick from an
// empty mask / band (row = col = -1 from ia_nonzero_select) falls back to a uniform pixel, floor(u H W) clamped to H W - 1.
__global__ __launch_bounds__(256) void k_edge_indices(const int32_t *__restrict__ r_m, const int32_t *__restrict__ c_m,
                                                      const int32_t *__restrict__ r_e, const int32_t *__restrict__ c_e,
                                                      const float *__restrict__ draws, int n_mask, int n_edge, int n_rand, int H, int W,
                                                      int32_t *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_mask + n_edge + n_rand) return;
  const float hw = (float)(H * W);
  const int uni = (int)fminf(floorf(draws[i] * hw), (float)(H * W - 1));
  int v = uni;
  if (i < n_mask) { const int p = r_m[i] * W + c_m[i]; if (p >= 0) v = p; }
  else if (i < n_mask + n_edge) { const int k = i - n_mask; const int p = r_e[k] * W + c_e[k]; if (p >= 0) v = p; }
  out[i] = v;
}

extern "C" int ia_edge_indices(const int32_t *row_mask, const int32_t *col_mask, const int32_t *row_edge, const int32_t *col_edge,
                               const float *draws, int n_mask, int n_edge, int n_rand, int H, int W, int32_t *out, void *stream) {
  const int n = n_mask + n_edge + n_rand;
  IA_CHECK_ARG(n_mask >= 0 && n_edge >= 0 && n_rand >= 0 && H > 0 && W > 0, "ia_edge_indices: bad sizes");
  if (n == 0) return IA_OK;
  IA_CHECK_ARG(draws && out && (n_mask == 0 || (row_mask && col_mask)) && (n_edge == 0 || (row_edge && col_edge)), "ia_edge_indices: null pointer");
  hipLaunchKernelGGL(k_edge_indices, dim3(ia_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, row_mask, col_mask, row_edge, col_edge, draws,
                     n_mask, n_edge, n_rand, H, W, out);
  IA_LAUNCH_CHECK("k_edge_indices");
  return IA_OK;
}

extern "C" int ia_near_far(const float *transl, int n, float *near_out, float *far_out, void *stream) {
  IA_CHECK_ARG(n >= 0, "ia_near_far: n < 0");
  if (n == 0) return IA_OK;
  IA_CHECK_ARG(transl && near_out && far_out, "ia_near_far: null pointer");
  hipLaunchKernelGGL(k_near_far, dim3(ia_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, transl, n, near_out, far_out);
  IA_LAUNCH_CHECK("k_near_far");
  return IA_OK;
}
