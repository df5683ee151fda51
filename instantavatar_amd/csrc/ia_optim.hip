// ia_optim.hip -- the optimiser step of DNeRFModel.training_step as two launches over EVERY parameter tensor (gfx950).
//
// Reference: models/DNeRF.py:46-50 (`torch.optim.Adam(params, lr, betas=(0.9, 0.99), eps=1e-15)`, three parameter groups),
// :151-159 (`self.scaler.unscale_(optimizer)`, `self.scaler.step(optimizer)`: GradScaler's inf / NaN check over all
// gradients skips the update), `optimizer.zero_grad()`; tcnn keeps an fp32 master copy next to the half parameters its
// kernels read.  On the torch stack that is, per step: a multi-tensor non-finite check (35 us), two multi-
// tensor Adam launches (68 us) plus their step-counter bookkeeping, two fp32 -> fp16 casts (18 us), a 52 MB zero-fill of the
// table gradient and a dozen small launches -- ~200 us of a 1.9 ms training step (profiles/r05_train_*_kernel_stats.csv).
//
//   k_adam_check    one read pass over all gradients (HBM-bound: 52 MB): any inf / NaN sets a flag.
//   k_adam_update   one pass: p, g, m, v read; p, m, v written; the fp16 copy of p written; g zero-filled (optional).
//                   HBM-bound: 28 algorithmic bytes per parameter (16 read + 12 written), 34 with the fp16 copy and the zero-fill.
//                   The bias corrections are computed per workgroup in double precision, as Python computes them.
//   k_adam_finish   one workgroup: `found_inf` for the caller, the step counters, the flag cleared for the next step.
//
// Arithmetic = torch.optim.Adam's single-tensor path (torch/optim/adam.py `_single_tensor_adam`) as its vectorised CPU
// kernels evaluate it (FMA in lerp and addcmul), IEEE sqrt / division:
//   m' = fma(g - m, 1 - b1, m);  v' = fma((1 - b2) g, g, b2 v);  denom = sqrt(v') / sqrt(1 - b2^t) + eps;
//   p' = p + (-(lr / (1 - b1^t)) m') / denom
// restated in oracle/oracle.py `adam_step` (numpy), which tests/test_cpu_oracle.py pins to torch.optim.Adam on the CPU.
#include "ia_common.h"

#define IA_ADAM_THREADS 256
#define IA_ADAM_PER_BLOCK (IA_ADAM_THREADS * 16)   // elements per workgroup: 4 x float4 per thread

struct AdamWs {             // caller-provided, zero-initialised once
  unsigned int any_bad;     // some gradient element is inf / NaN (set by k_adam_check, cleared by k_adam_finish)
  unsigned int pad[3];
};
struct AdamArgs {
  float *param[IA_ADAM_MAX_TENSORS], *grad[IA_ADAM_MAX_TENSORS], *m[IA_ADAM_MAX_TENSORS], *v[IA_ADAM_MAX_TENSORS];
  uint16_t *shadow[IA_ADAM_MAX_TENSORS];
  float *step[IA_ADAM_MAX_TENSORS];
  const float *lr_dev[IA_ADAM_MAX_TENSORS];
  double lr[IA_ADAM_MAX_TENSORS], beta1[IA_ADAM_MAX_TENSORS], beta2[IA_ADAM_MAX_TENSORS];
  float eps[IA_ADAM_MAX_TENSORS], w1[IA_ADAM_MAX_TENSORS], w2[IA_ADAM_MAX_TENSORS], b2f[IA_ADAM_MAX_TENSORS];
  long long numel[IA_ADAM_MAX_TENSORS];
  int block0[IA_ADAM_MAX_TENSORS + 1];   // first workgroup of every tensor
  int n;
};

extern "C" size_t ia_adam_workspace_bytes(void) { return ia_align(sizeof(AdamWs)); }

__device__ __forceinline__ bool non_finite(float x) { return (__float_as_uint(x) & 0x7f800000u) == 0x7f800000u; }

__device__ __forceinline__ int adam_tensor_of(const AdamArgs &a, int b) {
  int t = 0;
#pragma unroll
  for (int k = 1; k < IA_ADAM_MAX_TENSORS; k++) t += (k < a.n && b >= a.block0[k]) ? 1 : 0;
  return t;
}

// (No "last workgroup finishes" ticket here: 3 187 workgroups x one device-scope atomic on ONE address serialise memory-side at
// ~11 ns each -- the first version of this kernel took 79 us, 35 us of it the ticket.  The flag is only touched when a
// non-finite value was actually seen; the bookkeeping moved into k_adam_update's prologue and the one-workgroup k_adam_finish.)
__global__ __launch_bounds__(IA_ADAM_THREADS) void k_adam_check(AdamArgs a, AdamWs *__restrict__ ws) {
  const int t = adam_tensor_of(a, blockIdx.x);
  const long long base = (long long)(blockIdx.x - a.block0[t]) * IA_ADAM_PER_BLOCK;
  const long long n = a.numel[t];
  const float *__restrict__ g = a.grad[t];
  bool bad = false;
  float4 q[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const long long i = base + ((long long)k * IA_ADAM_THREADS + threadIdx.x) * 4;
    q[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i + 3 < n) q[k] = *reinterpret_cast<const float4 *>(g + i);
    else for (long long j = i; j < n; j++) bad |= non_finite(g[j]);
  }
#pragma unroll
  for (int k = 0; k < 4; k++) bad |= non_finite(q[k].x) | non_finite(q[k].y) | non_finite(q[k].z) | non_finite(q[k].w);
  if (__ballot(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(&ws->any_bad, 1u);
}

// one workgroup, behind k_adam_update: the outcome for the caller, the step counters, the flag for the next step
__global__ void k_adam_finish(AdamArgs a, const float *__restrict__ skip_in, float *__restrict__ found_inf, AdamWs *__restrict__ ws) {
  const bool skip = ws->any_bad != 0u || (skip_in != nullptr && !(*skip_in == 0.f));
  __syncthreads();
  if (threadIdx.x < (unsigned)a.n && !skip) a.step[threadIdx.x][0] += 1.f;   // a skipped step leaves the counters alone (torch's fused Adam with found_inf)
  if (threadIdx.x == 0) {
    if (found_inf) *found_inf = skip ? 1.f : 0.f;
    ws->any_bad = 0u;
  }
}

template <bool ZERO_GRAD>
__global__ __launch_bounds__(IA_ADAM_THREADS) void k_adam_update(AdamArgs a, const float *__restrict__ skip_in, const AdamWs *__restrict__ ws) {
  const int t = adam_tensor_of(a, blockIdx.x);
  const long long base = (long long)(blockIdx.x - a.block0[t]) * IA_ADAM_PER_BLOCK;
  const long long n = a.numel[t];
  float *__restrict__ p = a.param[t];
  float *__restrict__ g = a.grad[t];
  float *__restrict__ m = a.m[t];
  float *__restrict__ v = a.v[t];
  uint16_t *__restrict__ sh = a.shadow[t];
  // every workgroup derives the step's scalars itself (one lane, ~1 k double-precision instructions once per
  // workgroup, against 4 096 parameters x ~60 instructions): torch/optim/adam.py -- step += 1; bias_correction{1,2} = 1 - beta ** step;
  // step_size = lr / bias_correction1; bias_correction2_sqrt = bias_correction2 ** 0.5, Python floats (double), cast to the tensors'
  // type where they are applied.  (a NaN skip flag skips too)
  const bool skip = ws->any_bad != 0u || (skip_in != nullptr && !(*skip_in == 0.f));
  const float w1 = a.w1[t], w2 = a.w2[t], b2f = a.b2f[t], eps = a.eps[t];
  __shared__ float s_sc[2];
  if (threadIdx.x == 0) {
    const double tt = (double)a.step[t][0] + 1.0;
    const double lr = a.lr_dev[t] ? (double)a.lr_dev[t][0] : a.lr[t];
    s_sc[0] = (float)(-(lr / (1.0 - pow(a.beta1[t], tt))));
    s_sc[1] = (float)sqrt(1.0 - pow(a.beta2[t], tt));
  }
  __syncthreads();
  const float nss = s_sc[0], bc2s = s_sc[1];
  auto one = [&](float pi, float gi, float &mi, float &vi) -> float {
    mi = __builtin_fmaf(gi - mi, w1, mi);               // exp_avg.lerp_(grad, 1 - beta1)
    vi = __builtin_fmaf(w2 * gi, gi, b2f * vi);         // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    const float denom = sqrtf(vi) / bc2s + eps;         // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
    return pi + (nss * mi) / denom;                     // param.addcdiv_(exp_avg, denom, value=-step_size)
  };
  if (skip && !ZERO_GRAD) return;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const long long i = base + ((long long)k * IA_ADAM_THREADS + threadIdx.x) * 4;
    if (i + 3 < n) {
      if (!skip) {
        const float4 gq = *reinterpret_cast<const float4 *>(g + i);
        float4 pq = *reinterpret_cast<const float4 *>(p + i);
        float4 mq = *reinterpret_cast<const float4 *>(m + i);
        float4 vq = *reinterpret_cast<const float4 *>(v + i);
        pq.x = one(pq.x, gq.x, mq.x, vq.x); pq.y = one(pq.y, gq.y, mq.y, vq.y);
        pq.z = one(pq.z, gq.z, mq.z, vq.z); pq.w = one(pq.w, gq.w, mq.w, vq.w);
        *reinterpret_cast<float4 *>(p + i) = pq;
        *reinterpret_cast<float4 *>(m + i) = mq;
        *reinterpret_cast<float4 *>(v + i) = vq;
        if (sh) {
          union { _Float16 h[4]; uint2 u; } c;
          c.h[0] = (_Float16)pq.x; c.h[1] = (_Float16)pq.y; c.h[2] = (_Float16)pq.z; c.h[3] = (_Float16)pq.w;
          *reinterpret_cast<uint2 *>(sh + i) = c.u;
        }
      }
      if (ZERO_GRAD) *reinterpret_cast<float4 *>(g + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      for (long long j = i; j < n; j++) {
        if (!skip) {
          float mi = m[j], vi = v[j];
          const float pn = one(p[j], g[j], mi, vi);
          p[j] = pn; m[j] = mi; v[j] = vi;
          if (sh) { union { _Float16 h; uint16_t u; } c; c.h = (_Float16)pn; sh[j] = c.u; }
        }
        if (ZERO_GRAD) g[j] = 0.f;
      }
    }
  }
}

extern "C" int ia_adam_step(const ia_adam_tensor *tensors, int n_tensors, const float *skip_in, float *found_inf,
                            int zero_grad, void *ws, size_t ws_bytes, void *stream) {
  IA_CHECK_ARG(tensors && n_tensors >= 1 && n_tensors <= IA_ADAM_MAX_TENSORS, "ia_adam_step: 1..%d tensors per call", IA_ADAM_MAX_TENSORS);
  IA_CHECK_ARG(ws && ws_bytes >= ia_adam_workspace_bytes(), "ia_adam_step: workspace too small");
  AdamArgs a;
  memset(&a, 0, sizeof(a));
  a.n = n_tensors;
  long long blocks = 0;
  for (int k = 0; k < n_tensors; k++) {
    const ia_adam_tensor &t = tensors[k];
    IA_CHECK_ARG(t.param && t.grad && t.exp_avg && t.exp_avg_sq && t.step && t.numel > 0, "ia_adam_step: tensor %d: null pointer or empty", k);
    IA_CHECK_ARG((((uintptr_t)t.param | (uintptr_t)t.grad | (uintptr_t)t.exp_avg | (uintptr_t)t.exp_avg_sq) & 15u) == 0 && ((uintptr_t)t.shadow & 7u) == 0,
                 "ia_adam_step: tensor %d: pointers must be 16-byte aligned (fp16 copy: 8)", k);
    IA_CHECK_ARG(t.beta1 >= 0 && t.beta1 < 1 && t.beta2 >= 0 && t.beta2 < 1 && t.eps >= 0, "ia_adam_step: tensor %d: bad hyper-parameters", k);
    a.param[k] = t.param; a.grad[k] = t.grad; a.m[k] = t.exp_avg; a.v[k] = t.exp_avg_sq; a.shadow[k] = t.shadow;
    a.step[k] = t.step; a.lr_dev[k] = t.lr_dev; a.lr[k] = t.lr; a.beta1[k] = t.beta1; a.beta2[k] = t.beta2;
    a.eps[k] = (float)t.eps; a.w1[k] = (float)(1.0 - t.beta1); a.w2[k] = (float)(1.0 - t.beta2); a.b2f[k] = (float)t.beta2;
    a.numel[k] = t.numel;
    a.block0[k] = (int)blocks;
    blocks += (t.numel + IA_ADAM_PER_BLOCK - 1) / IA_ADAM_PER_BLOCK;
    IA_CHECK_ARG(blocks < INT_MAX, "ia_adam_step: too many elements");
  }
  for (int k = n_tensors; k <= IA_ADAM_MAX_TENSORS; k++) a.block0[k] = (int)blocks;
  hipStream_t s = (hipStream_t)stream;
  AdamWs *w = (AdamWs *)ws;
  hipLaunchKernelGGL(k_adam_check, dim3((unsigned)blocks), dim3(IA_ADAM_THREADS), 0, s, a, w);
  IA_LAUNCH_CHECK("k_adam_check");
  if (zero_grad) hipLaunchKernelGGL(k_adam_update<true>, dim3((unsigned)blocks), dim3(IA_ADAM_THREADS), 0, s, a, skip_in, (const AdamWs *)w);
  else hipLaunchKernelGGL(k_adam_update<false>, dim3((unsigned)blocks), dim3(IA_ADAM_THREADS), 0, s, a, skip_in, (const AdamWs *)w);
  IA_LAUNCH_CHECK("k_adam_update");
  hipLaunchKernelGGL(k_adam_finish, dim3(1), dim3(64), 0, s, a, skip_in, found_inf, w);
  IA_LAUNCH_CHECK("k_adam_finish");
  return IA_OK;
}
