"""Training-side glue of the hot path (rows a7/a8/a15/a17 of SURVEY.md section 8).

* `field_autograd`: NeRFNGPNet under autograd (what tcnn's torch binding does at ngp.py:78,81).  Forward = the fused HIP
  kernel in training mode (`ia_field_fwd_train`: identical outputs + the fp16 activation record); backward = ONE MFMA
  kernel for both tiny MLPs (`ia_field_bwd`: weight and input gradients, fp32 accumulation; the ten-GEMM formulation
  `_mlp_backward_gemm` is kept as its checker) and the hash-table scatter-add `ia_hashgrid_bwd` (fp32 atomics,
  quad-cooperative; optionally the input gradient dx).  Gradients are rounded to half under a per-call scale
  (`ia_field_grad_scale`; tcnn relies on a fixed 1024x loss scale, DNeRF.py:58) and accumulated in fp32 straight into
  `param.grad`.
* `NeRFLoss` / `NGPLoss`: instant_avatar/utils/loss.py, value + gradient from one kernel (`ia_nerf_loss`).
* `training_step`: DNeRFModel.training_step (models/DNeRF.py:112-161) without Lightning, all configurations (plain, fit,
  refine); non-finite gradients and candidate overflows skip the optimiser step on the device (fused Adam's found_inf).
* data parallel: ONE bucketed gradient average per step, started from inside the backward pass (`parallel.GradReducer`:
  finished level groups of the 52 MB table gradient go to RCCL while the next group is scattered), MAX-reduce of the cached
  occupancy densities every 20 steps.
* `GraphedTrainStep`: the whole step captured into a HIP graph and replayed.
"""
import ctypes as C

import torch
import torch.nn.functional as F

from . import _lib
from .deformers import _opt


_MM_OUT_DTYPE = [None]


class ZeroPool:
    """The zero-initialised fp32 work tensors of ONE training step (field outputs past the live count, the dense weight
    tensor the compositor fills sparsely, the candidate-gradient buffers of its backward, the five loss values) as slices of
    ONE buffer zero-filled by ONE launch: each of these was its own ~5 us fill kernel in a 2 ms step.  `Raymarcher.
    render_train_fused` opens the pool (it knows the sizes), `training_step` closes it; outside a pool -- and for requests
    the pool cannot serve -- `pooled_zeros` is `torch.zeros`."""
    current = None

    def __init__(self, numel, device):
        self.buf = torch.zeros(int(numel), device=device)
        self.off = 0

    def take(self, shape):
        n = 1
        for v in shape:
            n *= int(v)
        if self.off + n > self.buf.numel():
            return None
        out = self.buf[self.off:self.off + n].view(*shape)
        self.off += (n + 63) // 64 * 64          # 256-byte aligned slices
        return out


def pooled_zeros(shape, device):
    shape = tuple(shape) if isinstance(shape, (tuple, list, torch.Size)) else (int(shape),)
    pool = ZeroPool.current
    if pool is not None and pool.buf.device == torch.device(device):
        t = pool.take(shape)
        if t is not None:
            return t
    return torch.zeros(shape, device=device)


def _mm_f32(a, b):
    """fp16 x fp16 -> fp32 GEMM (MFMA, fp32 accumulate and output): aten::mm.dtype where the
    backend provides it, fp32 GEMM otherwise."""
    if _MM_OUT_DTYPE[0] is None:
        try:
            torch.mm(a[:1].contiguous(), b[:, :1].contiguous() if b.dim() == 2 else b, out_dtype=torch.float32)
            _MM_OUT_DTYPE[0] = True
        except Exception:
            _MM_OUT_DTYPE[0] = False
    if _MM_OUT_DTYPE[0]:
        return torch.mm(a, b, out_dtype=torch.float32)
    return a.float() @ b.float()


class _FieldFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, enc_params, col_params, net, n_dev):
        """n_dev: optional device int32[1] live count -- x is then a capacity-sized buffer whose
        tail is never read; outputs past the count are zero."""
        L = _lib.lib()
        xc = x.detach().reshape(-1, 3).float().contiguous()
        V = xc.shape[0]
        stride = L.ia_field_act_stride(net.n_levels)
        acts = torch.empty((V, stride), device=x.device, dtype=torch.float16)
        alloc = pooled_zeros if n_dev is not None else (lambda shp, device: torch.empty(shp, device=device))
        rgb = alloc((V, 3), device=x.device)
        sigma = alloc((V,), device=x.device)
        desc = net.field_desc(V)
        split = desc.enc_split
        desc.enc_split = int(getattr(net, "enc_split_train", split))   # (the training batches' own XCD balance hint: see NeRFNGPNet)
        try:
            _lib.check(L.ia_field_fwd_train(_lib.ptr(xc), V, _lib.ptr(n_dev), C.byref(desc), _lib.ptr(rgb),
                                            _lib.ptr(sigma), _lib.ptr(acts), _lib.stream()), "ia_field_fwd_train")
        finally:
            desc.enc_split = split
        ctx.net = net
        ctx.need_dx = x.requires_grad
        ctx.n_dev = n_dev
        ctx.save_for_backward(xc, acts, rgb)
        return rgb, sigma

    @staticmethod
    def backward(ctx, d_rgb, d_sigma):
        """Gradients of the two parameter vectors are ACCUMULATED IN PLACE into `param.grad` (created
        zero-filled on the first contribution after `zero_grad(set_to_none=True)`) and `None` is returned
        for them: the 52 MB table gradient is written once, by the scatter kernel, into the buffer the
        optimiser reads -- no autograd copy -- and, with several ranks, finished level groups are handed
        to the all-reduce while the remaining groups are still being scattered (parallel.GradReducer).
        Consequence: `torch.autograd.grad(..., params)` is not supported for these two tensors; use
        `.backward()` and read `.grad` (as Lightning's training loop does)."""
        net = ctx.net
        xc, acts, rgb = ctx.saved_tensors
        V = xc.shape[0]
        nf = 2 * net.n_levels
        d_rgb = d_rgb.reshape(V, 3).float().contiguous()
        d_sigma = d_sigma.reshape(V).float().contiguous()
        # gradients are rescaled per call so that their largest magnitude sits at 2^10 before the
        # cast to half (tcnn relies on a fixed 1024x loss scale, DNeRF.py:58); device scalar, no sync, one launch
        L = _lib.lib()
        S = torch.empty(1, device=xc.device)
        state = getattr(net, "_grad_scale_state", None)
        if state is None or state.device != xc.device:
            state = net._grad_scale_state = torch.zeros(2, dtype=torch.int32, device=xc.device)
        _lib.check(L.ia_field_grad_scale(_lib.ptr(rgb), _lib.ptr(d_rgb), _lib.ptr(d_sigma), V, _lib.ptr(ctx.n_dev), _lib.ptr(state),
                                         _lib.ptr(S), _lib.stream()), "ia_field_grad_scale")
        # a frozen parameter vector (requires_grad False: eval.py:70-73 freezes the field and optimises the SMPL tables only)
        # keeps `.grad` None, as autograd would leave it -- the kernels still need somewhere to accumulate: a scratch buffer
        enc_live, col_live = net.encoder.params.requires_grad, net.color_net.params.requires_grad
        g_enc = _grad_buffer(net.encoder.params) if enc_live else _scratch_grad(net, "enc", net.encoder.params)
        g_col = _grad_buffer(net.color_net.params) if col_live else _scratch_grad(net, "col", net.color_net.params)
        n1 = net.sig_w1_size
        if FUSED_MLP_BACKWARD:
            dfeat = torch.empty((V, nf), device=xc.device)
            base_e, base_c = g_enc.data_ptr(), g_col.data_ptr()
            ws = torch.empty(int(L.ia_field_bwd_workspace_bytes(V, net.n_levels)), dtype=torch.uint8, device=xc.device)
            _lib.check(L.ia_field_bwd(_lib.ptr(acts), _lib.ptr(rgb), _lib.ptr(d_rgb), _lib.ptr(d_sigma), V,
                                      _lib.ptr(ctx.n_dev), _lib.ptr(S), C.byref(net.field_desc()), _lib.ptr(dfeat), base_e,
                                      base_e + 4 * n1, base_c, base_c + 4 * 1024, base_c + 4 * 5120, _lib.ptr(ws), ws.numel(),
                                      _lib.stream()), "ia_field_bwd")
        else:
            dfeat = _mlp_backward_gemm(net, acts, rgb, d_rgb, d_sigma, S, g_enc, g_col)
        w_end = n1 + 1024
        dtable = g_enc[w_end:]
        dx = (torch.zeros if ctx.n_dev is not None else torch.empty)((V, 3), device=xc.device) if ctx.need_dx else None
        from . import parallel
        red = parallel.current_reducer()
        last = red is not None and red.active and red.field_backward_done()  # last field call of this step's graph?
        if last and dx is None and not FLAT_ALLREDUCE and enc_live and col_live:
            # level groups, finest first; each finished slice of the table gradient goes to RCCL while the next
            # group is scattered.  The MLP weight gradients travel with the last (small, dense-level) bucket.
            red.reduce_async(g_col)
            for (l0, l1), (lo, hi) in gradient_buckets(net):
                _lib.check(L.ia_hashgrid_bwd_levels(_lib.ptr(xc), V, _lib.ptr(ctx.n_dev), C.byref(net.field_desc()),
                                                    _lib.ptr(dfeat), dtable.data_ptr(), l0, l1, _lib.stream()),
                           "ia_hashgrid_bwd_levels")
                red.reduce_async(g_enc[lo:hi])
        else:
            _lib.check(L.ia_hashgrid_bwd(_lib.ptr(xc), V, _lib.ptr(ctx.n_dev), C.byref(net.field_desc()),
                                         _lib.ptr(dfeat), dtable.data_ptr(), _lib.ptr(dx), _lib.stream()),
                       "ia_hashgrid_bwd")
        return dx, None, None, None, None


def _scratch_grad(net, tag, p):
    """Where the backward kernels accumulate for a FROZEN parameter vector: one persistent buffer per network (its address
    is part of a captured graph), never read."""
    buf = getattr(net, "_frozen_grad_" + tag, None)
    if buf is None or buf.shape != p.shape or buf.device != p.device:
        buf = torch.zeros_like(p, dtype=torch.float32)
        setattr(net, "_frozen_grad_" + tag, buf)
    return buf


def _grad_buffer(p):
    """`p.grad`, created zero-filled on the first contribution of a step."""
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


def gradient_buckets(net, n_groups=4):
    """[((l0, l1), (lo, hi))]: level groups in scatter order (finest first) with the slice [lo, hi) of `encoder.params.grad`
    that is complete once levels [l0, l1) have been scattered.  encoder.params = [W1 | W2 | level 0 | ... | level L-1]; the
    slice of the coarsest group starts at 0, so the MLP weight gradients travel with the last (smallest) bucket.  The slices
    are disjoint and cover the whole vector."""
    off = [int(o) for o in net.hash_desc.offset[:net.n_levels + 1]]
    w_end = net.sig_w1_size + 1024
    out = []
    for l0, l1 in _level_groups(net.n_levels, n_groups):
        lo = w_end + 2 * off[l0] if l0 > 0 else 0
        out.append(((l0, l1), (lo, w_end + 2 * off[l1])))
    return out


def _level_groups(n_levels, n_groups=4):
    """[(l0, l1)] finest first: 16 levels -> (12,16) (8,12) (4,8) (0,4): three 16.8 MB buckets + the dense levels"""
    per = max(n_levels // n_groups, 1)
    cuts = list(range(0, n_levels, per))
    return [(c, min(c + per, n_levels)) for c in reversed(cuts)]


#: True (or IA_FLAT_ALLREDUCE=1): no bucketed / overlapped all-reduce from inside the hash-grid backward -- the whole scatter runs as one
#: launch and `GradReducer.finish` reduces every gradient as one flat collective per tensor afterwards (the fallback, and the
#: reference point the bucketed path is compared against)
FLAT_ALLREDUCE = __import__("os").environ.get("IA_FLAT_ALLREDUCE", "0") == "1"

#: MLP backward through the fused MFMA kernel (`ia_field_bwd`).  False = the GEMM formulation
#: below (kept as the reference the fused kernel is tested against).
FUSED_MLP_BACKWARD = True


def _mlp_backward_gemm(net, acts, rgb, d_rgb, d_sigma, S, g_enc, g_col):
    """The same backward as ten fp16 GEMMs with fp32 accumulation (hipBLASLt): weight gradients
    are [<=64 x V] x [V x <=64] products, the shape the fused kernel exists to avoid."""
    V = acts.shape[0]
    nf = 2 * net.n_levels
    enc_h, col_h = net._half_params()
    h = torch.float16
    W1h, W2h = enc_h[:net.sig_w1_size].view(64, nf), enc_h[net.sig_w1_size:net.sig_w1_size + 1024].view(16, 64)
    Wc1h, Wc2h, Wc3h = col_h[:1024].view(64, 16), col_h[1024:5120].view(64, 64), col_h[5120:6144].view(16, 64)
    feat, h1, o16 = acts[:, :nf], acts[:, nf:nf + 64], acts[:, nf + 64:nf + 80]
    c1, c2 = acts[:, nf + 80:nf + 144], acts[:, nf + 144:nf + 208]
    cin = torch.cat([o16[:, 1:], torch.ones_like(o16[:, :1])], dim=1)  # colour input: out[1:16] + padding 1
    dY32 = torch.zeros((V, 16), device=acts.device)
    dY32[:, :3] = d_rgb * rgb * (1 - rgb)  # sigmoid'
    dY = (dY32 * S).to(h)
    mm = torch.matmul                       # data path: K <= 64, fp16 in/out, fp32 accumulate
    dWc3 = _mm_f32(dY.t(), c2)              # weight gradients: K = V samples -> fp32 output
    dC2 = mm(dY, Wc3h) * (c2 > 0)
    dWc2 = _mm_f32(dC2.t(), c1)
    dC1 = mm(dC2, Wc2h) * (c1 > 0)
    dWc1 = _mm_f32(dC1.t(), cin)
    dcin = mm(dC1, Wc1h)
    dO = torch.cat([(d_sigma * S).to(h)[:, None], dcin[:, :15]], dim=1)
    dW2 = _mm_f32(dO.t(), h1)
    dH1 = mm(dO, W2h) * (h1 > 0)
    dW1 = _mm_f32(dH1.t(), feat)
    inv = 1.0 / S
    dfeat = (_mm_f32(dH1, W1h) * inv).contiguous()
    n1 = net.sig_w1_size
    g_enc[:n1] += (dW1 * inv).reshape(-1)
    g_enc[n1:n1 + 1024] += (dW2 * inv).reshape(-1)
    g_col += torch.cat([(dWc1 * inv).reshape(-1), (dWc2 * inv).reshape(-1), (dWc3 * inv).reshape(-1)])
    return dfeat


def field_autograd(net, x, n_dev=None):
    if torch.is_grad_enabled() and (net.encoder.params.requires_grad or net.color_net.params.requires_grad or x.requires_grad):
        from . import parallel
        red = parallel.current_reducer()
        if red is not None:
            # the LAST field backward of the step starts the bucketed all-reduce; only calls that are recorded into the
            # autograd graph are counted (a no-grad probe has no backward and would keep the count from reaching zero)
            red.field_forward()
    return _FieldFn.apply(x, net.encoder.params, net.color_net.params, net, n_dev)


def _nerf_loss_kernel(rgb, alpha, weight, tgt_rgb, tgt_alpha, w_rgb, w_alpha, w_reg, poison=None, overflow_src=None):
    """`ia_nerf_loss`: (out = {loss, mse_loss, loss_alpha_coarse, reg_alpha, reg_density[, overflow]}, d_rgb, d_alpha, d_weight), flat.
    overflow_src = (device int32 counter [1], capacity): the kernel poisons the step itself when the counter exceeds the capacity
    and reports that as the sixth value."""
    r, a, w = (t.detach().reshape(-1).float().contiguous() for t in (rgb, alpha, weight))
    tr, ta = tgt_rgb.detach().reshape(-1).float().contiguous(), tgt_alpha.detach().reshape(-1).float().contiguous()
    out = pooled_zeros((6 if overflow_src is not None else 5,), r.device)
    d_r, d_a, d_w = torch.empty_like(r), torch.empty_like(a), torch.empty_like(w)
    pz = poison.detach().reshape(()).float().contiguous() if poison is not None else None
    cnt, cap = overflow_src if overflow_src is not None else (None, 0)
    if cnt is not None:
        assert cnt.dtype == torch.int32 and cnt.is_cuda and cnt.numel() >= 1, "overflow counter: device int32"
    _lib.check(_lib.lib().ia_nerf_loss(_lib.ptr(r), _lib.ptr(tr), _lib.ptr(a), _lib.ptr(ta), _lib.ptr(w), a.numel(),
                                       w.numel(), w_rgb, w_alpha, w_reg, _lib.ptr(pz), _lib.ptr(cnt), int(cap), _lib.ptr(out),
                                       _lib.ptr(d_r), _lib.ptr(d_a), _lib.ptr(d_w), _lib.stream()), "ia_nerf_loss")
    return out, d_r, d_a, d_w


class _NeRFLossFn(torch.autograd.Function):
    """Value + gradient of NeRFLoss from one HIP kernel (`ia_nerf_loss`)."""

    @staticmethod
    def forward(ctx, rgb, alpha, weight, tgt_rgb, tgt_alpha, w_rgb, w_alpha, w_reg):
        shapes = rgb.shape, alpha.shape, weight.shape
        out, d_r, d_a, d_w = _nerf_loss_kernel(rgb, alpha, weight, tgt_rgb, tgt_alpha, w_rgb, w_alpha, w_reg)
        ctx.save_for_backward(d_r, d_a, d_w)
        ctx.shapes = shapes
        ctx.mark_non_differentiable(out)
        return out[0], out  # the differentiable scalar + the five reported values

    @staticmethod
    def backward(ctx, g, _g_parts):
        d_r, d_a, d_w = ctx.saved_tensors
        s = ctx.shapes
        if d_r.is_cuda and g.dim() == 0:
            d_r, d_a, d_w = torch._foreach_mul([d_r, d_a, d_w], g)   # one multi-tensor launch instead of three
        else:
            d_r, d_a, d_w = d_r * g, d_a * g, d_w * g
        return d_r.reshape(s[0]), d_a.reshape(s[1]), d_w.reshape(s[2]), None, None, None, None, None


class NeRFLoss(torch.nn.Module):
    """instant_avatar/utils/loss.py:53-77.  On the GPU the loss and its gradient come from one
    HIP kernel; `fused=False` evaluates the same expression with torch ops (the reference the
    kernel is tested against)."""

    def __init__(self, opt=None, fused=True):
        super().__init__()
        self.w_rgb = _opt.get(opt, "w_rgb", 1.0)
        self.w_alpha = _opt.get(opt, "w_alpha", 0.1)
        self.w_reg = _opt.get(opt, "w_reg", 0.1)
        self.fused = fused

    def direct_backward_ok(self, predicts):
        """the loss is exactly the kernel's five terms (no LPIPS / depth term active): `value_and_grads` may replace autograd"""
        return bool(self.fused) and predicts["rgb_coarse"].is_cuda and type(self).forward is NeRFLoss.forward

    def value_and_grads(self, predicts, targets, poison=None, overflow_src=None):
        """The losses (detached) and d loss / d (rgb_coarse, alpha_coarse, weight_coarse) straight from the kernel: the caller
        seeds autograd with them (`torch.autograd.backward(outputs, grads)`) instead of building loss -> mul -> backward out of
        nine small launches.  poison: device scalar, > 0 = NaN loss and gradients; overflow_src = (device int32 counter,
        capacity): the same decided inside the kernel, reported as losses["skipped_overflow"] (see `ia_nerf_loss`)."""
        r, a, w = predicts["rgb_coarse"], predicts["alpha_coarse"], predicts["weight_coarse"]
        out, d_r, d_a, d_w = _nerf_loss_kernel(r, a, w, targets["rgb"], targets["alpha"], float(self.w_rgb), float(self.w_alpha),
                                               float(self.w_reg), poison=poison, overflow_src=overflow_src)
        losses = {"mse_loss": out[1], "loss_alpha_coarse": out[2], "reg_alpha": out[3], "reg_density": out[4], "loss": out[0]}
        if overflow_src is not None:
            losses["skipped_overflow"] = out[5]
        return losses, (d_r.reshape(r.shape), d_a.reshape(a.shape), d_w.reshape(w.shape))

    def forward(self, predicts, targets):
        if self.fused and predicts["rgb_coarse"].is_cuda:
            loss, parts = _NeRFLossFn.apply(predicts["rgb_coarse"], predicts["alpha_coarse"], predicts["weight_coarse"],
                                            targets["rgb"], targets["alpha"], float(self.w_rgb), float(self.w_alpha),
                                            float(self.w_reg))
            return {"mse_loss": parts[1], "loss_alpha_coarse": parts[2], "reg_alpha": parts[3], "reg_density": parts[4],
                    "loss": loss}
        OFFSET = 0.313262
        ent = lambda v: (-torch.log(torch.exp(-v) + torch.exp(v - 1))).mean() + OFFSET
        losses = {"mse_loss": F.mse_loss(predicts["rgb_coarse"], targets["rgb"], reduction="mean"),
                  "loss_alpha_coarse": F.mse_loss(predicts["alpha_coarse"], targets["alpha"]),
                  "reg_alpha": ent(predicts["alpha_coarse"]), "reg_density": ent(predicts["weight_coarse"])}
        losses["loss"] = (self.w_rgb * losses["mse_loss"] + self.w_alpha * losses["loss_alpha_coarse"] +
                          self.w_reg * losses["reg_alpha"] + self.w_reg * losses["reg_density"])
        return losses


class NGPLoss(NeRFLoss):
    """instant_avatar/utils/loss.py:8-50: NeRFLoss + (on patch batches [1, n_patch, P, P, 3]) the LPIPS term and the
    depth-variance regulariser.  LPIPS (`utils.lpips.LPIPS`, VGG-16, v0.1) needs two weight files that cannot be
    fetched offline -- `opt.lpips_lin_weights` (third_parties/lpips/weights/v0.1/vgg.pth of a reference checkout) and
    `opt.lpips_vgg_weights` (torchvision's vgg16 feature weights saved with torch.save) -- or a ready module passed as
    `lpips=`; `w_lpips > 0` without them raises instead of silently training against random features."""

    def __init__(self, opt=None, fused=True, lpips=None):
        super().__init__(opt, fused=fused)
        self.w_lpips = _opt.get(opt, "w_lpips", 0)
        self.w_depth_reg = _opt.get(opt, "w_depth_reg", 0)
        self.lpips = None
        if self.w_lpips > 0:
            if lpips is None:
                lin, vgg = _opt.get(opt, "lpips_lin_weights", None), _opt.get(opt, "lpips_vgg_weights", None)
                if not (lin and vgg):
                    raise NotImplementedError(
                        "NGPLoss: w_lpips > 0 needs the pretrained LPIPS weights, which are not available offline: give "
                        "opt.lpips_lin_weights (third_parties/lpips/weights/v0.1/vgg.pth) and opt.lpips_vgg_weights "
                        "(torch.save(torchvision.models.vgg16(weights='DEFAULT').features.state_dict(), path)), pass a "
                        "loaded utils.lpips.LPIPS as lpips=, or set w_lpips=0")
                from .utils.lpips import LPIPS
                lpips = LPIPS().load_lin_weights(lin).load_trunk_weights(vgg)
            self.lpips = lpips
            for p in self.lpips.parameters():
                p.requires_grad = False                                            # loss.py:12

    def direct_backward_ok(self, predicts):
        patches = predicts["rgb_coarse"].dim() == 5
        extra = patches and (self.w_lpips > 0 or self.w_depth_reg > 0)
        return bool(self.fused) and predicts["rgb_coarse"].is_cuda and not extra and type(self).forward is NGPLoss.forward

    def forward(self, predicts, targets):
        losses = super().forward(predicts, targets)
        patches = predicts["rgb_coarse"].dim() == 5
        if self.w_lpips > 0 and patches:                                           # loss.py:28-32 (BGR order, NCHW, clip)
            pr = predicts["rgb_coarse"][..., [2, 1, 0]].flatten(0, 1).permute(0, 3, 1, 2).clip(max=1)
            tg = targets["rgb"][..., [2, 1, 0]].flatten(0, 1).permute(0, 3, 1, 2)
            lp = self.lpips.to(pr.device)(pr.float(), tg.float()).sum()
            losses["loss_lpips"] = lp
            losses["loss"] = losses["loss"] + self.w_lpips * lp
        if self.w_depth_reg > 0 and patches:                                       # loss.py:33-39
            a, d = predicts["alpha_coarse"], predicts["depth_coarse"]
            alpha_sum = a.sum(dim=(-1, -2))
            depth_avg = (d * a).sum(dim=(-1, -2)) / (alpha_sum + 1e-3)
            reg = (a * (d - depth_avg[..., None, None]).abs()).mean()
            losses["loss_depth_reg"] = reg
            losses["loss"] = losses["loss"] + self.w_depth_reg * reg
        return losses


def configure_optimizer(model, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, smpl_lr=5e-4):
    """DNeRFModel.configure_optimizers (DNeRF.py:32-59): one Adam; hash encoding, the rest, and -- when the model
    carries a `SMPL_param` embedding (optimize_SMPL.enable) -- the SMPL tables with their own learning rate."""
    enc, rest, body = [], [], []
    for name, p in model.named_parameters():
        if name.startswith("loss_fn"):
            continue
        if name.startswith("SMPL_param"):
            body.append(p)
        else:
            (enc if "encoder" in name else rest).append(p)
    # always three groups, the SMPL one possibly empty: that is what DNeRF.py:46-50 builds, and an optimiser state saved
    # by the reference (Lightning's `optimizer_states`) only loads into an optimiser with the same number of groups
    groups = [{"params": enc}, {"params": rest}, {"params": body, "lr": smpl_lr}]
    if enc and enc[0].is_cuda:
        # one C-ABI call per step for all groups: non-finite check, Adam, fp16 copy of the field parameters, gradient zero-fill
        # (optim.FusedAdam -> ia_adam_step; same state / state_dict layout as torch.optim.Adam)
        from .optim import FusedAdam
        opt = FusedAdam(groups, lr=lr, betas=betas, eps=eps)
        net = getattr(model, "net_coarse", None)
        if net is not None and hasattr(net, "half_shadow"):
            opt.register_shadow(net.encoder.params, lambda: net.half_shadow(0))
            opt.register_shadow(net.color_net.params, lambda: net.half_shadow(1))
            opt._shadow_owner = net
    else:   # (no GPU: the host-side tests of the training loop and of the multi-rank plumbing)
        opt = torch.optim.Adam(groups, lr=lr, betas=betas, eps=eps)
    return opt


def configure_scheduler(optimizer, max_epochs):
    """DNeRF.py:52-55: LambdaLR(lambda k: (1 - k / max_epochs) ** 1.5).  WHEN it is stepped: the reference optimises manually
    (`automatic_optimization = False`, DNeRF.py:20), so Lightning never steps the scheduler itself; the only call is
    `on_validation_epoch_end` (DNeRF.py:163-166) -- once per VALIDATION run, i.e. every `check_val_every_n_epoch` (10 in every
    shipped config) epochs, and k counts validation runs, not epochs.  With SNARF_NGP.yaml (30 epochs) the learning rate ends
    at (1 - 3/30)^1.5 = 0.85 of its start.  The drivers (`drivers/train.fit`, `drivers/fit.fit_sequence`, `drivers/eval`) step
    it at exactly those epochs."""
    return torch.optim.lr_scheduler.LambdaLR(optimizer, lambda epoch: (1 - min(epoch, max_epochs) / max_epochs) ** 1.5)


def _non_finite_flag(params):
    """float32 device scalar: 1.0 when any gradient holds an inf / NaN (no host synchronisation).  One
    multi-tensor launch over all gradients -- the kernel GradScaler.unscale_ uses, with a scale of 1 (the
    reference's `self.scaler.step`, DNeRF.py:151-154, runs exactly this check before the optimiser)."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return None
    dev = grads[0].device
    flag = torch.zeros((), device=dev)
    one = _ONES.get(dev)
    if one is None:
        one = _ONES[dev] = torch.ones((), device=dev)
    by_dev = {}
    for g in grads:
        by_dev.setdefault((g.device, g.dtype), []).append(g)
    for (d, _), gs in by_dev.items():
        if d == dev:
            torch._amp_foreach_non_finite_check_and_unscale_(gs, flag, one)
        else:  # parameters on another device (not the case in the shipped configurations)
            f2 = torch.zeros((), device=d)
            torch._amp_foreach_non_finite_check_and_unscale_(gs, f2, torch.ones((), device=d))
            flag = torch.maximum(flag, f2.to(dev))
    return flag


_ONES = {}


def optimizer_step_skip_non_finite(optimizer, params, extra_flag=None):
    """GradScaler.step's inf/NaN skip (DNeRF.py:151-154: `self.scaler.step(optimizer)`) without its host
    synchronisation: torch's fused Adam takes a device-side `found_inf` flag and leaves parameters, moments
    and step counters untouched when it is set.  One non-finite gradient would otherwise poison the whole
    52 MB table (the per-call gradient scale S = 1024 / amax turns a single NaN into NaN everywhere).
    Non-fused optimisers (CPU tests) fall back to a host check.  Returns the flag (device scalar), which also covers
    `extra_flag` (device scalar, optional): a step whose training render dropped candidates is skipped the same way."""
    from .optim import FusedAdam
    if isinstance(optimizer, FusedAdam):
        optimizer.step(skip_flag=extra_flag)
        return optimizer.found_inf
    flag = _non_finite_flag(params)
    if extra_flag is not None and flag is not None:
        # a second device-side reason to skip the update (a training render whose candidates overflowed their capacity)
        flag = torch.maximum(flag, extra_flag.to(flag))
    fused = all(g.get("fused") for g in optimizer.param_groups)
    if flag is None:
        optimizer.step()
        return flag
    if fused and flag.is_cuda:
        optimizer.grad_scale, optimizer.found_inf = None, flag
        try:
            optimizer.step()
        finally:
            optimizer.grad_scale, optimizer.found_inf = None, None
    elif not bool(flag):
        optimizer.step()
    return flag


def all_reduce_grads(model, world_size, reducer=None):
    """Data-parallel gradient averaging.  With a `parallel.GradReducer` that was active during backward the
    big buckets are already in flight (started from inside the hash-grid backward); this reduces what is
    left, waits, and turns sums into means.  Without one: a fresh reducer, i.e. one all-reduce per
    parameter tensor (the 52 MB hash-table gradient as ONE flat bucket; xGMI is point-to-point, so few
    large collectives beat many small ones)."""
    from .parallel import GradReducer, collectives_on
    if not collectives_on(world_size):
        return
    (reducer or GradReducer(world_size)).finish([p for p in model.parameters()])


def update_density_grid(model, world_size=1, jitter=None, differentiable=True):
    """DNeRFModel.update_density_grid (DNeRF.py:99-110); with several ranks the cached densities are
    MAX-reduced between the EMA update and the thresholding (DensityGrid.update's reduce hook), so every
    rank thresholds -- and regularises with -- the same field, once.
    differentiable=False: the probe runs without recording an autograd graph (the refine configuration drops the
    regulariser, DNeRF.py:137: the graph over 262 144 probe points would be built and thrown away)."""
    N = 1 if getattr(model.renderer, "smpl_init", False) else 20   # DNeRF.py:100
    if model.global_step % N != 0:
        return None
    grid = model.renderer.density_grid_train
    hook = None
    from .parallel import collectives_on
    if collectives_on(world_size):
        from .parallel import reduce_density_cache
        hook = lambda cached: reduce_density_cache(cached, world_size)
    density, valid = grid.update(model.deformer, model.net_coarse, model.global_step, reduce_hook=hook, jitter=jitter,
                                 differentiable=differentiable)
    inv = (~valid).to(density.dtype)   # mean over the cells outside the grid, without a boolean-mask gather (host sync)
    reg = N * (density * inv).sum() / inv.sum().clamp(min=1.0)
    if model.global_step < 500:
        reg = reg + 0.5 * density.mean()
    return reg


def training_step(model, batch, optimizer, loss_fn, world_size=1, is_refine=False, _capturing=False, draws=None):
    """DNeRFModel.training_step (DNeRF.py:112-161), all four configurations: plain (SNARF_NGP.yaml), the fit stage
    (SMPLDeformer + SMPLParamEmbedding) and refinement (SNARF_NGP_refine.yaml: SNARFDeformer + SMPLParamEmbedding,
    `is_refine` -- no sigma noise, no density regulariser; the gradient reaches the SMPL tables through tfs by the
    implicit differentiation of the roots, on the fused route).
    `_capturing`: the call is being recorded into a HIP graph (GraphedTrainStep): nothing executes, so the
    host-side step counter is left alone.
    `draws`: optional dict of injected random tensors for reproducible tests -- `grid_jitter` [64,64,64,3]
    (density_grid.py:47), `ray_jitter` [n_rays, MAX_SAMPLES] (raymarcher_acc.py:156), `noise` [n_rays, MAX_SAMPLES] (:167)."""
    draws = draws or {}
    from . import parallel
    if getattr(model, "SMPL_param", None) is not None:            # DNeRF.py:113-128 (optimize_SMPL.enable)
        batch = dict(batch)
        idx_dev = batch.get("idx_dev")     # device copy of the frame index when the data side provides one (no H2D copy per step)
        if idx_dev is None:
            idx_dev = batch["idx"].reshape(-1).long().to(model.SMPL_param.betas.weight.device)
        body_params = model.SMPL_param(idx_dev.reshape(-1).long())
        for k in ("global_orient", "body_pose", "transl"):
            batch[k] = body_params[k]
        from .deformers.smpl_deformer import SMPLDeformer
        if isinstance(model.deformer, SMPLDeformer):
            batch["betas"] = body_params["betas"]
        # near / far follow the refined translation (DNeRF.py:124-127: |transl| -/+ 1 for every ray): one launch
        tr = batch["transl"].detach().reshape(-1)[:3].float().contiguous()
        if tr.is_cuda and batch["near"].is_cuda:
            near, far = torch.empty_like(batch["near"], dtype=torch.float32).contiguous(), torch.empty_like(batch["far"], dtype=torch.float32).contiguous()
            _lib.check(_lib.lib().ia_near_far(_lib.ptr(tr), near.numel(), _lib.ptr(near), _lib.ptr(far), _lib.stream()), "ia_near_far")
            batch["near"], batch["far"] = near, far
        else:
            dist = torch.norm(batch["transl"], dim=-1, keepdim=True).detach()
            batch["near"] = (dist - 1).reshape(1, *([1] * (batch["near"].dim() - 1))).expand_as(batch["near"]).contiguous()
            batch["far"] = (dist + 1).reshape(1, *([1] * (batch["far"].dim() - 1))).expand_as(batch["far"]).contiguous()
    model.renderer.idx = int(batch["idx"][0]) if "idx" in batch else 0
    from .deformers.snarf_deformer import SNARFDeformer
    prep = model.deformer.prepare_deformer
    if type(model.deformer) is SNARFDeformer and getattr(prep, "__func__", None) is SNARFDeformer.prepare_deformer:
        prep(batch, want_bbox=False)   # no consumer of the deformed-voxel box in a training step (computed on demand otherwise)
    else:                              # (another deformer plugin, or a caller's wrapper around the method)
        prep(batch)
    reducer = parallel.GradReducer(world_size)
    parallel.set_current_reducer(reducer if reducer.active else None)
    try:
        reg = update_density_grid(model, world_size, jitter=draws.get("grid_jitter"), differentiable=not is_refine)
        model.net_coarse.initialize(model.deformer.bbox)
        use_noise = model.global_step < 1000 and not is_refine
        model.renderer.train_draws = draws if draws else None
        try:
            predicts = model.forward(batch, eval_mode=False, noise=1 if use_noise else 0)
        finally:
            model.renderer.train_draws = None
        overflow = getattr(model.renderer, "train_overflow_flag", None)
        # (the fused render leaves the SOURCE of the flag -- its device-side candidate counter and the capacity of its buffers --
        #  instead of a flag tensor: the loss kernel of the direct path compares them itself, two launches less per step)
        overflow_src = getattr(model.renderer, "train_overflow_src", None) if overflow is None else None
        with_reg = reg is not None and not is_refine
        direct = (not with_reg) and hasattr(loss_fn, "direct_backward_ok") and loss_fn.direct_backward_ok(predicts) and \
            all(torch.is_tensor(predicts.get(k)) and predicts[k].requires_grad for k in ("rgb_coarse", "alpha_coarse", "weight_coarse"))
        # a render that dropped candidates (capacity overflow) must not update anything, on ANY rank: its loss is turned
        # into NaN before the backward pass, so every gradient of this rank is NaN, the gradient average carries that to
        # all ranks, and the ordinary non-finite check skips the step everywhere -- no extra collective, no host read
        if direct:
            # the loss kernel already holds d loss / d (rgb, alpha, weights): seed autograd with them (and let the kernel do the
            # NaN poisoning) instead of loss -> where -> mul -> backward -> foreach_mul: nine small launches less per step
            losses, grads = loss_fn.value_and_grads(predicts, batch, poison=overflow, overflow_src=overflow_src)
            if overflow_src is not None:
                overflow = losses["skipped_overflow"]     # (device scalar written by the loss kernel; also fed to the optimiser below)
        else:
            if overflow_src is not None:
                overflow = (overflow_src[0].reshape(-1)[0] > overflow_src[1]).to(torch.float32).reshape(())
            losses = loss_fn(predicts, batch)
            if with_reg:
                losses["reg"] = reg
                losses["loss"] = losses["loss"] + reg
        if not getattr(optimizer, "grads_zeroed", False):   # (a FusedAdam step with fused_zero_grad left every buffer zero-filled)
            optimizer.zero_grad(set_to_none=True)
        if direct:
            torch.autograd.backward([predicts["rgb_coarse"], predicts["alpha_coarse"], predicts["weight_coarse"]], list(grads))
        else:
            total = losses["loss"]
            if overflow is not None:
                total = total * torch.where(overflow > 0, torch.full_like(overflow, float("nan")), torch.ones_like(overflow))
            total.backward()
        all_reduce_grads(model, world_size, reducer)
    finally:
        parallel.set_current_reducer(None)
        ZeroPool.current = None
    params = [p for g in optimizer.param_groups for p in g["params"]]
    model.renderer.train_overflow_flag = None
    model.renderer.train_overflow_src = None
    losses["skipped_non_finite"] = optimizer_step_skip_non_finite(optimizer, params, extra_flag=overflow)
    if overflow is not None:
        losses["skipped_overflow"] = overflow
    if hasattr(model.net_coarse, "mark_updated"):
        # refresh the fp16 shadow + MFMA fragments on next use (the fused optimiser step has already written the shadow of
        # the network it was configured for: only the fragment image is rebuilt then)
        if getattr(optimizer, "_shadow_owner", None) is model.net_coarse:
            model.net_coarse.mark_updated(shadow_fresh=True)
        else:
            model.net_coarse.mark_updated()
    if not _capturing:
        model.global_step += 1
    # nothing the caller gets keeps the autograd graph of this step alive (see SNARFDeformer.release_graph)
    if hasattr(model.deformer, "release_graph"):
        model.deformer.release_graph()
    return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in losses.items()}


class GraphedTrainStep:
    """`training_step` replayed from a captured HIP graph (torch.cuda.CUDAGraph).

    A training step has no host synchronisation and fixed launch geometry (sample and candidate counts live on
    the device, buffers are capacity-sized), but it is ~60 launches and a millisecond of Python: measured on
    MI355X the host needs 1.1-1.3 ms to enqueue a step whose kernels take 0.9 ms.  The whole step -- deformer
    preparation, march, search, field forward, compositing, loss, backward, non-finite check, fused Adam -- is
    captured once and replayed; the batch is copied into static input tensors (`inputs`, which callers may
    also fill in place) and the loss dictionary is returned as static output tensors (overwritten by the
    next call: clone what must be kept).

    Run eagerly, through `training_step`, are: steps that update the occupancy grid (every 20th, DNeRF.py:100;
    their regulariser changes the autograd graph, and with several ranks the cached densities are MAX-reduced) and any batch
    whose tensor shapes differ from the captured ones.  With several ranks the bucketed all-reduce is part of the captured
    graph (`graph_collectives`); every rank replays / runs eagerly at the same steps, so the collectives stay matched.  Models with a `SMPL_param` embedding (fit stage, refinement) are captured
    too: the frame index reaches the embedding tables as the device tensor `idx_dev` of the batch.  One graph is
    held per (noise on/off, capacity) state; a candidate-capacity overflow (renderer.train_overflow) drops
    the graphs so that they are captured again with the grown capacity.  Learning rates are turned into
    device tensors, so that an `lr_scheduler` keeps working across replays."""

    #: captured graphs kept alive (one per (noise, capacity, per-frame grid) state; each holds its own activation buffers)
    max_graphs = 256

    def __init__(self, model, optimizer, loss_fn, world_size=1, is_refine=False, enabled=True, graph_collectives=None):
        """graph_collectives: capture the step of a multi-rank job too -- the bucketed RCCL all-reduce is recorded into the
        graph together with the kernels (collectives are capturable), so that an N-rank step is launched the same way as a
        1-rank step instead of ~60 eager launches behind a millisecond of Python.  Default: the environment variable
        IA_GRAPH_COLLECTIVES (1 = on), else on."""
        import os
        from . import parallel
        self.model, self.optimizer, self.loss_fn = model, optimizer, loss_fn
        self.world_size, self.is_refine = world_size, is_refine
        if graph_collectives is None:
            graph_collectives = os.environ.get("IA_GRAPH_COLLECTIVES", "1") != "0"
        self.graph_collectives = bool(graph_collectives)
        self.enabled = bool(enabled) and (not parallel.collectives_on(world_size) or self.graph_collectives)
        if getattr(model, "SMPL_param", None) is not None:
            # SMPL parameters under optimisation: capturable on the fused SNARF route only (the SMPLDeformer's training
            # query -- fit stage -- reads validity counts on the host and inverts 6 890 vertex transforms with the LU library)
            fused = getattr(model.deformer, "fused_train_route", None)
            inner = getattr(model.deformer, "deformer", None)      # SNARFDeformer's ForwardDeformer (version 2 + SMPL tables: dense route)
            from .deformers.smpl_deformer import SMPLDeformer
            if isinstance(model.deformer, SMPLDeformer):
                from .deformers import smpl_deformer as _sdm
                # the fit stage (fit.py): capturable since round 6 (fused body model + compact render); one rank only -- every rank
                # walks its own frames, so per-frame captures would happen at different steps on different ranks
                self.enabled = self.enabled and fused is not None and _sdm.FUSED_LBS and not parallel.collectives_on(world_size)
            else:
                self.enabled = self.enabled and fused is not None and inner is not None and getattr(inner, "version", 1) == 1
        from .optim import FusedAdam
        if isinstance(optimizer, FusedAdam) and self.enabled:
            optimizer.fused_zero_grad = True    # the step leaves every gradient buffer zero-filled for the next one
        self.graphs = {}
        self.inputs = None
        self.replays = 0
        self.eager_steps = 0
        self.capture_error = None
        self._agreed = False   # the ranks have agreed on the outcome of the first capture (see __call__)
        self._warmed = False   # an eager step has run through this object (lazy initialisation done)

    # -- helpers ---------------------------------------------------------------------------------------
    def _update_period(self):
        return 1 if getattr(self.model.renderer, "smpl_init", False) else 20

    def _signature(self, batch):
        return tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(batch.items()) if torch.is_tensor(v) and v.is_cuda)

    def _make_capturable(self):
        from .optim import FusedAdam
        if isinstance(self.optimizer, FusedAdam):
            self.optimizer.capturable_lr()
            return
        for g in self.optimizer.param_groups:
            if not g.get("fused"):
                raise RuntimeError("GraphedTrainStep needs the fused Adam of configure_optimizer")
            g["capturable"] = True
            if not torch.is_tensor(g["lr"]):
                dev = g["params"][0].device if g["params"] else self.optimizer.param_groups[0]["params"][0].device
                g["lr"] = torch.tensor(float(g["lr"]), device=dev)
                if "initial_lr" in g and not torch.is_tensor(g["initial_lr"]):
                    g["initial_lr"] = float(g["initial_lr"])

    def _capture(self, key, use_noise):
        m, r = self.model, self.model.renderer
        self._make_capturable()
        if hasattr(m.net_coarse, "mark_updated"):
            # the refresh of the kernel-side weight copies belongs to every replay: the MFMA fragment image only when the
            # optimiser step writes the fp16 shadow itself (FusedAdam), both fp32 -> fp16 casts otherwise
            if getattr(m.net_coarse, "_dirty", False) is True:
                m.net_coarse.refresh()          # (a pending FULL refresh -- checkpoint load, broadcast -- runs now, eagerly)
            if getattr(self.optimizer, "_shadow_owner", None) is m.net_coarse:
                m.net_coarse.mark_updated(shadow_fresh=True)
            else:
                m.net_coarse.mark_updated()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        r._graph_capture = True
        try:
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                out = training_step(m, self.inputs, self.optimizer, self.loss_fn, self.world_size, self.is_refine, _capturing=True)
        finally:
            r._graph_capture = False
        params = [p for g in self.optimizer.param_groups for p in g["params"]]
        entry = dict(graph=graph, out=out, grads=[p.grad for p in params], params=params, cap=r.train_cand_capacity)
        self.graphs[key] = entry
        return entry

    # -- the step --------------------------------------------------------------------------------------
    def __call__(self, batch=None):
        import os
        from . import parallel
        m, r = self.model, self.model.renderer
        if batch is None:
            if self.inputs is None:
                raise ValueError("GraphedTrainStep(): no batch given and no static inputs yet (pass the first batches explicitly)")
            batch = self.inputs
        eager = (not self.enabled) or m.global_step % self._update_period() == 0
        if not eager and torch.is_tensor(batch.get("idx")) and batch["idx"].is_cuda:
            eager = True   # the frame index is read on the host (DNeRF.py:113); keep it a host tensor for graph replay
        if not eager and getattr(m, "SMPL_param", None) is not None and not torch.is_tensor(batch.get("idx_dev")):
            batch = dict(batch)   # the embedding row is looked up on the device: give the static inputs an `idx_dev` slot
            batch["idx_dev"] = torch.full((1,), int(batch["idx"][0]), dtype=torch.long, device=m.SMPL_param.betas.weight.device)
        if not eager and self.inputs is not None and batch is not self.inputs:
            eager = self._signature(batch) != self._sig
        if eager:
            if self._agreed and parallel.collectives_on(self.world_size) and m.global_step % self._update_period() == 0:
                # occupancy-update steps run eagerly on EVERY rank at the same global steps: the place to agree on a capture
                # that failed on one rank only AFTER the first agreement (a re-capture with a grown candidate capacity is a
                # per-rank event) -- from here on all ranks launch eagerly together instead of one rank's eager collectives
                # meeting the others' replays for the rest of the run
                import torch.distributed as dist
                flag = torch.tensor([1.0 if self.enabled else 0.0], device=next(m.parameters()).device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if float(flag.item()) == 0.0 and self.enabled:
                    self.enabled = False
                    self.capture_error = self.capture_error or "HIP-graph re-capture failed on another rank: all ranks launch eagerly"
            self.eager_steps += 1
            self._warmed = True
            return training_step(m, batch, self.optimizer, self.loss_fn, self.world_size, self.is_refine)
        if self.inputs is None:
            self.inputs = {k: (v.clone() if torch.is_tensor(v) and v.is_cuda else v) for k, v in batch.items()}
            go, bp = self.inputs.get("global_orient"), self.inputs.get("body_pose")
            if torch.is_tensor(go) and torch.is_tensor(bp) and go.is_cuda and go.numel() == 3 and bp.numel() == 69 and go.dtype == bp.dtype == torch.float32:
                # the two static pose inputs as the halves of ONE 72-float record (see snarf_deformer._pose72: used in place by the
                # joint-chain kernel, one concatenation launch less in the captured step)
                rec = torch.cat([go.reshape(-1), bp.reshape(-1)])
                self.inputs["global_orient"], self.inputs["body_pose"] = rec[:3].view(go.shape), rec[3:].view(bp.shape)
            self._sig = self._signature(batch)
        elif batch is not self.inputs:
            for k, v in batch.items():
                dst = self.inputs.get(k)
                if torch.is_tensor(dst) and dst.is_cuda:
                    if v is not dst:
                        dst.copy_(v, non_blocking=True)
                else:
                    self.inputs[k] = v
        if "idx" in self.inputs:
            r.idx = int(self.inputs["idx"][0])
        # deferred look at the counts of an earlier step: an overflow grows the capacity -> capture again
        r._train_counts_check()                      # (pending event of a preceding eager step, if any)
        r._train_counts_peek(r.train_cand_capacity)
        use_noise = m.global_step < 1000 and not self.is_refine
        # (a renderer with one occupancy grid PER FRAME -- Raymarcher(smpl_init=True).initialize(N_frames), raymarcher_acc.py:66-70 -- bakes
        # the frame's grid into the captured launches: one graph per frame then, at most `max_graphs` of them alive)
        grid_key = int(r.idx) if len(getattr(r, "density_grid_train_all", ())) > 1 else 0
        key = (bool(use_noise), r.train_cand_capacity, grid_key)
        entry = self.graphs.get(key)
        if entry is None and not self._warmed:
            # one eager step on these inputs before the first capture of this state: lazy initialisation (pinned counter
            # buffers, workspace growth, library handles, the optimiser's state tensors) must not land inside a capture --
            # it would either fail it or bake one-off allocations into the graph's pool
            # (a resume at a global_step that is not a multiple of the update period gets here; normally step 0 -- an
            # occupancy-update step, eager by rule -- has already done it)
            self._warmed = True
            self.eager_steps += 1
            return training_step(m, self.inputs, self.optimizer, self.loss_fn, self.world_size, self.is_refine)
        if entry is None:
            self.graphs = {k: e for k, e in self.graphs.items() if k[1] == r.train_cand_capacity}
            while len(self.graphs) >= self.max_graphs:          # oldest first (dicts keep insertion order)
                self.graphs.pop(next(iter(self.graphs)))
            err = None
            try:
                entry = self._capture(key, use_noise)
            except Exception as e:  # capture not possible on this stack: stay eager (same kernels, host-launched)
                err = repr(e)[:300]
            if not self._agreed and parallel.collectives_on(self.world_size):
                # The FIRST capture happens at the same step on every rank (step 0 is an occupancy update, the warm-up rules
                # do not depend on the rank): agree on its outcome, once, outside any capture.  One rank whose capture failed
                # would otherwise launch its collectives eagerly against the other ranks' replays for the rest of the run --
                # order-compatible by construction, but a mix nothing has exercised on 8 GPUs.  All ranks captured, or
                # all ranks run eagerly.  (Later re-captures -- a candidate-capacity overflow is a per-rank event -- cannot
                # be agreed on: the other ranks are inside a replay then; a failure there falls back on that rank alone.)
                self._agreed = True
                import torch.distributed as dist
                flag = torch.tensor([0.0 if err else 1.0], device=next(m.parameters()).device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if float(flag.item()) == 0.0 and err is None:
                    err = "HIP-graph capture failed on another rank: all ranks launch eagerly"
                    self.graphs.pop(key, None)
            if err is not None:
                self.capture_error = err
                self.enabled = False
                import warnings
                warnings.warn("GraphedTrainStep: HIP-graph capture failed, training continues with eager launches (slower): " + self.capture_error)
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
                self.eager_steps += 1
                return training_step(m, batch, self.optimizer, self.loss_fn, self.world_size, self.is_refine)
        if getattr(m.net_coarse, "_dirty", False) is True and getattr(self.optimizer, "_shadow_owner", None) is m.net_coarse:
            m.net_coarse.refresh()   # weights changed behind the optimiser's back (checkpoint load, broadcast): the captured step only
                                     # rebuilds the fragment image, the fp16 copy is brought up to date here
        entry["graph"].replay()
        self.replays += 1
        for p, g in zip(entry["params"], entry["grads"]):
            p.grad = g
        if hasattr(m.net_coarse, "mark_updated"):
            if getattr(self.optimizer, "_shadow_owner", None) is m.net_coarse:
                m.net_coarse.mark_updated(shadow_fresh=True)
            else:
                m.net_coarse.mark_updated()
        m.global_step += 1
        return entry["out"]
