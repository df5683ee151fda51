"""The routes for callables that are NOT the native (SNARFDeformer, NeRFNGPNet) pair -- in ONE place.

The plugin surface takes an arbitrary `model(pts, _) -> (rgb, sigma)` closure (DNeRF.py:66-67 builds one per
call); the fused kernels only serve the pair they recognise.  Everything else lands here: the same HIP kernels the C ABI
exports one by one (`ia_raymarch_test`, `ia_composite_test`, `ia_raymarch_train`, the unfused Broyden search), driven from
the host with DENSE intermediate tensors -- slot-padded samples, [P, 13] candidate blocks -- and the callable in between.
These routes synchronise with the host every wave-front iteration and are several times slower than the fused ones; they
exist (a) so that a user's own field or deformer keeps working behind the plugin classes, and (b) as the independent
cross-check of the fused routes in tests/ (test_closure_route_equals_fused_route, test_refine_fused_route_equals_dense_torch_route).

Semantics (what the results must equal): raymarcher_acc.py:83-138 (test loop), :140-186 with :25-36 (training render),
snarf_deformer.py:127-159 (candidate reduction).  There is no CPU path here either: the tensors must live on the GPU.
"""
import ctypes as C

import torch

from . import _lib


def _flat_rays(rays):
    f = lambda t, w: t.reshape(-1, w).float().contiguous() if w > 1 else t.reshape(-1).float().contiguous()
    return f(rays.o, 3), f(rays.d, 3), f(rays.near, 1), f(rays.far, 1)


def _outputs(rays, color, depth, alpha, last_key, last):
    return {"rgb_coarse": color.reshape(rays.o.shape), "depth_coarse": depth.reshape(rays.near.shape),
            "alpha_coarse": alpha.reshape(rays.near.shape), last_key: last}


def _masked_field(model, pts, mask, fill_sigma):
    """`model` on the masked entries of a dense block; (rgb, sigma) of the block's shape, `fill_sigma` / zero colour elsewhere"""
    rgb = torch.zeros(pts.shape, dtype=torch.float32, device=pts.device)
    sigma = torch.full(pts.shape[:-1], float(fill_sigma), dtype=torch.float32, device=pts.device)
    if bool(mask.any()):                      # host read: these routes are not capturable
        r, s = model(pts[mask], None)
        rgb = rgb.masked_scatter(mask[..., None].expand_as(rgb), r.float())
        sigma = sigma.masked_scatter(mask, s.float())
    return rgb, sigma


@torch.no_grad()
def render_test(renderer, rays, model, bg_color):
    """Test-time wave front around an arbitrary callable.  Per iteration: every alive ray marches N_step =
    clamp(MAX_BATCH_SIZE // alive, 1, MAX_SAMPLES) slots, the callable sees the occupied ones, the compositor folds the
    block into the rays' running colour / depth / transmittance, rays that ran out of slots or of transmittance retire."""
    L = _lib.lib()
    _lib.require_cuda(rays.o)
    o, d, near, far = _flat_rays(rays)
    near = near.clone()                                           # advanced in place by the marcher
    n, dev, S = o.shape[0], o.device, renderer.MAX_SAMPLES
    color, depth = torch.zeros((n, 3), device=dev), torch.zeros(n, device=dev)
    trans, counter = torch.ones(n, device=dev), torch.zeros(n, device=dev)
    step = ((far - near) / S).contiguous()
    grid = renderer.density_grid_test
    occ = renderer._occ_desc(grid)
    alive = torch.arange(n, device=dev)
    done = 0
    while done < S and alive.numel() > 0:
        a = alive.numel()
        n_step = max(min(renderer.MAX_BATCH_SIZE // a, S), 1)
        pts = torch.empty((a, n_step, 3), device=dev)
        delta, z = torch.empty((a, n_step), device=dev), torch.empty((a, n_step), device=dev)
        _lib.check(L.ia_raymarch_test(_lib.ptr(o), _lib.ptr(d), _lib.ptr(near), _lib.ptr(far), _lib.ptr(alive), a, _lib.ptr(grid.occ_bits),
                                      C.byref(occ), _lib.ptr(step), n_step, _lib.ptr(pts), _lib.ptr(delta), _lib.ptr(z), _lib.stream()),
                   "ia_raymarch_test")
        hit = delta > 0
        counter[alive] += hit.sum(dim=-1)
        rgb, sigma = _masked_field(model, pts, hit, 0.0)
        _lib.check(L.ia_composite_test(_lib.ptr(rgb.contiguous()), _lib.ptr(sigma.contiguous()), _lib.ptr(delta), _lib.ptr(z), _lib.ptr(alive), a,
                                       n_step, _lib.ptr(color), _lib.ptr(depth), _lib.ptr(trans), 0.01, _lib.stream()), "ia_composite_test")
        alive = alive[(trans[alive] > 1e-4) & (z[:, -1] > 0)]
        done += n_step
    color = color + trans[:, None] * (bg_color.reshape(-1, 3) if bg_color is not None else 1.0)
    return _outputs(rays, color, depth, 1 - trans, "counter_coarse", counter.reshape(rays.near.shape))


def composite_train(sigma, dists):
    """training compositing: alpha = 1 - exp(-relu(sigma) dt), T = cumprod(1 - alpha + 1e-10), weights = alpha T (differentiable)"""
    alpha = 1.0 - torch.exp(-torch.relu(sigma) * dists)
    trans = torch.cat([torch.ones_like(alpha[..., :1]), torch.cumprod(1 - alpha + 1e-10, dim=-1)], dim=-1)
    return alpha * trans[..., :-1], trans[..., -1]


def render_train(renderer, rays, model, noise, bg_color):
    """Training render with MAX_SAMPLES slots per ray (dense): marcher -> jitter -> callable on the occupied slots ->
    (+ sigma noise) -> differentiable compositing in torch ops.  `renderer.train_draws` may inject the random draws."""
    L = _lib.lib()
    _lib.require_cuda(rays.o)
    o, d, near, far = _flat_rays(rays)
    n, S = o.shape[0], renderer.MAX_SAMPLES
    step = ((far - near) / S).contiguous()
    grid = renderer.density_grid_train
    occ = renderer._occ_desc(grid)
    z = torch.empty((n, S), device=o.device)
    with torch.no_grad():
        _lib.check(L.ia_raymarch_train(_lib.ptr(o.detach()), _lib.ptr(d.detach()), _lib.ptr(near.detach()), _lib.ptr(far.detach()), n,
                                       _lib.ptr(grid.occ_bits), C.byref(occ), _lib.ptr(step.detach()), S, _lib.ptr(z), _lib.stream()), "ia_raymarch_train")
    occupied = z > 0
    draws = getattr(renderer, "train_draws", None) or {}
    draw = lambda key, make: draws[key].to(z).reshape(z.shape) if key in draws else make(z)
    z = z + draw("ray_jitter", torch.rand_like) * step[:, None]
    pts = z[..., None] * d[:, None] + o[:, None]
    rgb, sigma = _masked_field(model, pts, occupied, -1e3)
    if noise > 0:
        sigma = sigma + noise * draw("noise", torch.randn_like)
    weights, t_end = composite_train(sigma, step[:, None].expand_as(sigma))
    color = (weights[..., None] * rgb).sum(dim=-2) + t_end[..., None] * (bg_color.reshape(-1, 3) if bg_color is not None else 1.0)
    return _outputs(rays, color, (weights * z).sum(dim=-1), weights.sum(-1), "weight_coarse", weights.reshape(*rays.near.shape, -1))


def deform_query(deformer, pts, model, eval_mode):
    """Candidate reduction around an arbitrary field: all 13 Broyden candidates of every point as a dense [P, 13, 3] block
    (`deformer.deform`), the callable on the valid ones, the largest sigma per point wins and takes its colour along.
    Invalid candidates count as sigma 0 at test time (NaN / inf outputs too) and as -1e5 in training."""
    cand, valid = deformer.deform(pts, eval_mode=eval_mode)
    rgb, sigma = _masked_field(model, cand, valid, 0.0 if eval_mode else -1e5)
    if eval_mode:
        rgb, sigma = torch.nan_to_num(rgb, 0, 0, 0), torch.nan_to_num(sigma, 0, 0, 0)
    best, idx = torch.max(sigma, dim=-1)
    return torch.gather(rgb, 1, idx[:, None, None].expand(-1, 1, 3)).reshape(-1, 3), best.reshape(-1)


def deform_query_single(deformer, pts, model, eval_mode):
    """The one-candidate deformers (SMPLDeformer: nearest-vertex inverse skinning, smpl_deformer.py:112-131) around an arbitrary
    field: the callable on the points that have a canonical position, sigma 0 (test) / -1e5 (training) for the others; in
    training a non-finite output counts as invalid too."""
    cano, valid = deformer.deform(pts)
    rgb, sigma = _masked_field(model, cano, valid, 0.0 if eval_mode else -1e5)
    if not eval_mode:
        ok = torch.isfinite(rgb).all(-1) & torch.isfinite(sigma)
        rgb, sigma = torch.where(ok[:, None], rgb, torch.zeros_like(rgb)), torch.where(ok, sigma, torch.full_like(sigma, -1e5))
    return rgb, sigma
