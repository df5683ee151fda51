"""GPU tests of the training-side kernels (-m gpu): hash-grid/MLP backward against a
plain PyTorch fp32 reference of the same op, and one full training step."""
import numpy as np
import pytest
import torch

from instantavatar_amd import synthetic as syn
from instantavatar_amd.pipeline import make_batch
from instantavatar_amd.training import NeRFLoss, configure_optimizer, field_autograd, training_step

import world as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def torch_field_reference(net, x, enc, col):
    """Plain PyTorch fp32 restatement of NeRFNGPNet.forward (hash grid + MLPs),
    differentiable w.r.t. enc / col / x; no fp16 rounding."""
    L = net.n_levels
    hd = net.hash_desc
    xn = ((x - net.center) / net.scale + 0.5).clamp(0, 1)
    nw1 = net.sig_w1_size
    W1, W2 = enc[:nw1].view(64, 2 * L), enc[nw1:nw1 + 1024].view(16, 64)
    table = enc[nw1 + 1024:].view(-1, 2)
    feats = []
    for l in range(L):
        scale, res = float(hd.scale[l]), int(hd.res[l])
        off, size = int(hd.offset[l]), int(hd.offset[l + 1] - hd.offset[l])
        pos = xn * scale + 0.5
        g = pos.floor()
        w = pos - g
        g = g.long()
        acc = 0
        for idx in range(8):
            c = [g[:, d] + ((idx >> d) & 1) for d in range(3)]
            wt = 1
            for d in range(3):
                wt = wt * (w[:, d] if (idx >> d) & 1 else 1 - w[:, d])
            if res ** 3 <= size:
                index = (c[0] + c[1] * res + c[2] * res * res) % size
            else:
                index = ((c[0] * 1) ^ ((c[1] * 2654435761) & 0xffffffff) ^ ((c[2] * 805459861) & 0xffffffff)) % size
            acc = acc + wt[:, None] * table[off + index]
        feats.append(acc)
    feat = torch.cat(feats, dim=1)
    h1 = torch.relu(feat @ W1.t())
    o16 = h1 @ W2.t()
    sigma = o16[:, 0]
    cin = torch.cat([o16[:, 1:], torch.ones_like(o16[:, :1])], dim=1)
    Wc1, Wc2, Wc3 = col[:1024].view(64, 16), col[1024:5120].view(64, 64), col[5120:].view(16, 64)
    c2 = torch.relu(torch.relu(cin @ Wc1.t()) @ Wc2.t())
    rgb = torch.sigmoid((c2 @ Wc3.t())[:, :3])
    return rgb, sigma


@pytest.fixture(scope="module")
def gw():
    model, body, fp, init = W.build(DEV, 64, 16)
    return model, body, fp, init


def _cos(a, b):
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-30))


def test_field_backward_matches_torch_reference(gw):
    model = gw[0]
    net = model.net_coarse
    g = torch.Generator(device=DEV).manual_seed(3)
    bb = model.deformer.bbox
    x = (torch.rand((20000, 3), device=DEV, generator=g) * (bb[1] - bb[0]) + bb[0]).requires_grad_(True)
    wr = torch.rand((20000, 3), device=DEV, generator=g)
    ws = torch.rand(20000, device=DEV, generator=g) * 0.01
    for p in net.parameters():
        p.grad = None
    rgb, sigma = field_autograd(net, x)
    ((rgb * wr).sum() + (sigma * ws).sum()).backward()
    g_enc, g_col, g_x = net.encoder.params.grad.clone(), net.color_net.params.grad.clone(), x.grad.clone()
    enc = net.encoder.params.detach().half().float().requires_grad_(True)  # the kernel uses the fp16 shadow
    col = net.color_net.params.detach().half().float().requires_grad_(True)
    x2 = x.detach().clone().requires_grad_(True)
    rgb_r, sigma_r = torch_field_reference(net, x2, enc, col)
    # forward agreement (fp16 activations vs fp32 reference)
    assert (rgb - rgb_r).abs().max() < 2e-2 and ((sigma - sigma_r).abs() / (1 + sigma_r.abs())).max() < 2e-2
    ((rgb_r * wr).sum() + (sigma_r * ws).sum()).backward()
    nw = net.sig_w1_size + 1024
    # MLP weight gradients
    assert _cos(g_enc[:nw], enc.grad[:nw]) > 0.999 and _cos(g_col, col.grad) > 0.999
    assert (g_enc[:nw] - enc.grad[:nw]).norm() / enc.grad[:nw].norm() < 3e-2
    # hash-table gradient (scatter-add)
    assert _cos(g_enc[nw:], enc.grad[nw:]) > 0.999
    assert (g_enc[nw:] - enc.grad[nw:]).norm() / enc.grad[nw:].norm() < 3e-2
    assert ((g_enc[nw:] != 0) == (enc.grad[nw:] != 0)).float().mean() > 0.999
    # input gradient
    assert _cos(g_x, x2.grad) > 0.995


def test_fused_mlp_backward_equals_gemm_formulation(gw):
    """ia_field_bwd (one MFMA kernel) against the same backward written as ten fp16 GEMMs with fp32
    accumulation: identical roundings of the intermediate gradients, so only the summation order
    of the fp32 accumulators differs."""
    from instantavatar_amd import training
    model = gw[0]
    g = torch.Generator(device=DEV).manual_seed(5)
    bb = model.deformer.bbox
    nets = [model.net_coarse, W.build(DEV, 64, 8)[0].net_coarse]
    for net in nets:
        for V in (1, 31, 33, 4097, 50001):
            x = (torch.rand((V, 3), device=DEV, generator=g) * (bb[1] - bb[0]) + bb[0]).requires_grad_(True)
            wr = torch.rand((V, 3), device=DEV, generator=g) - 0.3
            ws = (torch.rand(V, device=DEV, generator=g) - 0.5) * 0.01
            res = []
            for fused in (True, False):
                training.FUSED_MLP_BACKWARD = fused
                try:
                    for p in net.parameters():
                        p.grad = None
                    x.grad = None
                    rgb, sigma = field_autograd(net, x)
                    ((rgb * wr).sum() + (sigma * ws).sum()).backward()
                    res.append((net.encoder.params.grad.clone(), net.color_net.params.grad.clone(), x.grad.clone()))
                finally:
                    training.FUSED_MLP_BACKWARD = True
            (ge_f, gc_f, gx_f), (ge_g, gc_g, gx_g) = res
            nw = net.sig_w1_size + 1024
            for a, b, name in ((ge_f[:nw], ge_g[:nw], "sigma-net weights"), (gc_f, gc_g, "colour-net weights"),
                               (ge_f[nw:], ge_g[nw:], "hash table"), (gx_f, gx_g, "input")):
                err = (a - b).norm() / (b.norm() + 1e-30)
                assert err < 2e-3, (net.n_levels, V, name, float(err))
            assert torch.isfinite(ge_f).all() and torch.isfinite(gc_f).all()


def test_mlp_weight_gradients_are_bitwise_reproducible(gw):
    """ia_field_bwd reduces its per-workgroup partial sums in a fixed order (no atomics): two runs on the
    same inputs must give identical MLP weight gradients (the hash-table scatter uses atomics and is
    only compared within rounding)."""
    model = gw[0]
    net = model.net_coarse
    g = torch.Generator(device=DEV).manual_seed(13)
    bb = model.deformer.bbox
    x = (torch.rand((60001, 3), device=DEV, generator=g) * (bb[1] - bb[0]) + bb[0])
    wr = torch.rand((60001, 3), device=DEV, generator=g)
    ws = torch.rand(60001, device=DEV, generator=g) * 0.01
    outs = []
    for _ in range(2):
        for p in net.parameters():
            p.grad = None
        rgb, sigma = field_autograd(net, x)
        ((rgb * wr).sum() + (sigma * ws).sum()).backward()
        outs.append((net.encoder.params.grad.clone(), net.color_net.params.grad.clone()))
    nw = net.sig_w1_size + 1024
    assert torch.equal(outs[0][0][:nw], outs[1][0][:nw]) and torch.equal(outs[0][1], outs[1][1])
    assert (outs[0][0][nw:] - outs[1][0][nw:]).norm() <= 1e-5 * outs[0][0][nw:].norm()


def test_fused_loss_kernel_matches_torch_expression():
    """ia_nerf_loss: the five reported values and the three gradients against the torch-op
    evaluation of loss.py:53-77 (same fp32 functions, different summation order)."""
    g = torch.Generator(device=DEV).manual_seed(9)
    for n, s in ((1, 1), (4096, 256), (1000, 7)):
        pred = {"rgb_coarse": torch.rand((1, n, 3), device=DEV, generator=g).requires_grad_(True),
                "alpha_coarse": torch.rand((1, n), device=DEV, generator=g).requires_grad_(True),
                "weight_coarse": (torch.rand((1, n, s), device=DEV, generator=g) ** 4).requires_grad_(True)}
        tgt = {"rgb": torch.rand((1, n, 3), device=DEV, generator=g), "alpha": (torch.rand((1, n), device=DEV, generator=g) > 0.5).float()}
        outs = []
        for fused in (True, False):
            for t in pred.values():
                t.grad = None
            losses = NeRFLoss(fused=fused)(pred, tgt)
            (losses["loss"] * 3.0).backward()
            outs.append(({k: float(v) for k, v in losses.items()}, {k: t.grad.clone() for k, t in pred.items()}))
        (lf, gf), (lt, gt) = outs
        for k in lt:
            assert abs(lf[k] - lt[k]) <= 2e-5 * max(1.0, abs(lt[k])), (n, s, k, lf[k], lt[k])
        for k in gt:
            assert torch.allclose(gf[k], gt[k], rtol=1e-4, atol=1e-9), (n, s, k)


def test_training_step_learns(gw):
    """One frame, 4096 random rays: loss must be finite, every parameter tensor must receive
    gradient, and a few Adam steps on a fixed batch must reduce the loss."""
    model, body, fp, init = W.build(DEV, 64, 16)
    from instantavatar_amd.pipeline import build_synthetic_model
    tmodel, _, _ = build_synthetic_model(DEV, resolution=64, n_levels=16)
    torch.manual_seed(0)
    poses, tr = W.poses()
    res = 64
    # target image from the synthetic "ground-truth" field
    gt_batch = make_batch(DEV, res, poses[1], tr[1])
    rgb_gt, _, alpha_gt, _ = model.render_image_fast(gt_batch, (res, res))
    # the trainee starts from tcnn-style random init, with the synthetic occupancy as its train grid
    tmodel.net_coarse.reset_parameters()
    tmodel.train()
    sel = torch.randperm(res * res, device=DEV)[:4096]
    batch = make_batch(DEV, res, poses[1], tr[1])
    for k in ("rays_o", "rays_d"):
        batch[k] = batch[k][:, sel]
    for k in ("near", "far"):
        batch[k] = batch[k][:, sel]
    batch["rgb"] = rgb_gt.reshape(1, -1, 3)[:, sel]
    batch["alpha"] = alpha_gt.reshape(1, -1)[:, sel]
    batch["bg_color"] = torch.ones_like(batch["rgb"])
    opt = configure_optimizer(tmodel)
    loss_fn = NeRFLoss(dict(w_rgb=1.0, w_alpha=0.1, w_reg=0.1))
    losses = []
    for it in range(12):
        out = training_step(tmodel, batch, opt, loss_fn)
        assert torch.isfinite(out["loss"])
        losses.append(float(out["mse_loss"]))
        if it == 0:
            for n, p in tmodel.named_parameters():
                assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().sum() > 0, n
    assert losses[-1] < losses[0], losses


def test_fused_train_render_matches_reference_structure(gw):
    """render_train over compact samples (ia_march_train_compact / ia_composite_train_*) against
    the reference-structured route (dense [n,256] tensors, boolean masks, torch cumprod compositing,
    raymarcher_acc.py:140-186): same outputs and same parameter gradients."""
    from instantavatar_amd.models.structures.utils import Rays
    model = gw[0]
    poses, tr = W.poses()
    res = 64
    batch = make_batch(DEV, res, poses[2], tr[2])
    model.deformer.prepare_deformer(batch)
    grid = model.renderer.density_grid_train
    # give the training grid a real occupancy: the test-time grid of this frame, in the train aabb
    model.renderer.density_grid_test.initialize(model.deformer, model.net_coarse, iters=2)
    G = 64
    coords = (grid.coords + 0.5 / G) * (grid.aabb[1] - grid.aabb[0]) + grid.aabb[0]
    with torch.no_grad():
        _, dens = model.deformer(coords.reshape(-1, 3), model.net_coarse, True)
    grid._postprocess(dens.reshape(G, G, G))
    assert grid.density_field.sum() > 500
    sel = torch.randperm(res * res, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))[:2048]
    net = model.net_coarse
    bg = torch.rand((1, 2048, 3), device=DEV)
    wc = torch.rand((2048, 3), device=DEV); wa = torch.rand(2048, device=DEV); ww = torch.rand((2048, 256), device=DEV) * 0.01

    def run(fused):
        rays = Rays(o=batch["rays_o"][:, sel].clone(), d=batch["rays_d"][:, sel].clone(), near=batch["near"][:, sel].clone(),
                    far=batch["far"][:, sel].clone())
        model.deformer.transform_rays_w2s(rays)
        for p in net.parameters():
            p.grad = None
        torch.manual_seed(7)
        if fused:
            out = model.renderer.render_train_fused(rays, model.deformer, net, 0, bg)
        else:  # hide the native pair from the renderer and the deformer: generic masked route
            out = model.renderer.render_train(rays, lambda x, _: model.deformer(x, lambda p, d: net(p, d), False), 0, bg)
        loss = (out["rgb_coarse"].reshape(-1, 3) * wc).sum() + (out["alpha_coarse"].reshape(-1) * wa).sum() + \
            (out["weight_coarse"].reshape(-1, 256) * ww).sum() + out["depth_coarse"].sum() * 0.01
        loss.backward()
        return {k: v.detach().clone() for k, v in out.items()}, net.encoder.params.grad.clone(), net.color_net.params.grad.clone()

    of, ge_f, gc_f = run(True)
    og, ge_g, gc_g = run(False)
    assert (og["alpha_coarse"] > 0.5).float().mean() > 0.02
    for k in ("rgb_coarse", "alpha_coarse", "depth_coarse", "weight_coarse"):
        assert torch.allclose(of[k].reshape(-1), og[k].reshape(-1), atol=2e-5), k
    assert _cos(ge_f, ge_g) > 0.9999 and _cos(gc_f, gc_g) > 0.9999
    assert (ge_f - ge_g).norm() / ge_g.norm() < 1e-3 and (gc_f - gc_g).norm() / gc_g.norm() < 1e-3


def test_implicit_differentiation_of_roots_wrt_pose(gw):
    """Row a7 (deformer_torch.py:50-67): canonical roots are differentiable w.r.t. the SMPL pose
    through x_c <- x_c* - J^-1 (d(x_c*) - sg[d(x_c*)]).  Autograd gradient of a linear functional of
    the roots against central finite differences of the pose (config 4: SMPL refinement)."""
    model = gw[0]
    dfm = model.deformer
    poses, tr = W.poses()
    base = make_batch(DEV, 16, poses[3], tr[3])
    g = torch.Generator(device=DEV).manual_seed(5)

    def roots(body_pose):
        b = dict(base)
        b["body_pose"] = body_pose
        dfm.prepare_deformer(b)       # differentiable torch route when the pose carries grad
        return dfm.deform(pts, eval_mode=False)

    bp0 = base["body_pose"].clone()
    dfm.prepare_deformer(base)
    vd = dfm.deformer.voxel_d[0].reshape(3, -1)
    sel = torch.randint(0, vd.shape[1], (3000,), device=DEV, generator=g)
    pts = (vd[:, sel].T + 0.01 * torch.randn((3000, 3), device=DEV, generator=g)).contiguous()
    r = torch.randn((3000, 13, 3), device=DEV, generator=g)
    bp = bp0.clone().requires_grad_(True)
    xc, valid = roots(bp)
    assert valid.float().mean() > 0.05
    loss = (xc * r)[valid].sum()
    (grad,) = torch.autograd.grad(loss, bp)
    assert torch.isfinite(grad).all() and grad.abs().max() > 0
    # finite differences on the 6 pose entries with the largest gradient
    idx = grad.abs().reshape(-1).topk(6).indices
    eps = 2e-3
    fd = []
    for k in idx.tolist():
        vals = []
        for sgn in (+1, -1):
            p = bp0.clone(); p.reshape(-1)[k] += sgn * eps
            with torch.no_grad():
                x2, v2 = roots(p)
            vals.append((x2, v2))
        both = valid & vals[0][1] & vals[1][1]
        fd.append((((vals[0][0] - vals[1][0]) * r)[both].sum() / (2 * eps)).item())
        # restrict the analytic value to the same candidates
    fd = torch.tensor(fd)
    an = grad.reshape(-1)[idx].cpu()
    cos = float((fd * an).sum() / (fd.norm() * an.norm()))
    assert cos > 0.98, (cos, fd, an)


def test_implicit_diff_kernel_equals_torch_formulation(gw):
    """ia_snarf_implicit_bwd against the reference's formulation of a7 (grid_sample with border
    padding + einsum + batched mat-vec under autograd): same gradient w.r.t. the bone transforms."""
    from instantavatar_amd.deformers.fast_snarf import forward_deformer as fd
    model = gw[0]
    dfm = model.deformer
    poses, tr = W.poses()
    base = make_batch(DEV, 16, poses[2], tr[2])
    dfm.prepare_deformer(base)
    g = torch.Generator(device=DEV).manual_seed(6)
    vd = dfm.deformer.voxel_d[0].reshape(3, -1)
    sel = torch.randint(0, vd.shape[1], (5000,), device=DEV, generator=g)
    pts = (vd[:, sel].T + 0.01 * torch.randn((5000, 3), device=DEV, generator=g)).contiguous()
    pts[:50] += 5.0   # far outside: no roots, and border-clamped weight samples must not matter
    r = torch.randn((5000, 13, 3), device=DEV, generator=g)
    grads = []
    for fused in (True, False):
        fd.FUSED_IMPLICIT_DIFF = fused
        try:
            tfs = dfm.tfs.detach().clone().requires_grad_(True)
            xc, others = dfm.deformer.forward(pts[None], None, tfs, eval_mode=False)
            valid = others["valid_ids"]
            assert valid.float().mean() > 0.05
            ((xc * r[None])[valid]).sum().backward()
            grads.append(tfs.grad.clone())
        finally:
            fd.FUSED_IMPLICIT_DIFF = True
    a, b = grads
    assert torch.isfinite(a).all() and b.abs().max() > 0
    assert (a[..., 3, :] == 0).all()
    assert (a - b).norm() / b.norm() < 1e-4, float((a - b).norm() / b.norm())


def test_implicit_diff_kernel_matches_oracle(oracle, gw):
    """ia_snarf_implicit_bwd against the CPU restatement of a7 (oracle.implicit_diff_grad) on the
    same roots, Broyden J_inv, validity mask and incoming gradient."""
    import ctypes as C
    from instantavatar_amd import _lib
    model = gw[0]
    dfm = model.deformer
    poses, tr = W.poses()
    dfm.prepare_deformer(make_batch(DEV, 16, poses[1], tr[1]))
    fd = dfm.deformer
    g = torch.Generator(device=DEV).manual_seed(8)
    vd = fd.voxel_d[0].reshape(3, -1)
    sel = torch.randint(0, vd.shape[1], (4000,), device=DEV, generator=g)
    pts = (vd[:, sel].T + 0.01 * torch.randn((4000, 3), device=DEV, generator=g)).contiguous()
    xc, others = fd.search(pts[None], None, dfm.tfs, eval_mode=True, want_J_inv=True)
    valid, J_inv = others["valid_ids"], others["J_inv"]
    r = torch.randn(xc.shape, device=DEV, generator=g)
    L = _lib.lib()
    n = valid.numel()
    d_tfs = torch.zeros((24, 4, 4), device=DEV)
    ws = torch.empty(int(L.ia_snarf_implicit_bwd_workspace_bytes(n)), dtype=torch.uint8, device=DEV)
    x_, J_, m_, r_ = xc.reshape(-1, 3).contiguous(), J_inv.reshape(-1, 9).contiguous(), valid.reshape(-1).to(torch.uint8), r.reshape(-1, 3).contiguous()
    _lib.check(L.ia_snarf_implicit_bwd(_lib.ptr(x_), _lib.ptr(J_), _lib.ptr(m_), _lib.ptr(r_), n, _lib.ptr(fd.lbs_voxel_final),
                                       C.byref(fd.grid_desc()), _lib.ptr(d_tfs), _lib.ptr(ws), ws.numel(), _lib.stream()),
               "ia_snarf_implicit_bwd")
    init = dict(lbs_voxel=fd.lbs_voxel_final[0].cpu().numpy(), offset_kernel=fd.offset_kernel.reshape(3).cpu().numpy(),
                scale_kernel=fd.scale_kernel.reshape(3).cpu().numpy())
    ref = oracle.implicit_diff_grad(init, x_.cpu().numpy(), J_.cpu().numpy(), m_.cpu().numpy(), r_.cpu().numpy())
    got = d_tfs.cpu().numpy()
    assert valid.float().mean() > 0.05 and np.abs(ref).max() > 1
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-4


def test_train_render_matches_oracle(oracle, gw):
    """Row a15 end to end: render_train over compact samples (march + jitter, candidate search, field,
    wave-per-ray compositing) against oracle.render_train on the same rays, occupancy, jitter and
    background: rgb / alpha within 1e-3, weights within 1e-3."""
    from instantavatar_amd.models.structures.utils import Rays
    model, body, fp, init = gw
    poses, tr = W.poses()
    res, i, n = 64, 2, 1024
    batch = make_batch(DEV, res, poses[i], tr[i])
    model.deformer.prepare_deformer(batch)
    grid = model.renderer.density_grid_train
    model.renderer.density_grid_test.initialize(model.deformer, model.net_coarse, iters=2)
    G = 64
    coords = (grid.coords + 0.5 / G) * (grid.aabb[1] - grid.aabb[0]) + grid.aabb[0]
    with torch.no_grad():
        _, dens = model.deformer(coords.reshape(-1, 3), model.net_coarse, True)
    grid._postprocess(dens.reshape(G, G, G))
    sel = torch.randperm(res * res, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))[:n]
    rays = Rays(o=batch["rays_o"][:, sel].clone(), d=batch["rays_d"][:, sel].clone(), near=batch["near"][:, sel].clone(),
                far=batch["far"][:, sel].clone())
    model.deformer.transform_rays_w2s(rays)
    bg = torch.rand((1, n, 3), device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
    ow = W.oracle_world(oracle, body, fp, init, poses[i], tr[i])
    c = lambda t: t.detach().reshape(-1, t.shape[-1]).cpu().numpy() if t.dim() > 2 else t.detach().reshape(-1).cpu().numpy()
    for noise_scale in (0, 1):
        torch.manual_seed(11)
        jitter = torch.rand((n, 256), device=DEV)          # the first draw of the render (raymarcher_acc.py:156)
        noise = torch.randn((n, 256), device=DEV)          # the second, when noise > 0 (:167): one value per (ray, slot)
        torch.manual_seed(11)
        out = model.renderer.render_train_fused(rays, model.deformer, model.net_coarse, noise_scale, bg)
        ref = oracle.render_train(c(rays.o), c(rays.d), c(rays.near), c(rays.far), grid.density_field.cpu().numpy(),
                                  grid.aabb.cpu().numpy(), lambda p: oracle.deform_query(p, ow, eval_mode=False),
                                  jitter.cpu().numpy(), bg=c(bg), noise=(noise * noise_scale).cpu().numpy() if noise_scale else None)
        rgb = out["rgb_coarse"].detach().reshape(-1, 3).cpu().numpy()
        alpha = out["alpha_coarse"].detach().reshape(-1).cpu().numpy()
        w = out["weight_coarse"].detach().reshape(n, 256).cpu().numpy()
        assert (ref["alpha"] > 0.5).mean() > 0.02 and ref["n_field"] > 1000
        err_rgb, err_a, err_w = np.abs(rgb - ref["rgb"]).max(1), np.abs(alpha - ref["alpha"]), np.abs(w - ref["weights"]).max(1)
        for e, nm in ((err_rgb, "rgb"), (err_a, "alpha"), (err_w, "weights")):
            W.rays_within(e, "training render (noise %g) %s" % (noise_scale, nm), frac=5e-4)
        assert np.median(err_rgb) < 1e-4


def _train_setup(seed_model, res=64, n_rays=4096):
    from instantavatar_amd.pipeline import build_synthetic_model
    model, body, fp, init = W.build(DEV, 64, 16)
    poses, tr = W.poses()
    torch.manual_seed(seed_model)
    tmodel, _, _ = build_synthetic_model(DEV, resolution=64, n_levels=16)
    tmodel.net_coarse.reset_parameters()
    tmodel.train()
    batches = []
    gsel = torch.Generator(device=DEV).manual_seed(5)
    for f in (1, 2, 3):
        b = make_batch(DEV, res, poses[f], tr[f])
        rgb_gt, _, alpha_gt, _ = model.render_image_fast(b, (res, res))
        sel = torch.randperm(res * res, device=DEV, generator=gsel)[:n_rays]
        for k in ("rays_o", "rays_d", "near", "far"):
            b[k] = b[k][:, sel].contiguous()
        b["rgb"] = rgb_gt.reshape(1, -1, 3)[:, sel].contiguous()
        b["alpha"] = alpha_gt.reshape(1, -1)[:, sel].contiguous()
        b["bg_color"] = torch.ones_like(b["rgb"])
        batches.append(b)
    return tmodel, batches


def test_graphed_train_step_matches_eager_steps(gw):
    """GraphedTrainStep (the whole step replayed from a captured HIP graph) against `training_step` launched eagerly:
    same initial weights, same batches, same RNG seed -> the same loss curve (the jitter / noise draws of a replay
    continue the philox sequence exactly like eager calls; what differs is the order of the scatter atomics)."""
    from instantavatar_amd.training import GraphedTrainStep
    n_steps = 14
    curves = []
    for graphed in (False, False, True):
        tmodel, batches = _train_setup(seed_model=3)
        opt = configure_optimizer(tmodel)
        loss_fn = NeRFLoss(dict(w_rgb=1.0, w_alpha=0.1, w_reg=0.1))
        stepper = GraphedTrainStep(tmodel, opt, loss_fn, enabled=graphed)
        torch.manual_seed(11)
        ls = []
        for it in range(n_steps):
            out = stepper(batches[it % len(batches)])
            ls.append(float(out["mse_loss"]))
            assert float(out["skipped_non_finite"]) == 0.0
        if graphed:
            assert stepper.capture_error is None, stepper.capture_error
            assert stepper.replays == n_steps - 1 and stepper.eager_steps == 1, (stepper.replays, stepper.eager_steps)
            assert tmodel.global_step == n_steps
            # (the fused optimiser step of a graphed trainer zero-fills every gradient it consumed -- `FusedAdam.fused_zero_grad`,
            # set by GraphedTrainStep -- so the next step accumulates into clean buffers without a 52 MB fill launch)
            assert opt.fused_zero_grad and opt.grads_zeroed
            for n, p in tmodel.named_parameters():
                assert p.grad is not None and float(p.grad.abs().sum()) == 0.0, n
        curves.append(ls)
    e0, e, g = np.array(curves[0]), np.array(curves[1]), np.array(curves[2])
    # two eager runs: identical draws for identical (ray, slot) cells; what is left is the order of float atomics
    assert np.allclose(e0, e, rtol=5e-3, atol=1e-6), (e0, e)
    assert e[-1] < e[0] and g[-1] < g[0], (e, g)
    assert np.allclose(e, g, rtol=2e-2, atol=1e-6), (e, g)


def test_graphed_train_step_sees_lr_changes_and_skips_non_finite(gw):
    """A replay must honour what the host changes between steps: the learning rate (device tensor, written by
    an lr_scheduler) and a non-finite loss (found_inf flag of the fused Adam: parameters stay untouched)."""
    from instantavatar_amd.training import GraphedTrainStep, configure_scheduler
    tmodel, batches = _train_setup(seed_model=4)
    opt = configure_optimizer(tmodel)
    sched = configure_scheduler(opt, max_epochs=2)
    loss_fn = NeRFLoss(dict(w_rgb=1.0, w_alpha=0.1, w_reg=0.1))
    stepper = GraphedTrainStep(tmodel, opt, loss_fn)
    for it in range(4):
        stepper(batches[it % 3])
    assert stepper.replays == 3 and stepper.capture_error is None, stepper.capture_error
    p = tmodel.net_coarse.encoder.params
    before = p.detach().clone()
    stepper(batches[1])
    torch.cuda.synchronize()
    assert not torch.equal(before, p.detach())
    # epoch 2 of 2: the LambdaLR factor is 0 -> a replayed step must leave the parameters where they are
    sched.step()
    sched.step()
    assert all(float(g["lr"]) == 0.0 for g in opt.param_groups)
    before = p.detach().clone()
    stepper(batches[2])
    torch.cuda.synchronize()
    assert torch.equal(before, p.detach())
    for g in opt.param_groups:
        g["lr"].fill_(1e-2)
    bad = dict(batches[0])
    bad["rgb"] = batches[0]["rgb"].clone()
    bad["rgb"][0, 0, 0] = float("nan")
    out = stepper(bad)
    torch.cuda.synchronize()
    assert float(out["skipped_non_finite"]) == 1.0
    assert torch.equal(before, p.detach())
    out = stepper(batches[0])
    torch.cuda.synchronize()
    assert float(out["skipped_non_finite"]) == 0.0 and not torch.equal(before, p.detach())
    assert stepper.eager_steps == 1


def test_hashgrid_backward_is_independent_of_call_granularity(gw):
    """The scatter forms its sums in a launch-dependent order (in-wave run reduction, quad-cooperative atomics, grid-stride
    rounds): one call over 150 000 clustered samples (four small boxes, like a patch batch: heavy same-cell traffic) must
    equal the sum of five 30 000-sample calls up to fp32 summation order -- with and without the input gradient -- and so
    must the level-range calls of the bucketed all-reduce path."""
    import ctypes as C
    from instantavatar_amd import _lib
    for net in (gw[0].net_coarse, W.build(DEV, 64, 8)[0].net_coarse):
        L = _lib.lib()
        g = torch.Generator(device=DEV).manual_seed(9)
        V, nf = 150000, 2 * net.n_levels
        bb = gw[0].deformer.bbox
        centre = torch.rand((4, 3), device=DEV, generator=g) * 0.6 + 0.2
        box = centre[torch.randint(0, 4, (V,), device=DEV, generator=g)] + (torch.rand((V, 3), device=DEV, generator=g) - 0.5) * 0.08
        x = (box * (bb[1] - bb[0]) + bb[0]).contiguous()
        dfeat = torch.randn((V, nf), device=DEV, generator=g) * 1e-2
        dfeat[torch.rand(V, device=DEV, generator=g) < 0.3] = 0      # candidates without gradient
        n_tab = 2 * net.n_entries
        fd = net.field_desc()

        def run(chunks, want_dx):
            dt = torch.zeros(n_tab, device=DEV)
            dx = torch.zeros((V, 3), device=DEV) if want_dx else None
            for a in range(0, V, chunks):
                b = min(V, a + chunks)
                _lib.check(L.ia_hashgrid_bwd(_lib.ptr(x[a:b].contiguous()), b - a, None, C.byref(fd), _lib.ptr(dfeat[a:b].contiguous()),
                                             dt.data_ptr(), _lib.ptr(dx[a:b]) if want_dx else None, _lib.stream()), "ia_hashgrid_bwd")
            return dt, dx

        ref, dx_ref = run(30000, True)
        for want_dx in (False, True):
            got, dx = run(V, want_dx)
            assert ((got != 0) == (ref != 0)).all()
            err = (got - ref).abs().max() / ref.abs().max()
            assert err < 1e-5, (net.n_levels, want_dx, float(err))
            if want_dx:
                assert torch.equal(dx, dx_ref)   # the input gradient is lane-local: bit identical
        # level ranges (the bucketed all-reduce path) add up to the same table
        parts = torch.zeros(n_tab, device=DEV)
        for l0, l1 in ((net.n_levels // 2, net.n_levels), (0, net.n_levels // 2)):
            _lib.check(L.ia_hashgrid_bwd_levels(_lib.ptr(x), V, None, C.byref(fd), _lib.ptr(dfeat), parts.data_ptr(), l0, l1, _lib.stream()))
        assert (parts - ref).abs().max() / ref.abs().max() < 1e-5


def test_render_candidate_overflow_skips_the_optimizer_step_on_the_device(gw):
    """a training render whose candidates exceed the capacity drops some of them -- its gradients are
    wrong.  The step is then skipped ON THE DEVICE (found_inf of the fused Adam, no host read), the deferred count check
    grows the capacity, and the following steps are whole."""
    tmodel, batches = _train_setup(seed_model=6, n_rays=1024)
    opt = configure_optimizer(tmodel)
    loss_fn = NeRFLoss(dict(w_rgb=1.0, w_alpha=0.1, w_reg=0.1))
    r = tmodel.renderer
    training_step(tmodel, batches[0], opt, loss_fn)              # step 0: builds the occupancy grid, plenty of capacity
    r.train_cand_capacity, r.train_overflow = 512, 0
    p = tmodel.net_coarse.encoder.params
    skipped = []
    for it in range(4):
        before = p.detach().clone()
        out = training_step(tmodel, batches[(it + 1) % 3], opt, loss_fn)
        torch.cuda.synchronize()
        s = float(out["skipped_overflow"])
        skipped.append(s)
        assert float(out["skipped_non_finite"]) == s
        assert torch.equal(before, p.detach()) == (s == 1.0), (it, s)
    assert skipped[0] == 1.0 and skipped[-1] == 0.0, skipped     # the capacity grew: the last step went through
    assert r.train_overflow >= 1 and r.train_cand_capacity > 512


def test_non_finite_upstream_gradient_skips_the_step(gw):
    """one NaN in d_sigma must end in `skipped_non_finite` through the gradient scale (S = NaN), not through luck."""
    model = gw[0]
    net = model.net_coarse
    bb = model.deformer.bbox
    x = torch.rand((4096, 3), device=DEV, generator=torch.Generator(device=DEV).manual_seed(9)) * (bb[1] - bb[0]) + bb[0]
    for p in net.parameters():
        p.grad = None
    rgb, sigma = field_autograd(net, x)
    w = torch.ones_like(sigma)
    w[17] = float("nan")
    ((rgb.sum()) + (sigma * w).sum()).backward()
    from instantavatar_amd.training import _non_finite_flag
    assert float(_non_finite_flag(list(net.parameters()))) == 1.0
    assert not torch.isfinite(net.encoder.params.grad).all()
    for p in net.parameters():
        p.grad = None


def test_direct_loss_gradient_seeding_equals_the_autograd_route(gw):
    """training_step seeds autograd with the loss kernel's own gradients (`NeRFLoss.value_and_grads`, NaN poisoning inside
    `ia_nerf_loss`) when the loss is exactly the kernel's five terms; the route through the autograd scalar (loss -> where -> mul ->
    backward -> foreach_mul) is kept for losses with further terms.  Same state, same draws -> same losses and gradients (up to the
    order of the scatter atomics), with and without a poisoned (overflowed) render."""
    res = {}
    for direct in (True, False):
        tmodel, batches = _train_setup(seed_model=8, n_rays=1024)
        opt = configure_optimizer(tmodel)
        loss_fn = NeRFLoss(dict(w_rgb=1.0, w_alpha=0.1, w_reg=0.1))
        if not direct:
            loss_fn.direct_backward_ok = lambda predicts: False
        torch.manual_seed(3)
        training_step(tmodel, batches[0], opt, loss_fn)      # step 0 (occupancy update + regulariser: autograd route on both sides)
        torch.manual_seed(4)
        out = training_step(tmodel, batches[1], opt, loss_fn)
        g = tmodel.net_coarse.encoder.params.grad.detach().clone()
        res[direct] = ({k: float(v) for k, v in out.items() if torch.is_tensor(v) and v.numel() == 1}, g, tmodel.net_coarse.encoder.params.detach().clone())
    (l1, g1, p1), (l0, g0, p0) = res[True], res[False]
    for k in ("loss", "mse_loss", "loss_alpha_coarse", "reg_alpha", "reg_density"):
        # (step 1 starts from the parameters step 0 left: those differ by the order of the scatter atomics)
        assert abs(l1[k] - l0[k]) <= 5e-3 * abs(l0[k]) + 1e-9, (k, l1[k], l0[k])
    assert float(g0.abs().max()) > 0
    assert float((g1 - g0).norm() / g0.norm()) < 1e-3 and float((p1 - p0).abs().max()) < 2e-3      # (Adam: sign-like first steps)
