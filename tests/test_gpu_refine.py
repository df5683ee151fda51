"""GPU tests (-m gpu) of BASELINE config 4 (confs/SNARF_NGP_refine.yaml): SMPL parameters optimised together with the field
through the SNARF deformer.  The HIP product path (`training.training_step` with `SMPLParamEmbedding` + `SNARFDeformer`,
`tfs.requires_grad`, fused route: `ia_snarf_search_compact_jinv` -> field -> `ia_hashgrid_bwd` (dx) ->
`ia_snarf_implicit_bwd_compact`) against tests/golden/refine_golden.npz = the REFERENCE's `DNeRFModel.training_step`
(DNeRF.py:112-161) executing on the CPU for three steps (tests/golden/make_refine_golden.py)."""
import os

import numpy as np
import pytest
import torch

from instantavatar_amd import synthetic as syn
from instantavatar_amd.models.structures.body_model_param import SMPLParamEmbedding
from instantavatar_amd.training import NGPLoss, configure_optimizer, training_step

import world as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
_L55 = "_l55" if os.environ.get("IA_TCNN_LEVEL3_RES", "54") == "55" else ""
# Every test runs twice: on the zero-blendshape body of SURVEY 8d (refine_golden.npz) and on the blend-shape body with
# betas = synthetic.BLEND_BETAS (refine_golden_blend.npz = the same recipe with IA_GOLDEN_BLEND=1: the reference's lbs.py under
# autograd with non-zero shapedirs / posedirs and a dense J_regressor -- the rest joints `ia_smpl_tfs[_bwd]` chains from then
# depend on betas) -- VERDICT r05 missing 3.
G = None
BLEND = False


@pytest.fixture(autouse=True, params=[False, True], ids=["zero-blendshapes", "blendshapes"])
def golden(request):
    global G, BLEND
    BLEND = request.param
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "refine_golden%s%s.npz" % ("_blend" if BLEND else "", _L55)))
    return G


def _betas():
    return torch.as_tensor(syn.BLEND_BETAS, device=DEV)[None] if BLEND else torch.zeros(1, 10, device=DEV)


def _cos(a, b):
    a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
    return float((a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


def _rel(a, b):
    a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))


def _setup():
    from instantavatar_amd.pipeline import build_synthetic_model
    model, body, fp = build_synthetic_model(DEV, resolution=32, n_levels=16, blendshapes=BLEND, betas=syn.BLEND_BETAS if BLEND else None)
    model.SMPL_param = SMPLParamEmbedding(**{k: torch.as_tensor(G["table_" + k]) for k in ("betas", "global_orient", "transl", "body_pose")}).to(DEV)
    opt = configure_optimizer(model, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, smpl_lr=1e-5)       # SNARF_NGP_refine.yaml
    loss_fn = NGPLoss(dict(w_rgb=1.0, w_alpha=0.1, w_reg=0.1))
    model.train()
    prep = model.deformer.prepare_deformer

    def prepare_and_hook(params):   # tfs is not a leaf (and training_step drops the graph at its end): record its gradient
        prep(params)
        model.last_tfs = model.deformer.tfs.detach().clone()
        if model.deformer.tfs.requires_grad:
            model.deformer.tfs.register_hook(lambda g: setattr(model, "d_tfs", g.detach().clone()))
    model.deformer.prepare_deformer = prepare_and_hook
    return model, opt, loss_fn


def _batch(k):
    res, n_rays = int(G["res"]), int(G["n_rays"])
    f = k % int(G["n_frames"])
    _, tr = syn.procedural_pose_track(8)
    ro, rd = syn.make_camera_rays(res)
    s = G["sel"][k]
    dist = float(np.sqrt((tr[f] ** 2).sum()))
    t = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=DEV)
    return {"rays_o": t(ro[s])[None], "rays_d": t(rd[s])[None], "near": torch.full((1, n_rays), dist - 1, device=DEV),
            "far": torch.full((1, n_rays), dist + 1, device=DEV), "betas": _betas(),
            "rgb": t(G["tgt_rgb_%d" % k])[None], "alpha": t(G["tgt_alpha_%d" % k])[None],
            "bg_color": torch.ones(1, n_rays, 3, device=DEV), "idx": torch.tensor([f])}


def _draws(k, model):
    """the numbers the golden's torch.rand_like calls returned, in call order (ref_cpu_harness.SeededDraws)"""
    rs = np.random.RandomState(int(G["seeds"][0]) + k)
    d = {}
    if model.global_step % 20 == 0:
        d["grid_jitter"] = torch.as_tensor(rs.rand(64, 64, 64, 3).astype(np.float32), device=DEV)          # density_grid.py:47
    d["ray_jitter"] = torch.as_tensor(rs.rand(int(G["n_rays"]), 256).astype(np.float32), device=DEV)       # raymarcher_acc.py:156
    return d


def _grads(model):
    n1 = model.net_coarse.sig_w1_size + 1024
    ge = model.net_coarse.encoder.params.grad.detach().cpu().numpy()
    return dict(d_tfs=model.d_tfs[0].cpu().numpy(), mlp_sigma=ge[:n1], table=ge[n1:],
                mlp_color=model.net_coarse.color_net.params.grad.detach().cpu().numpy(),
                **{n: getattr(model.SMPL_param, n).weight.grad.detach().cpu().numpy() for n in ("global_orient", "transl", "body_pose")})


def test_refine_training_steps_match_reference_training_step_golden():
    model, opt, loss_fn = _setup()
    n_steps = int(G["n_steps"])
    for k in range(n_steps):
        f = k % int(G["n_frames"])
        losses = training_step(model, _batch(k), opt, loss_fn, is_refine=True, draws=_draws(k, model))
        assert "reg" not in losses                                                     # DNeRF.py:137: no regulariser when refining
        assert float(losses["skipped_non_finite"]) == 0
        assert model.renderer.train_overflow == 0
        got = np.array([float(losses[n].detach()) for n in ("loss", "mse_loss", "loss_alpha_coarse", "reg_alpha", "reg_density")])
        ref = G["loss_%d" % k]
        # step 0 starts from identical states; later steps carry the (sign-like, lr 1e-2) Adam updates of both sides
        # (measured on MI355X, step 0: loss 6.01394e-3 against 6.01398e-3)
        # (steps 1, 2: 6.542704e-3 / 8.418754e-3 against 6.542713e-3 / 8.418824e-3 -- the two optimisers stay together)
        tol = 2e-4 if k == 0 else 1e-3
        print("step", k, "losses", got, "reference", ref)
        assert np.all(np.abs(got - ref) <= tol * np.abs(ref) + 1e-6), (k, got, ref)
        g = _grads(model)
        assert np.abs(model.last_tfs[0].cpu().numpy() - G["tfs_%d" % k]).max() < (2e-5 if k == 0 else 2e-4)
        # measured with the joint chain as kernels (tfs within 2e-5 of the reference's torch ops: a few (point, init) pairs at a
        # validity boundary flip, and with them their gradient contributions): step 0 d tfs / body_pose rel 3e-3, MLP weights
        # cos 0.9998 rel 2e-2.  With the chain as torch ops (same operation order as the golden) the same step gives rel 1e-3 /
        # 2e-3: test_refine_step0_with_torch_joint_chain_is_tight pins that.
        # (blend-shape body, measured: step 0 cos >= 0.99978 rel <= 2.2e-2; step 1 -- which carries both sides' first Adam update --
        # MLP colour weights cos 0.99869 rel 5.2e-2, everything else cos >= 0.9996 rel <= 3e-2; with the joint chain as torch ops
        # step 0 agrees to rel <= 1.8e-3 on this body too: test_refine_step0_with_torch_joint_chain_is_tight[blendshapes])
        c_min, r_max = (0.9995, 3e-2) if k == 0 else ((0.998, 7e-2) if BLEND else (0.999, 5e-2))
        report = {}
        for name, ref_g in (("d_tfs", G["d_tfs_%d" % k]), ("body_pose", G["g_body_pose_%d" % k]), ("mlp_sigma", G["g_mlp_sigma_%d" % k]),
                            ("mlp_color", G["g_mlp_color_%d" % k])):
            report[name] = (_cos(g[name], ref_g), _rel(g[name], ref_g))
        print("step", k, "loss", got[0], ref[0], {n: ("cos %.5f rel %.4f" % v) for n, v in report.items()})
        for name, (c, r) in report.items():
            assert c > c_min and r < r_max, (k, name, c, r)
        # only the row of the frame that was used has a gradient; the bone transforms in the SMPL-root frame do not depend on
        # the global orientation / translation (tfs = inv(A_0) A inv(A_rest)): their gradients are rounding noise on both sides
        bp = g["body_pose"]
        assert np.abs(bp[f]).sum() > 0 and np.abs(np.delete(bp, f, axis=0)).sum() == 0
        assert np.abs(g["global_orient"]).max() < 1e-5 and np.abs(g["transl"]).max() < 1e-5
        assert np.abs(G["g_global_orient_%d" % k]).max() < 1e-5 and np.abs(G["g_transl_%d" % k]).max() < 1e-5
        # hash-table gradient: norm, support and a sample of entries
        tn = float(np.sqrt((g["table"].astype(np.float64) ** 2).sum()))
        assert abs(tn - float(G["g_table_norm_%d" % k])) < (0.03 if k == 0 else 0.1) * float(G["g_table_norm_%d" % k])
        if k == 0:
            # support: gradients are rounded to half under the per-call scale (tcnn's fp16 backward): the smallest of the
            # reference's fp32 contributions flush to zero, nothing appears that the reference does not have
            nnz, nnz_ref = int((g["table"] != 0).sum()), int(G["g_table_nnz_%d" % k])
            print("table gradient non-zeros", nnz, "reference", nnz_ref)
            assert 0.85 * nnz_ref < nnz <= nnz_ref
            assert _cos(g["table"][G["g_table_idx_%d" % k]], G["g_table_val_%d" % k]) > 0.99
    # after three steps the SMPL tables moved by at most 3 x lr = 3e-5, in the reference's direction
    for name in ("global_orient", "transl", "body_pose"):
        tab = getattr(model.SMPL_param, name).weight.detach().cpu().numpy()
        # (Adam normalises: even the rounding-noise gradients of global_orient / transl move their rows by lr per step,
        # on both sides, in directions that need not agree -> at most 2 x 3 x lr apart)
        assert np.abs(tab - G["final_" + name]).max() < 6.5e-5, name
    ref_move = G["final_body_pose"] - G["table_body_pose"]
    move = model.SMPL_param.body_pose.weight.detach().cpu().numpy() - G["table_body_pose"]
    big = np.abs(ref_move) > 5e-6
    agree = (np.sign(move[big]) == np.sign(ref_move[big])).mean()
    print("body_pose entries moved by the reference:", int(big.sum()), "same direction:", agree)
    assert big.sum() > 50 and agree > 0.9


def test_refine_fused_route_equals_dense_torch_route():
    """The fused refine route (compact candidates + `ia_snarf_implicit_bwd_compact`) against the dense route that follows the
    reference's structure line by line (ForwardDeformer.forward: dense [1,P,13,*] tensors, boolean-mask gathers,
    `_ImplicitDiffFn`): same step, same draws -> same loss, same gradients up to summation order."""
    res = {}
    for dense in (False, True):
        model, opt, loss_fn = _setup()
        model.deformer.force_dense_train = dense
        losses = training_step(model, _batch(0), opt, loss_fn, is_refine=True, draws=_draws(0, model))
        res[dense] = (float(losses["loss"].detach()), _grads(model))
    (l0, g0), (l1, g1) = res[False], res[True]
    assert abs(l0 - l1) < 1e-5 * abs(l1), (l0, l1)
    for name in ("d_tfs", "body_pose", "mlp_sigma", "mlp_color"):
        assert _cos(g0[name], g1[name]) > 0.9999 and _rel(g0[name], g1[name]) < 5e-3, (name, _cos(g0[name], g1[name]), _rel(g0[name], g1[name]))


@pytest.mark.parametrize("with_grad", [True, False])
def test_version2_fused_route_equals_dense_torch_route(with_grad):
    """ForwardDeformer `version: 2` (deformer_torch.py:68-75, confs/deformer/fast_snarf_debug.yaml): the fused route (compact
    candidates, `ia_expand_candidate_points` + `ia_snarf_inverse_skinning[_bwd]`) against the dense route whose version-2 branch
    is the reference's torch expression (FUSED_IMPLICIT_DIFF off: grid_sample + einsum + batched vector-matrix under autograd).
    With the SMPL tables under optimisation the gradients must agree; without, the inverse-skinned candidates still replace the
    roots (the reference's training branch does so whether or not tfs requires a gradient) -- and differ from version 1's."""
    from instantavatar_amd.deformers.fast_snarf import forward_deformer as fdm
    res = {}
    for route in ("fused", "dense-kernel", "dense-torch", "v1"):
        model, opt, loss_fn = _setup()
        model.deformer.deformer.version = 1 if route == "v1" else 2
        model.deformer.force_dense_train = route.startswith("dense")
        old = fdm.FUSED_IMPLICIT_DIFF
        fdm.FUSED_IMPLICIT_DIFF = route != "dense-torch"
        try:
            if not with_grad:
                for p in model.SMPL_param.parameters():
                    p.requires_grad_(False)
            losses = training_step(model, _batch(0), opt, loss_fn, is_refine=True, draws=_draws(0, model))
        finally:
            fdm.FUSED_IMPLICIT_DIFF = old
        n1 = model.net_coarse.sig_w1_size + 1024
        mlp = dict(mlp_sigma=model.net_coarse.encoder.params.grad.detach().cpu().numpy()[:n1],
                   mlp_color=model.net_coarse.color_net.params.grad.detach().cpu().numpy())
        res[route] = (float(losses["loss"].detach()), {k: v for k, v in _grads(model).items() if k != "table"} if with_grad else mlp)
    l_ref, g_ref = res["dense-torch"]
    for route in ("fused", "dense-kernel"):
        l, g = res[route]
        assert abs(l - l_ref) < 1e-5 * abs(l_ref), (route, l, l_ref)
        for name in g_ref:
            assert _cos(g[name], g_ref[name]) > 0.9999 and _rel(g[name], g_ref[name]) < 5e-3, (route, name, _cos(g[name], g_ref[name]), _rel(g[name], g_ref[name]))
    assert abs(res["v1"][0] - l_ref) > 1e-4 * abs(l_ref), "version 2 must not render from the roots themselves"


def test_inverse_skinning_kernels_match_oracle(oracle):
    """`ia_snarf_inverse_skinning` / `_bwd` (dense layout with a validity mask AND compact layout with cand_pt + device count)
    against oracle.inverse_skinning -- itself pinned to the reference's forward and autograd
    (test_inverse_skinning_version2_matches_reference_autograd_golden)."""
    import ctypes as C
    from instantavatar_amd import _lib
    from instantavatar_amd.pipeline import make_batch
    model, opt, loss_fn = _setup()
    dfm = model.deformer
    poses, tr = syn.procedural_pose_track(8)
    dfm.prepare_deformer(make_batch(DEV, 16, poses[1], tr[1]))
    fd = dfm.deformer
    g = torch.Generator(device=DEV).manual_seed(21)
    vd = fd.voxel_d[0].reshape(3, -1)
    sel = torch.randint(0, vd.shape[1], (3000,), device=DEV, generator=g)
    pts = (vd[:, sel].T + 0.01 * torch.randn((3000, 3), device=DEV, generator=g)).contiguous()
    tfs = dfm.tfs.detach()
    xc, others = fd.search(pts[None], None, tfs, eval_mode=True, want_J_inv=False)
    valid = others["valid_ids"]
    r = torch.randn(xc.shape, device=DEV, generator=g)
    init = dict(lbs_voxel=fd.lbs_voxel_final[0].cpu().numpy(), offset_kernel=fd.offset_kernel.reshape(3).cpu().numpy(),
                scale_kernel=fd.scale_kernel.reshape(3).cpu().numpy())
    ref_v, ref_g = oracle.inverse_skinning(init, xc[0].cpu().numpy(), pts.cpu().numpy(), valid[0].cpu().numpy(), tfs[0].cpu().numpy(), r[0].cpu().numpy())
    assert valid.float().mean() > 0.05 and np.abs(ref_g).max() > 1
    L = _lib.lib()
    # dense layout
    t = tfs.clone().requires_grad_(True)
    out = fdm_apply(t, xc, pts[None], valid, None, None, fd)
    (out.reshape(xc.shape) * r).sum().backward()
    assert np.abs(out.reshape(xc.shape)[0].detach().cpu().numpy() - ref_v).max() < 2e-5
    assert np.linalg.norm(t.grad[0].cpu().numpy() - ref_g) / np.linalg.norm(ref_g) < 1e-4
    # compact layout: the same roots through the compacting search, candidate -> point map from the kernel
    sc = dfm.search_compact(pts)
    n = int(sc["n_cand"])
    cand_pt = torch.full((sc["cand_xc"].shape[0],), -1, dtype=torch.int32, device=DEV)
    _lib.check(L.ia_expand_candidate_points(_lib.ptr(sc["pt_off"]), _lib.ptr(sc["pt_cnt"]), pts.shape[0], None, _lib.ptr(cand_pt), cand_pt.numel(), _lib.stream()))
    want_pt = torch.repeat_interleave(torch.arange(pts.shape[0], device=DEV), sc["pt_cnt"].long())
    order = torch.argsort(sc["pt_off"].long()[want_pt] * 16 + 0, stable=True)
    assert n == int(valid.sum()) and int((cand_pt[:n] >= 0).sum()) == n
    assert torch.equal(torch.sort(cand_pt[:n].long())[0], torch.sort(want_pt)[0])
    t2 = tfs.clone().requires_grad_(True)
    out_c = fdm_apply(t2, sc["cand_xc"], pts, None, cand_pt, sc["n_cand"], fd)
    rc = torch.randn(out_c.shape, device=DEV, generator=g)
    (out_c[:n] * rc[:n]).sum().backward()
    # reference for the compact list: every candidate is a (point, root) pair; evaluate the oracle on it as a [n, 1] dense problem
    cx, cp = sc["cand_xc"][:n].cpu().numpy(), cand_pt[:n].long().cpu().numpy()
    v2, g2 = oracle.inverse_skinning(init, cx[:, None, :], pts.cpu().numpy()[cp], np.ones((n, 1), bool), tfs[0].cpu().numpy(), rc[:n].cpu().numpy()[:, None, :])
    assert np.abs(out_c[:n].detach().cpu().numpy() - v2[:, 0]).max() < 2e-5 and float(out_c[n:].detach().abs().max()) == 0.0
    assert np.linalg.norm(t2.grad[0].cpu().numpy() - g2) / np.linalg.norm(g2) < 1e-4


def fdm_apply(*a):
    from instantavatar_amd.deformers.fast_snarf.forward_deformer import _InverseSkinningFn
    return _InverseSkinningFn.apply(*a)


def test_refine_step_rendered_image_matches_golden():
    """rgb / alpha of the training render of step 0 (same rays, same jitter) within 1e-3 of the reference's."""
    model, opt, loss_fn = _setup()
    batch = _batch(0)
    rec = {}
    fwd = model.forward

    def forward_rec(*a, **k):
        rec["pred"] = fwd(*a, **k)
        return rec["pred"]
    model.forward = forward_rec
    training_step(model, batch, opt, loss_fn, is_refine=True, draws=_draws(0, model))
    rgb = rec["pred"]["rgb_coarse"].detach().reshape(-1, 3).cpu().numpy()
    alpha = rec["pred"]["alpha_coarse"].detach().reshape(-1).cpu().numpy()
    e_rgb, e_a = np.abs(rgb - G["rgb_0"]).max(-1), np.abs(alpha - G["alpha_0"])
    W.rays_within(e_rgb, "refine training_step 0 (reference golden) rgb", frac=5e-4)
    W.rays_within(e_a, "refine training_step 0 (reference golden) alpha", frac=5e-4)
    assert (G["alpha_0"] > 0.5).mean() > 0.02


def test_refine_step_replays_from_a_hip_graph():
    """The refine step (embedding look-up on the device, SMPL forward under autograd, implicit differentiation, three
    parameter groups) captured and replayed by GraphedTrainStep: same losses as eager steps from the same state, the SMPL
    tables keep moving, and a resume in the middle of an update period warms up eagerly before it captures."""
    from instantavatar_amd.training import GraphedTrainStep
    curves, tables = [], []
    for graphed in (False, True):
        model, opt, loss_fn = _setup()
        torch.manual_seed(5)
        training_step(model, _batch(0), opt, loss_fn, is_refine=True)   # step 0 (builds the occupancy grid), outside the stepper:
        stepper = GraphedTrainStep(model, opt, loss_fn, is_refine=True, enabled=graphed)   # its first call is then NOT an update step
        ls = []
        for it in range(8):
            b = _batch(it % 3)
            b["idx_dev"] = torch.tensor([it % 3], device=DEV)
            out = stepper(b)
            ls.append(float(out["loss"].detach()))
        if graphed:
            assert stepper.capture_error is None, stepper.capture_error
            assert stepper.eager_steps == 1 and stepper.replays == 7, (stepper.eager_steps, stepper.replays)
        curves.append(ls)
        tables.append(model.SMPL_param.body_pose.weight.detach().cpu().numpy().copy())
    e, g = np.array(curves[0]), np.array(curves[1])
    print("eager", e, "graph", g)
    assert np.allclose(e, g, rtol=2e-2, atol=1e-6), (e, g)
    assert np.abs(tables[1] - G["table_body_pose"]).max() > 1e-5          # the replayed steps really optimise the SMPL tables
    # eager and replayed steps end at the same tables up to what the order of the atomics does: Adam normalises every element's
    # step to ~lr = 1e-5, so an element whose gradient is noise can differ by a few lr between two runs (8 steps: <= 1.6e-4);
    # on average the difference is a small fraction of the distance travelled
    travel, diff = np.abs(tables[1] - G["table_body_pose"]), np.abs(tables[0] - tables[1])
    assert diff.mean() < 0.2 * travel.mean() and diff.max() < 1.6e-4, (diff.mean(), travel.mean(), diff.max())


def test_smpl_chain_backward_kernel_equals_autograd_through_lbs():
    """`ia_smpl_tfs_bwd` (the kinematic chain, the inverse of the root transform and Rodrigues' formula differentiated by
    hand, one launch) against autograd through the lbs.py-style torch ops (the reference's route, ~370 launches): the same
    upstream gradient d tfs -> the same gradients of the SMPL tables; and tfs itself from the kernel vs the torch ops."""
    from instantavatar_amd.deformers import snarf_deformer as sd
    res = {}
    for fused in (True, False):
        model, opt, loss_fn = _setup()
        sd.FUSED_SMPL_BACKWARD = fused
        try:
            body = model.SMPL_param(torch.tensor([1], device=DEV))
            params = {"betas": _betas(), "body_pose": body["body_pose"], "global_orient": body["global_orient"],
                      "transl": body["transl"]}
            model.deformer.prepare_deformer(params)
            tfs = model.deformer.tfs
            assert tfs.requires_grad
            w = torch.randn(tfs.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))
            w[..., 3, :] = 0
            (tfs * w).sum().backward()
            res[fused] = (tfs.detach().cpu().numpy(), {n: getattr(model.SMPL_param, n).weight.grad.detach().cpu().numpy().copy()
                                                        for n in ("body_pose", "global_orient", "transl")})
        finally:
            sd.FUSED_SMPL_BACKWARD = True
    (t1, g1), (t0, g0) = res[True], res[False]
    assert np.abs(t1 - t0).max() < 2e-5
    print("d body_pose: cos %.7f rel %.2e;  d global_orient max |kernel| %.2e |autograd| %.2e" % (
        _cos(g1["body_pose"], g0["body_pose"]), _rel(g1["body_pose"], g0["body_pose"]), np.abs(g1["global_orient"]).max(), np.abs(g0["global_orient"]).max()))
    assert _cos(g1["body_pose"], g0["body_pose"]) > 0.999999 and _rel(g1["body_pose"], g0["body_pose"]) < 1e-4
    # the bone transforms live in the SMPL-root frame: both routes give rounding noise for the root orientation / translation
    scale = np.abs(g0["body_pose"]).max()
    assert np.abs(g1["global_orient"]).max() < 1e-4 * scale and np.abs(g1["transl"]).max() < 1e-4 * scale


def test_refine_step0_with_torch_joint_chain_is_tight():
    """Step 0 of the golden with prepare_deformer's joint chain as lbs.py-style torch ops (the reference's own operation order,
    `FUSED_SMPL_BACKWARD = False`): everything downstream -- search with J_inv, field, compositing, loss, MLP / hash-grid
    backward, implicit differentiation -- then agrees with the reference's training_step to 1e-3 .. 2e-3."""
    from instantavatar_amd.deformers import snarf_deformer as sd
    sd.FUSED_SMPL_BACKWARD = False
    try:
        model, opt, loss_fn = _setup()
        losses = training_step(model, _batch(0), opt, loss_fn, is_refine=True, draws=_draws(0, model))
    finally:
        sd.FUSED_SMPL_BACKWARD = True
    assert abs(float(losses["loss"]) - float(G["loss_0"][0])) < 2e-4 * float(G["loss_0"][0])
    g = _grads(model)
    for name, key in (("d_tfs", "d_tfs_0"), ("body_pose", "g_body_pose_0"), ("mlp_sigma", "g_mlp_sigma_0"), ("mlp_color", "g_mlp_color_0")):
        c, r = _cos(g[name], G[key]), _rel(g[name], G[key])
        print(name, "cos %.6f rel %.4f" % (c, r))
        assert c > 0.9999 and r < 1e-2, (name, c, r)
