"""GPU gates (-m gpu) on a body with NON-ZERO blend shapes (VERDICT r05 missing 3 / weak 2).

Rounds 1-5 ran every device test on `synthetic.make_body()` -- zero shapedirs / posedirs, a one-hot-ring J_regressor -- so betas
were a no-op and the round-5 rewrite of the SMPL body model's tiny / skinny GEMMs as broadcast multiply + sum
(`SMPL.forward(small_ops=True)`, deformers/smplx.py) only ever multiplied zeros on the device.  A real SMPL pickle is exactly
the other configuration class: lbs.py:185-222 (shape blend, joint regression from the SHAPED vertices, pose-corrective blend)
with non-zero operands.  Here, on `synthetic.make_body(blendshapes=True)` + `synthetic.BLEND_BETAS` and on the body of
tests/golden/lbs_golden.npz (the reference's lbs.py executed on the CPU):

  * `SMPL.forward` on the device, library route and small_ops route, and the `ia_smpl_tfs` joint-chain kernel against the
    reference's lbs.py (lbs_golden.npz) and against the oracle on the 6 890-vertex body;
  * `SNARFDeformer.initialize` (betas-dependent rest pose -> voxelised skinning weights) against the oracle and against the
    reference's Python (pipeline_golden_blend.npz);
  * one 512^2 frame per launch mode (eager / HIP graph / two in flight) with non-zero betas against the oracle, count gates of
    tests/world.py;
  * SMPLDeformer's per-step `prepare_deformer` under autograd: the small_ops route against the library route, values and the
    gradients w.r.t. betas / pose / translation (the fit stage optimises betas: DNeRF.py:121-123);
  * a fit step on a blend-shape subject.
The refine training steps, the reference-Python frame / occupancy update / SMPLDeformer frame and the SMPLDeformer kernel
tests run on both bodies through their own parametrised fixtures (test_gpu_refine.py, test_gpu_refpython.py,
test_gpu_smpl_deformer.py)."""
import os

import numpy as np
import pytest
import torch

from instantavatar_amd import _lib, synthetic as syn
from instantavatar_amd.deformers.smplx import SMPL
from instantavatar_amd.pipeline import build_synthetic_model, make_batch

import world as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))


def test_smpl_forward_on_device_matches_reference_lbs_golden():
    """The reference's lbs.py (executed on the CPU: tests/golden/make_lbs_golden.py, random shapedirs 0.02 / posedirs 0.01, dense
    J_regressor, four (betas, pose, transl) cases) against `SMPL.forward` ON THE DEVICE -- both routes -- and against the
    joint-chain kernel `ia_smpl_tfs` fed with the betas-dependent rest joints."""
    g = np.load(os.path.join(HERE, "golden", "lbs_golden.npz"))
    smpl = SMPL.from_dict(dict(v_template=g["v_template"], shapedirs=g["shapedirs"], posedirs=g["posedirs"],
                               J_regressor=g["J_regressor"], lbs_weights=g["lbs_weights"], parents=g["parents"])).to(DEV)
    t = lambda a: torch.as_tensor(a, device=DEV)
    ident = torch.eye(4, device=DEV).repeat(1, 24, 1, 1).contiguous()
    for i in range(int(g["n_cases"])):
        pose = t(g["pose%d" % i])
        for small_ops in (False, True):
            out = smpl(t(g["betas%d" % i]), pose[:, 3:], pose[:, :3], t(g["transl%d" % i]), small_ops=small_ops)
            for name, ref, tol in (("A", "A", 2e-5), ("vertices", "verts", 2e-5), ("T", "T", 2e-5), ("joints", "joints", 2e-5),
                                   ("shape_offsets", "shape_offsets", 1e-6), ("pose_offsets", "pose_offsets", 1e-6)):
                err = float((getattr(out, name).cpu() - torch.as_tensor(g["%s%d" % (ref, i)])).abs().max())
                assert err < tol, (i, small_ops, name, err)
        assert float(out.shape_offsets.abs().max()) > 1e-2 and (i == 0 or float(out.pose_offsets.abs().max()) > 1e-3)   # non-zero operands
        # the kernel: rest joints from the shaped vertices (betas != 0), identity rest transform -> tfs = inv(A_0) A
        jr = smpl.rest_joints(t(g["betas%d" % i])).contiguous()
        tfs, w2s, A = torch.empty(1, 24, 4, 4, device=DEV), torch.empty(1, 4, 4, device=DEV), torch.empty(1, 24, 4, 4, device=DEV)
        # (named tensors: a temporary handed to _lib.ptr() is freed at once and the NEXT temporary may reuse its block)
        par32, pose72, tr3 = smpl.parents.to(torch.int32).contiguous(), pose.reshape(1, 72).contiguous(), t(g["transl%d" % i]).reshape(3).contiguous()
        _lib.check(_lib.lib().ia_smpl_tfs(_lib.ptr(jr), _lib.ptr(par32), _lib.ptr(pose72), _lib.ptr(tr3), _lib.ptr(ident), _lib.ptr(tfs), _lib.ptr(w2s),
                                          _lib.ptr(A), _lib.stream()), "ia_smpl_tfs")
        A_ref = g["A%d" % i][0].astype(np.float64)
        w2s_ref = np.linalg.inv(A_ref[0])
        assert np.abs(A[0].cpu().numpy() - A_ref).max() < 2e-5, i
        assert np.abs(w2s[0].cpu().numpy() - w2s_ref).max() < 2e-5, i
        assert np.abs(tfs[0].cpu().numpy() - w2s_ref[None] @ A_ref).max() < 5e-5, i


@pytest.fixture(scope="module")
def blend_world(oracle):
    model, body, fp, init = W.build(DEV, 64, 16, blend=True)
    poses, tr = W.poses()
    return model, body, fp, init, poses, tr


def test_joint_chain_kernel_and_small_ops_forward_on_the_blend_body(oracle, blend_world):
    model, body, fp, init, poses, tr = blend_world
    betas = syn.BLEND_BETAS
    smpl = model.deformer.body_model
    bt = torch.as_tensor(betas, device=DEV)[None]
    # the shape coefficients move the rest joints by centimetres: the kernel's input really depends on them
    j0 = smpl.rest_joints(torch.zeros(1, 10, device=DEV))
    jb = smpl.rest_joints(bt)
    assert float((jb - j0).abs().max()) > 0.02
    assert torch.equal(model.deformer._joints_rest, jb.contiguous())
    for i in (0, 3, 5):
        model.deformer.prepare_deformer(make_batch(DEV, 64, poses[i], tr[i], betas=betas))
        tfs, w2s = oracle.prepare_deformer(body, init, betas, poses[i, 3:], poses[i, :3], tr[i])
        assert np.abs(model.deformer.tfs[0].cpu().numpy() - tfs).max() < 2e-5, i
        assert np.abs(model.deformer.w2s[0].cpu().numpy() - w2s).max() < 2e-5, i
        # ... and it is NOT what zero betas give (a test that passes with betas ignored is no test)
        tfs0, _ = oracle.prepare_deformer(body, init, np.zeros(10, np.float32), poses[i, 3:], poses[i, :3], tr[i])
        assert np.abs(tfs0 - tfs).max() > 1e-2
        ref = oracle.smpl_forward(body, betas, poses[i, 3:], poses[i, :3], tr[i])
        t = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=DEV)[None]
        for small_ops in (False, True):
            out = smpl(bt, t(poses[i, 3:]), t(poses[i, :3]), t(tr[i]), small_ops=small_ops)
            for name, key, tol in (("A", "A", 2e-5), ("joints", "joints", 2e-5), ("vertices", "vertices", 2e-5), ("T", "T", 2e-5),
                                   ("shape_offsets", "shape_offsets", 2e-6), ("pose_offsets", "pose_offsets", 2e-6)):
                err = float(np.abs(getattr(out, name)[0].cpu().numpy() - ref[key]).max())
                assert err < tol, (i, small_ops, name, err)
        assert np.abs(ref["pose_offsets"]).max() > 1e-3 and np.abs(ref["shape_offsets"]).max() > 1e-2
    # the posed vertices the smpl_init occupancy bootstrap reads (snarf_deformer.py:89): body model -> SMPL-root frame
    v = model.deformer.vertices[0].cpu().numpy()
    _, w2s = oracle.prepare_deformer(body, init, betas, poses[5, 3:], poses[5, :3], tr[5])
    v_ref = ref["vertices"] @ w2s[:3, :3].T + w2s[:3, 3]
    assert np.abs(v - v_ref).max() < 5e-5


def test_initialize_with_betas_matches_oracle_and_reference_python(oracle):
    """a20 on the blend-shape body: the rest pose depends on betas (shape blend) AND on the canonical pose's pose-corrective
    blend -> vertices -> bbox, KNN voxelisation.  Same gates as test_voxelise_kernel_matches_oracle_deformer_initialize."""
    res = 32
    model, body, fp = build_synthetic_model(DEV, resolution=res, blendshapes=True, betas=syn.BLEND_BETAS)
    init = oracle.deformer_initialize(body, syn.BLEND_BETAS, syn.cano_pose("A_pose"), resolution=res, n_smooth=30)
    init0 = oracle.deformer_initialize(syn.make_body(), np.zeros(10, np.float32), syn.cano_pose("A_pose"), resolution=res, n_smooth=30)
    assert np.abs(init["vs_template"] - init0["vs_template"]).max() > 0.02        # the blend shapes really moved the rest pose
    fd = model.deformer.deformer
    assert np.abs(model.deformer.vs_template[0].cpu().numpy() - init["vs_template"]).max() < 2e-5
    own = fd.lbs_voxel_final[0].cpu().numpy()
    err = np.abs(own - init["lbs_voxel"]).max(0)
    print("blend-shape initialisation: voxels off by > 1e-4: %.2e, max %.2e" % ((err > 1e-4).mean(), err.max()))
    assert (err > 1e-4).mean() < 2e-3 and np.median(err) < 1e-6
    assert np.abs(own.sum(0) - 1).max() < 1e-5 and own.min() >= 0
    assert np.abs(fd.offset_kernel.reshape(3).cpu().numpy() - init["offset_kernel"]).max() < 2e-6
    assert np.abs(fd.scale_kernel.reshape(3).cpu().numpy() - init["scale_kernel"]).max() < 1e-5
    assert np.abs(model.deformer.bbox.cpu().numpy() - init["bbox"]).max() < 2e-6
    assert np.abs(model.deformer.tfs_inv_t[0].cpu().numpy() - init["tfs_inv_t"]).max() < 1e-5
    # the reference's Python (snarf_deformer.py:41-69 + deformer_torch.py:130-202) on the same body and betas
    g = np.load(os.path.join(HERE, "golden", "pipeline_golden_blend%s.npz" % ("_l55" if os.environ.get("IA_TCNN_LEVEL3_RES", "54") == "55" else "")))
    assert np.abs(model.deformer.bbox.cpu().numpy() - g["bbox"]).max() < 2e-6
    d = np.abs(own.reshape(24, -1)[:, ::97] - g["D_lbs_sample"]).max(0)
    assert (d > 1e-4).mean() < 4e-3 and np.median(d) < 1e-6, ((d > 1e-4).mean(), d.max())


@pytest.mark.parametrize("blend", [True, False], ids=["blendshapes", "zero-blendshapes"])
def test_smpl_deformer_prepare_three_routes_values_and_gradients(blend):
    """SMPLDeformer.prepare_deformer under autograd (the fit stage's per-step path, smpl_deformer.py:32-77), three routes:
      fused      `ia_smpl_lbs_fwd / _bwd` (what the product runs: 2 + 4 launches, csrc/ia_smpl_lbs.hip)
      small_ops  SMPL.forward with its tiny GEMMs as multiply + sum, under autograd (round 5's route)
      library    SMPL.forward with the library GEMMs / einsum of the reference's lbs.py, under autograd
    T_inv, posed vertices, w2s, the template bounding box, and the gradients of a random functional of T_inv AND w2s (the ray
    frame: transform_rays_w2s is differentiable in the reference) w.r.t. betas, body pose, root orientation and translation.
    With zero blend shapes the betas / pose-offset terms of these gradients vanish; on the blend-shape body they do not."""
    from instantavatar_amd.deformers import smpl_deformer as sdm
    from instantavatar_amd.deformers.smpl_deformer import SMPLDeformer
    body = syn.make_body(blendshapes=blend)
    betas0 = syn.BLEND_BETAS if blend else np.zeros(10, np.float32)
    poses, tr = W.poses()
    res = {}
    for route in ("fused", "small_ops", "library"):
        smpl = SMPL.from_dict(body).to(DEV)
        if route == "library":   # SMPL.forward's default
            fwd = smpl.forward
            smpl.forward = lambda *a, **k: fwd(*a, **{**k, "small_ops": False})
        d = SMPLDeformer(None, "neutral", threshold=0.05, k=1, body_model=smpl)
        leaf = {"betas": torch.tensor(betas0[None], device=DEV, requires_grad=True),
                "body_pose": torch.tensor(poses[2][None, 3:], device=DEV, requires_grad=True),
                "global_orient": torch.tensor(poses[2][None, :3], device=DEV, requires_grad=True),
                "transl": torch.tensor(tr[2][None], device=DEV, requires_grad=True)}
        old = sdm.FUSED_LBS
        sdm.FUSED_LBS = route == "fused"
        try:
            d.prepare_deformer(leaf)                      # (not `initialized`: the template is rebuilt from these betas, as in the reference)
        finally:
            sdm.FUSED_LBS = old
        gen = torch.Generator(device=DEV).manual_seed(3)
        w1 = torch.randn(d.T_inv.shape, device=DEV, generator=gen)
        w1[..., 3, :] = 0
        w3 = torch.randn(d.w2s.shape, device=DEV, generator=gen) * 50.0
        w3[..., 3, :] = 0
        ((d.T_inv * w1).sum() + (d.w2s * w3).sum()).backward()
        res[route] = (d.T_inv.detach().cpu().numpy(), d.vertices.detach().cpu().numpy(), d.w2s.detach().cpu().numpy(), d.bbox.cpu().numpy(),
                      {k: v.grad.detach().cpu().numpy().copy() for k, v in leaf.items()})
    T0, v0, w0_, bb0, g0 = res["library"]
    for route in ("fused", "small_ops"):
        T1, v1, w1_, bb1, g1 = res[route]
        assert np.abs(T1 - T0).max() < 5e-5 and np.abs(v1 - v0).max() < 2e-5 and np.abs(w1_ - w0_).max() < 1e-5 and np.abs(bb1 - bb0).max() < 2e-5, route
        for k in g0:
            a, b = g1[k].astype(np.float64).reshape(-1), g0[k].astype(np.float64).reshape(-1)
            nb = np.linalg.norm(b)
            if nb == 0:
                assert np.linalg.norm(a) == 0, (route, k)      # (betas on the zero-blendshape body: J does not depend on them)
                continue
            cos = float((a * b).sum() / (np.linalg.norm(a) * nb))
            rel = float(np.linalg.norm(a - b) / nb)
            print("%-9s d %-13s |g| %.3e  cos %.8f  rel %.2e" % (route, k, nb, cos, rel))
            assert cos > 0.99999 and rel < 2e-3, (route, k, cos, rel)
    if blend:
        assert np.linalg.norm(res["fused"][4]["betas"]) > 1e-2


def test_fit_step_on_a_blend_shape_subject_moves_betas():
    """fit stage (fit.py, deformer=smpl) on the blend-shape subject: the betas row is optimised together with the poses
    (DNeRF.py:121-123 hands the table's betas to the SMPLDeformer) -- finite, non-zero gradient, loss decreases."""
    from instantavatar_amd.drivers import fit as fit_driver
    from instantavatar_amd.training import NGPLoss, configure_optimizer, training_step
    torch.manual_seed(0)
    frames, body_model, true = fit_driver.synthetic_frames(torch.device(DEV), res=96, n_frames=3, noise=0.03, patch=16, blendshapes=True)
    assert np.abs(true["betas"]).max() > 0.5 and float(body_model.shapedirs.abs().max()) > 0.01 and float(body_model.posedirs.abs().max()) > 0.005
    model = fit_driver.build_fit_model(frames, body_model, torch.device(DEV))
    opt = configure_optimizer(model, lr=1e-3, smpl_lr=1e-4)
    loss_fn = NGPLoss(dict(w_rgb=1.0, w_alpha=0.1, w_reg=0.1, w_lpips=0.0, w_depth_reg=0.01))
    model.train()
    b0 = model.SMPL_param.betas.weight.detach().clone()
    hist = []
    for it in range(45):
        losses = training_step(model, frames.batch(it % 3), opt, loss_fn)
        hist.append(float(losses["mse_loss"].detach()))
        if it == 0:
            g = model.SMPL_param.betas.weight.grad
            assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0
    assert np.isfinite(hist).all() and np.mean(hist[-9:]) < np.mean(hist[:9]), (hist[:9], hist[-9:])
    assert float((model.SMPL_param.betas.weight.detach() - b0).abs().max()) > 1e-4
