"""Generates tests/golden/implicit_diff_golden.npz: the gradient w.r.t. the bone transforms that the REFERENCE's
autograd produces for the training branch of ForwardDeformer.forward (version 1: "trick for implicit diff with autodiff",
/root/reference/instant_avatar/deformers/fast_snarf/deformer_torch.py:50-67 with forward_skinning :118-128,
query_weights :190-202, skinning_mask :204-219, bmv :222-223; and version 2, :68-75: value and gradient of the closed-form
inverse skinning).

The reference module is imported on the CPU with its three JIT-compiled CUDA extensions, pytorch3d's KNN and `.cuda()`
stubbed out -- none of them is on the differentiated path: the roots, their validity and J_inv (outputs of the CUDA search,
under torch.no_grad in the reference) are supplied by the oracle's Broyden search, and the skinning-weight voxels are the
oracle's (copied into the reference's `lbs_voxel_final` buffer after its own `switch_to_explicit` has built the
normalisation closures).  What the golden pins is the reference's differentiable arithmetic end to end.

Run from the repo root in the build container:  python tests/golden/make_implicit_diff_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(HERE, "implicit_diff_golden.npz")


def main():
    from instantavatar_amd import synthetic as syn
    from oracle import oracle
    # ---- inputs from the oracle (small world of tests/test_cpu_oracle.py) ----
    body = syn.make_body()
    init = oracle.deformer_initialize(body, np.zeros(10, np.float32), syn.cano_pose("A_pose"), resolution=32, n_smooth=30)
    fp = syn.make_field(init["cano_joints"], init["bbox"])
    poses, tr = syn.procedural_pose_track(8)
    world = oracle.make_world(body, init, fp, np.zeros(10, np.float32), poses[1, 3:], poses[1, :3], tr[1], syn.INIT_BONES)
    tfs, vJ, vd = world["tfs"], world["voxel_J"], world["voxel_d"]
    rng = np.random.RandomState(3)
    v = vd.reshape(3, -1)
    sel = rng.randint(0, v.shape[1], 600)
    xd = (v[:, sel].T + 0.005 * rng.randn(600, 3)).astype(np.float32)
    x, Jinv, valid = oracle.broyden(xd, vJ, tfs, init, syn.INIT_BONES)
    keep = oracle.filter_dup(x, valid).astype(bool)
    r = rng.randn(*x.shape).astype(np.float32)

    # ---- the reference module on the CPU ----
    p3d = types.ModuleType("third_parties.pytorch3d")
    ops = types.ModuleType("third_parties.pytorch3d.ops")

    def knn_points(a, b, K=30):   # only used by switch_to_explicit; its result is overwritten below
        d = torch.cdist(a, b) ** 2
        dist, idx = d.topk(K, dim=-1, largest=False)
        return dist, idx, None
    ops.knn_points = knn_points
    p3d.ops = ops
    sys.modules["third_parties.pytorch3d"], sys.modules["third_parties.pytorch3d.ops"] = p3d, ops
    import torch.utils.cpp_extension as cpp
    cpp.load = lambda *a, **k: types.SimpleNamespace()
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)
    import instant_avatar.deformers.fast_snarf.deformer_torch as ref
    d = ref.ForwardDeformer({"version": 1})
    d.device = torch.device("cpu")
    verts = torch.as_tensor(init["vs_template"])[None]
    lbs = torch.as_tensor(np.asarray(body["lbs_weights"], np.float32))[None]
    d.switch_to_explicit(resolution=32, smpl_verts=verts, smpl_weights=lbs, use_smpl=True)
    assert tuple(d.lbs_voxel_final.shape) == (1, 24, 8, 32, 32)
    assert np.allclose(d.offset_kernel.reshape(3).numpy(), init["offset_kernel"], atol=1e-6)
    assert np.allclose(d.scale_kernel.reshape(3).numpy(), init["scale_kernel"], rtol=1e-6)
    own = d.lbs_voxel_final.clone()
    d.lbs_voxel_final.copy_(torch.as_tensor(init["lbs_voxel"])[None])
    print("reference voxel weights vs oracle's: max abs diff %.2e" % float((own - d.lbs_voxel_final).abs().max()))

    xc_t = torch.as_tensor(x)[None].clone()
    others = {"valid_ids": torch.as_tensor(keep)[None], "J_inv": torch.as_tensor(Jinv)[None], "result": xc_t}
    d.search = lambda xd_, cond, tfs_, eval_mode=False: (xc_t.clone(), others)
    tfs_t = torch.tensor(tfs[None], requires_grad=True)
    xc, _ = d.forward(torch.as_tensor(xd)[None], {}, tfs_t, eval_mode=False)
    (xc * torch.as_tensor(r)[None]).sum().backward()
    g_ref = tfs_t.grad[0].numpy()
    g_orc = oracle.implicit_diff_grad(init, x, Jinv, keep, r)
    cos = float((g_ref * g_orc).sum() / (np.linalg.norm(g_ref) * np.linalg.norm(g_orc)))
    print("valid roots %d; |g_ref| %.4f; max abs diff to the oracle's closed form %.3e; cos %.8f" % (
        int(keep.sum()), float(np.abs(g_ref).max()), float(np.abs(g_ref - g_orc).max()), cos))
    # ---- version 2 (deformer_torch.py:68-75, selected by confs/deformer/fast_snarf_debug.yaml): closed-form inverse skinning
    # x_c = R^T (x_d - t) with the blended transform of the root's skinning weights; value AND gradient differ from version 1 ----
    d.version = 2
    tfs_2 = torch.tensor(tfs[None], requires_grad=True)
    xc2, _ = d.forward(torch.as_tensor(xd)[None], {}, tfs_2, eval_mode=False)
    (xc2 * torch.as_tensor(r)[None]).sum().backward()
    g_ref2 = tfs_2.grad[0].numpy()
    v_orc2, g_orc2 = oracle.inverse_skinning(init, x, xd, keep, tfs, r)
    print("version 2: |xc2 - roots| max %.3e; value vs the oracle %.3e; |g_ref2| %.4f; gradient max abs diff to the oracle %.3e" % (
        float((xc2.detach()[0] - torch.as_tensor(x))[torch.as_tensor(keep)].abs().max()), float(np.abs(xc2.detach().numpy()[0] - v_orc2).max()),
        float(np.abs(g_ref2).max()), float(np.abs(g_ref2 - g_orc2).max())))
    np.savez_compressed(OUT, xd=xd, xc=x, J_inv=Jinv, valid=keep, r=r, tfs=tfs, grad_tfs=g_ref,
                        xc_value=xc.detach().numpy()[0], xc_value_v2=xc2.detach().numpy()[0], grad_tfs_v2=g_ref2)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
