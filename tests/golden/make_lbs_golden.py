"""Generates tests/golden/lbs_golden.npz by running the REFERENCE's own
instant_avatar/deformers/smplx/lbs.py (importable on CPU) on a small random
body.  Run in the build container only (needs /root/reference):

    python tests/golden/make_lbs_golden.py

The fixture pins oracle.smpl_forward / batch_rodrigues / batch_rigid_transform
("LBS: call lbs.py directly"), including the per-vertex transforms T and the
shape / pose offsets the SMPLDeformer plugin consumes (smpl_deformer.py:36-45,66-75).
"""
import importlib.util
import sys as _sys
_sys.dont_write_bytecode = True  # never write into /root/reference
import os
import sys

import numpy as np
import torch

REF = "/root/reference/instant_avatar/deformers/smplx/lbs.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    # lbs.py does `from .utils import rot_mat_to_euler, Tensor`; give it a stub package
    import types
    pkg = types.ModuleType("refsmplx")
    pkg.__path__ = []
    sys.modules["refsmplx"] = pkg
    utils = types.ModuleType("refsmplx.utils")
    utils.Tensor = torch.Tensor
    utils.rot_mat_to_euler = lambda *a, **k: None
    sys.modules["refsmplx.utils"] = utils
    spec = importlib.util.spec_from_file_location("refsmplx.lbs", REF)
    lbs_mod = importlib.util.module_from_spec(spec)
    sys.modules["refsmplx.lbs"] = lbs_mod
    spec.loader.exec_module(lbs_mod)

    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from instantavatar_amd.synthetic import SMPL_PARENTS

    g = torch.Generator().manual_seed(1234)
    V = 96
    v_template = torch.randn(V, 3, generator=g) * 0.4
    shapedirs = torch.randn(V, 3, 10, generator=g) * 0.02
    posedirs = torch.randn(207, V * 3, generator=g) * 0.01
    J_regressor = torch.rand(24, V, generator=g)
    J_regressor = J_regressor / J_regressor.sum(1, keepdim=True)
    w = torch.rand(V, 24, generator=g) ** 4
    w = w / w.sum(1, keepdim=True)
    parents = torch.as_tensor(SMPL_PARENTS.astype(np.int64))
    cases = {}
    for i in range(4):
        betas = torch.randn(1, 10, generator=g)
        pose = torch.randn(1, 72, generator=g) * (0.0 if i == 0 else 0.6)
        transl = torch.randn(1, 3, generator=g)
        verts, joints, A, T, so, po = lbs_mod.lbs(betas, pose, v_template, shapedirs, posedirs, J_regressor, parents, w)
        # SMPL.forward folds transl (body_models.py:353-360)
        A2 = A.clone(); A2[..., :3, 3] += transl.unsqueeze(1)
        T2 = T.clone(); T2[..., :3, 3] += transl.unsqueeze(1)
        cases.update({"T%d" % i: T2.numpy(), "shape_offsets%d" % i: so.numpy(), "pose_offsets%d" % i: po.numpy()})
        cases.update({"betas%d" % i: betas.numpy(), "pose%d" % i: pose.numpy(), "transl%d" % i: transl.numpy(),
                      "verts%d" % i: (verts + transl.unsqueeze(1)).numpy(), "joints%d" % i: (joints + transl.unsqueeze(1)).numpy(),
                      "A%d" % i: A2.numpy(), "rot%d" % i: lbs_mod.batch_rodrigues(pose.view(-1, 3)).numpy()})
    np.savez_compressed(os.path.join(HERE, "lbs_golden.npz"), v_template=v_template.numpy(), shapedirs=shapedirs.numpy(),
                        posedirs=posedirs.numpy(), J_regressor=J_regressor.numpy(), lbs_weights=w.numpy(),
                        parents=parents.numpy().astype(np.int32), n_cases=4, **cases)
    print("wrote", os.path.join(HERE, "lbs_golden.npz"))


if __name__ == "__main__":
    main()
