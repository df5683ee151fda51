"""SMPL body model -- host-side mirror of instant_avatar/deformers/smplx
(body_models.py:37 `SMPL`, lbs.py:152 `lbs`), reduced to what the hot path
needs: rest-pose quantities at initialisation (torch ops, any device) and the
per-frame joint chain, which runs in the HIP kernel `ia_smpl_tfs`.

The licensed SMPL pickles are not redistributable; `SMPL(model_path, gender)`
loads `<model_path>/SMPL_<GENDER>.pkl|npz` when present, `SMPL.from_dict` wraps
a synthetic body (instantavatar_amd.synthetic.make_body).
"""
import os
import pickle
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch
import torch.nn as nn


@dataclass
class SMPLOutput:  # smplx/utils.py:59 (fields used on the path)
    vertices: Optional[torch.Tensor] = None
    joints: Optional[torch.Tensor] = None
    betas: Optional[torch.Tensor] = None
    global_orient: Optional[torch.Tensor] = None
    body_pose: Optional[torch.Tensor] = None
    A: Optional[torch.Tensor] = None
    T: Optional[torch.Tensor] = None
    shape_offsets: Optional[torch.Tensor] = None
    pose_offsets: Optional[torch.Tensor] = None


def _abs_smpl(p):
    """model_path as the plugins resolve it (hydra.utils.to_absolute_path when hydra is there, the path itself otherwise)"""
    from .snarf_deformer import _abs_path
    return _abs_path(p)


def batch_rodrigues(rot_vecs):
    """lbs.py:295-329"""
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    rot_dir = rot_vecs / angle
    cos = torch.cos(angle)[:, None]
    sin = torch.sin(angle)[:, None]
    rx, ry, rz = torch.split(rot_dir, 1, dim=1)
    zeros = torch.zeros_like(rx)
    K = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view(-1, 3, 3)
    ident = torch.eye(3, dtype=rot_vecs.dtype, device=rot_vecs.device)[None]
    return ident + sin * K + (1 - cos) * torch.bmm(K, K)


def small_matmul(a, b):
    """a [..., m, k] @ b [..., k, n] for tiny m, k, n (3 or 4) as one multiply and one sum instead of a library GEMM: on this
    stack a 4 x 4 `torch.matmul` is a ~40 us rocBLAS launch (forward) plus two more in the backward pass, and the joint chain
    issues 23 of them per frame one after the other (tools/prof_fit.sh: 150 such launches = 6 ms of a 14 ms fit step)."""
    return (a.unsqueeze(-1) * b.unsqueeze(-3)).sum(-2)


def batch_rigid_transform(rot_mats, joints, parents, parents_list=None, small_ops=False):
    """lbs.py:345-401.  small_ops: the 4 x 4 products as multiply + sum (`small_matmul`) instead of library GEMMs"""
    mm = small_matmul if small_ops else torch.matmul
    joints = joints.unsqueeze(-1)
    rel = joints.clone()
    rel[:, 1:] -= joints[:, parents[1:]]
    B, N = rot_mats.shape[:2]
    tm = torch.zeros(B, N, 4, 4, dtype=rot_mats.dtype, device=rot_mats.device)
    tm[..., :3, :3] = rot_mats
    tm[..., :3, 3:] = rel
    tm[..., 3, 3] = 1
    # (parents as Python ints: reading them from a device tensor would be one host synchronisation per joint)
    par = parents_list if parents_list is not None else [int(p) for p in parents.tolist()]
    chain = [tm[:, 0]]
    for i in range(1, N):
        chain.append(mm(chain[par[i]], tm[:, i]))
    transforms = torch.stack(chain, dim=1)
    posed = transforms[:, :, :3, 3]
    jh = torch.nn.functional.pad(joints, [0, 0, 0, 1])
    rel_t = transforms - torch.nn.functional.pad(mm(transforms, jh), [3, 0, 0, 0, 0, 0, 0, 0])
    return posed, rel_t


class SMPL(nn.Module):
    NUM_JOINTS = 23
    NUM_BODY_JOINTS = 23

    def __init__(self, model_path=None, gender="neutral", data=None):
        super().__init__()
        if data is None:
            data = self._load(model_path, gender)
        # (C-contiguous copies: the pickle layout arrives transposed / Fortran-ordered, and the reductions below sum in memory
        # order -- the same data must give the same bits whichever file it came from)
        f32 = lambda a: torch.as_tensor(np.ascontiguousarray(np.asarray(a), dtype=np.float32))
        self.register_buffer("v_template", f32(data["v_template"]))
        self.register_buffer("shapedirs", f32(data["shapedirs"])[..., :10])
        posedirs = np.asarray(data["posedirs"])
        if posedirs.ndim == 3:  # pkl layout [V,3,207] -> [207, V*3] (body_models.py:152-154)
            posedirs = posedirs.reshape(-1, posedirs.shape[-1]).T
        self.register_buffer("posedirs", f32(posedirs))
        self.register_buffer("J_regressor", f32(data["J_regressor"]))
        parents = np.asarray(data["parents"] if "parents" in data else data["kintree_table"][0]).astype(np.int64)
        parents[0] = -1
        self.register_buffer("parents", torch.as_tensor(parents))
        self.parents_list = [int(p) for p in parents]
        self.register_buffer("lbs_weights", f32(data["lbs_weights"] if "lbs_weights" in data else data["weights"]))
        faces = np.asarray(data["f"]).astype(np.int64) if "f" in data else np.zeros((0, 3), np.int64)
        self.register_buffer("faces_tensor", torch.as_tensor(faces))
        self.gender = gender

    @classmethod
    def from_dict(cls, data, gender="neutral"):
        return cls(data=data, gender=gender)

    @staticmethod
    def _load(model_path, gender):
        if model_path is None:
            raise ValueError("SMPL: model_path is None (use SMPL.from_dict for a synthetic body)")
        cands = [model_path] if os.path.isfile(model_path) else [
            os.path.join(model_path, "SMPL_%s.%s" % (gender.upper(), ext)) for ext in ("npz", "pkl")]
        for p in cands:
            if os.path.exists(p):
                if p.endswith(".npz"):
                    return dict(np.load(p, allow_pickle=True))
                try:
                    with open(p, "rb") as f:
                        d = pickle.load(f, encoding="latin1")
                except ModuleNotFoundError as e:
                    # the SMPL 1.0 pickles of the SMPL / SMPLify releases store their arrays as chumpy objects: unpickling
                    # imports chumpy (the reference has it installed, requirements of smplx); it is not needed afterwards
                    if "chumpy" in str(e):
                        raise ImportError(
                            "%s stores its arrays as chumpy objects and `chumpy` is not installed.  Either install chumpy "
                            "(as the reference's environment does), or convert the file once where chumpy is available: "
                            "`d = pickle.load(open(p, 'rb'), encoding='latin1'); np.savez(p[:-4] + '.npz', **{k: "
                            "(v.toarray() if hasattr(v, 'toarray') else np.asarray(v)) for k, v in d.items() if k in "
                            "('v_template', 'shapedirs', 'posedirs', 'J_regressor', 'kintree_table', 'weights', 'f')})` -- "
                            "SMPL_<GENDER>.npz next to the .pkl is picked up first." % p) from e
                    raise
                out = {}
                for k, v in d.items():
                    if hasattr(v, "toarray"):  # scipy sparse J_regressor
                        v = v.toarray()
                    try:
                        out[k] = np.asarray(v)
                    except Exception:  # chumpy leftovers we do not need
                        pass
                return out
        raise FileNotFoundError(
            "SMPL model not found (looked for %s). The SMPL pickles are licensed and not shipped; "
            "use SMPL.from_dict(instantavatar_amd.synthetic.make_body()) for synthetic runs." % cands)

    # -- constants of the fused body-model kernels (`ia_smpl_lbs_fwd / _bwd`): per subject, per device -------------
    def lbs_constants(self):
        """(ia_smpl_body descriptor, tensors it points into): contiguous fp32 copies of the blend-shape arrays and the joint
        regression folded into J0 = J_regressor v_template [24,3] and JS = J_regressor shapedirs [24,3,10] (lbs.py:190 applied to
        v_template + shapedirs beta).  Cached; rebuilt when the module moved to another device."""
        from .. import _lib
        dev = self.v_template.device
        c = getattr(self, "_lbs_const", None)
        if c is None or c[1]["v_template"].device != dev:
            with torch.no_grad():
                keep = dict(v_template=self.v_template.float().contiguous(), shapedirs=self.shapedirs.float().contiguous(),
                            posedirs=self.posedirs.float().contiguous(), lbs_weights=self.lbs_weights.float().contiguous(),
                            J0=torch.matmul(self.J_regressor, self.v_template).float().contiguous(),
                            JS=torch.einsum("ji,ikl->jkl", self.J_regressor, self.shapedirs).float().contiguous(),
                            parents=self.parents.to(torch.int32).contiguous())
            body = _lib.SmplBody(**{k: v.data_ptr() for k, v in keep.items()}, n_verts=int(self.v_template.shape[0]))
            c = self._lbs_const = (body, keep)
        return c

    # -- shaped rest joints: constant per subject, input of ia_smpl_tfs --------
    def rest_joints(self, betas):
        v_shaped = self.v_template + torch.einsum("bl,mkl->bmk", betas, self.shapedirs)[0]
        return torch.matmul(self.J_regressor, v_shaped)  # [24,3]

    def forward(self, betas, body_pose, global_orient=None, transl=None, return_verts=True, small_ops=False):
        """body_models.py:289-372 + lbs.py:152-250 (torch ops).
        small_ops=False (initialisation: the rest pose whose vertices feed the one-time KNN voxelisation; same library calls as
        the reference, results as in rounds 1-4).  small_ops=True (the per-step caller, SMPLDeformer.prepare_deformer under
        autograd): every tiny or skinny GEMM as a broadcast multiply + sum -- a 4 x 4 `matmul` is a ~40 us rocBLAS launch here
        and the [1,207] x [207, 20 670] pose blend 0.9 ms, forward and twice backward (tools/prof_fit.sh); results differ from
        the library's by summation order (ulps)."""
        B = max(betas.shape[0], body_pose.shape[0])
        if global_orient is None:
            global_orient = torch.zeros(B, 3, dtype=betas.dtype, device=betas.device)
        full_pose = torch.cat([global_orient, body_pose], dim=1)
        if small_ops:
            v_shaped = self.v_template + (betas[:, None, None, :] * self.shapedirs[None]).sum(-1)      # einsum("bl,mkl->bmk")
            J = (self.J_regressor[None, :, :, None] * v_shaped[:, None, :, :]).sum(2)                   # einsum("bik,ji->bjk")
        else:
            v_shaped = self.v_template + torch.einsum("bl,mkl->bmk", betas, self.shapedirs)
            J = torch.einsum("bik,ji->bjk", v_shaped, self.J_regressor)
        rot = batch_rodrigues(full_pose.view(-1, 3)).view(B, -1, 3, 3)
        Jt, A = batch_rigid_transform(rot, J, self.parents, self.parents_list, small_ops=small_ops)
        verts = None
        T = None
        shape_offsets = pose_offsets = None
        if return_verts:
            ident = torch.eye(3, dtype=betas.dtype, device=betas.device)
            pose_feature = (rot[:, 1:] - ident).view(B, -1)
            if small_ops:
                pose_offsets = (pose_feature[:, :, None] * self.posedirs[None]).sum(1).view(B, -1, 3)   # lbs.py:211-222
            else:
                pose_offsets = torch.matmul(pose_feature, self.posedirs).view(B, -1, 3)
            shape_offsets = v_shaped - self.v_template                                    # lbs.py:185-187
            v_posed = v_shaped + pose_offsets
            vh = torch.cat([v_posed, torch.ones_like(v_posed[..., :1])], dim=2)
            if small_ops:   # per-vertex blend T = W A ([V,24] x [24,16]) and v' = T v as multiply + reduction
                T = (self.lbs_weights[None, :, :, None] * A.reshape(B, 1, 24, 16)).sum(2).view(B, -1, 4, 4)
                verts = (T[:, :, :3, :] * vh[:, :, None, :]).sum(-1)
            else:
                W = self.lbs_weights.unsqueeze(0).expand(B, -1, -1)
                T = torch.matmul(W, A.view(B, 24, 16)).view(B, -1, 4, 4)
                verts = torch.matmul(T, vh.unsqueeze(-1))[:, :, :3, 0]
        if transl is not None:  # body_models.py:353-360
            Jt = Jt + transl.unsqueeze(1)
            A = A.clone()
            A[..., :3, 3] += transl.unsqueeze(1)
            if verts is not None:
                verts = verts + transl.unsqueeze(1)
                T = T.clone()
                T[..., :3, 3] += transl.unsqueeze(1)
        return SMPLOutput(vertices=verts, joints=Jt, betas=betas, global_orient=global_orient,
                          body_pose=body_pose, A=A, T=T, shape_offsets=shape_offsets, pose_offsets=pose_offsets)
