"""Shared test world: the synthetic body/field wired into the product plugins on
`device`, plus the SAME state exported as numpy for the CPU oracle, so every
stage is compared on identical inputs."""
import functools

import numpy as np
import torch

from instantavatar_amd import synthetic as syn
from instantavatar_amd.pipeline import build_synthetic_model, make_batch


@functools.lru_cache(maxsize=6)
def build(device, resolution=64, n_levels=16, blend=False):
    """blend=True: the body with non-zero shapedirs / posedirs and a dense J_regressor, initialised with synthetic.BLEND_BETAS
    (pass `betas=syn.BLEND_BETAS` to make_batch / oracle_world): what a real SMPL pickle + a data set's betas are."""
    model, body, fp = build_synthetic_model(device, resolution=resolution, n_levels=n_levels, blendshapes=blend,
                                            betas=syn.BLEND_BETAS if blend else None)
    fd = model.deformer.deformer
    init = dict(tfs_inv_t=model.deformer.tfs_inv_t[0].cpu().numpy(),
                lbs_voxel=np.ascontiguousarray(fd.lbs_voxel_final[0].cpu().numpy()),
                offset_kernel=fd.offset_kernel.reshape(3).cpu().numpy().astype(np.float32),
                scale_kernel=fd.scale_kernel.reshape(3).cpu().numpy().astype(np.float32),
                bbox=model.deformer.bbox.cpu().numpy(), D=resolution // 4, H=resolution, W=resolution)
    return model, body, fp, init


def oracle_world(orc, body, fp, init, pose72, transl, betas=None):
    return orc.make_world(body, init, fp, np.zeros(10, np.float32) if betas is None else np.asarray(betas, np.float32).reshape(10),
                          pose72[3:], pose72[:3], transl, syn.INIT_BONES)


def poses(n=4):
    return syn.procedural_pose_track(max(n, 8))


@functools.lru_cache(maxsize=4)
def build_smpl_deformer_world(device, n_levels=16, blend=False):
    """The second deformer plugin (SMPLDeformer) wired like `build`: synthetic body, a field whose
    density follows the capsule body in the deformer's TEMPLATE pose, NeRFNGPNet + Raymarcher.
    blend=True: the blend-shape body with synthetic.BLEND_BETAS (pose offsets matter here: smpl_deformer.py:36-45,66-75)."""
    from instantavatar_amd.deformers.smpl_deformer import SMPLDeformer
    from instantavatar_amd.deformers.smplx import SMPL
    from instantavatar_amd.models.networks.ngp import NeRFNGPNet
    from instantavatar_amd.pipeline import AvatarModel
    from instantavatar_amd.renderers.raymarcher_acc import Raymarcher
    body = syn.make_body(42, blendshapes=blend)
    smpl = SMPL.from_dict(body).to(device)
    deformer = SMPLDeformer(None, "neutral", threshold=0.05, k=1, body_model=smpl)
    betas = torch.as_tensor(syn.BLEND_BETAS, device=device)[None] if blend else torch.zeros(1, 10, device=device)
    deformer.initialize(betas, device)
    deformer.initialized = True
    pose_t = torch.zeros((1, 69), device=device)
    pose_t[:, 2], pose_t[:, 5] = torch.pi / 6, -torch.pi / 6
    cano = smpl(betas=betas, body_pose=pose_t, return_verts=False).joints[0].cpu().numpy()
    fp = syn.make_field(cano, deformer.bbox.cpu().numpy(), seed=42, n_levels=n_levels)
    net = NeRFNGPNet(dict(center=[0, -0.3, 0], scale=[2.5, 2.5, 2.5]), n_levels=n_levels).to(device)
    net.load_field_dict(fp)
    net.initialize(deformer.bbox)
    renderer = Raymarcher(256, 291600).to(device)
    renderer.initialize(1)
    return AvatarModel(deformer, net, renderer).to(device), body, fp


# ---- parity gates (VERDICT r04 task 3): bounds sized to what is measured, in COUNTS ------------------------------------
# Measured on MI355X over rounds 3-5 against the oracle: 0-3 of 262 144 rays beyond 1e-3 at 512^2 / 1024^2, <= 1 of 262 144
# occupancy cells flipped, 0 of 4 096 training rays.  The only legitimate run-to-run variation is the order in which atomics
# arrive (candidate / sample order -> which of two equal sigma maxima is taken first); a regression that moves hundreds of
# rays must not pass.  Small frames (32^2 ... 128^2) get a floor of 2 rays / 2 cells: one flipped cell can move a ray or two.
import math


def rays_within(err, what, tol=1e-3, frac=1e-4, floor=2):
    """at most max(floor, frac * n) entries of `err` (one per ray) exceed `tol`; prints what was measured"""
    err = np.asarray(err)
    n_bad, bound = int((err > tol).sum()), max(int(floor), int(math.ceil(frac * err.size)))
    print("GATE %-58s %d of %d rays beyond %.0e (bound %d), max %.2e" % (what, n_bad, err.size, tol, bound, float(err.max()) if err.size else 0.0))
    assert n_bad <= bound, (what, n_bad, bound, float(err.max()))
    return n_bad


def cells_within(a, b, what, frac=2e-5, floor=2):
    """at most max(floor, frac * n) cells differ between two occupancy grids (or any two equal-shape arrays)"""
    a, b = np.asarray(a), np.asarray(b)
    n_bad, bound = int((a != b).sum()), max(int(floor), int(math.ceil(frac * a.size)))
    print("GATE %-58s %d of %d cells differ (bound %d)" % (what, n_bad, a.size, bound))
    assert n_bad <= bound, (what, n_bad, bound)
    return n_bad
