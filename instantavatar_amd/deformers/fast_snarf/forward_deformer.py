"""ForwardDeformer: voxelised-LBS correspondence search on the GPU.

Host-side counterpart of the reference class of the same name
(instant_avatar/deformers/fast_snarf/deformer_torch.py:22).  The reference
JIT-builds three CUDA extensions at import time (deformer_torch.py:10-19);
here the same operations are C-ABI calls into libinstantavatar_hip.so:

    precompute(tfs)            -> ia_precompute          (deformer_torch.py:77-83)
    broyden_cuda(...) + filter -> ia_snarf_search        (deformer_torch.py:100-116)

`voxel_J` lives channel-LAST ([D,H,W,12], one trilinear corner = 48 contiguous
bytes); `.voxel_J` gives the reference's [1,12,D,H,W] view of the same memory.
"""
import ctypes as C

import torch
import torch.nn.functional as F

from .. import _opt
from ... import _lib

INIT_BONES = (0, 1, 2, 4, 5, 10, 11, 12, 15, 16, 17, 18, 19)  # deformer_torch.py:28
KNN_K = 30
SMOOTH_PASSES = 30
SMOOTH_BLEND = 0.7


class ForwardDeformer(torch.nn.Module):
    def __init__(self, opt, **kwargs):
        super().__init__()
        self.opt = opt
        self.init_bones = list(INIT_BONES)
        self.global_scale = 1.2
        self.version = _opt.get(opt, "version", 1)
        self.device = None
        self._bones_c = _lib.bone_array(self.init_bones)
        self._grid_c = None
        self._frame = None

    def clone_shared(self):
        """A second deformer on the SAME skinning-weight volume (buffers shared by reference) with its own per-frame
        outputs (voxel_J / voxel_d / bbox): see pipeline.PipelinedRenderer."""
        other = ForwardDeformer(self.opt)
        other.device = self.device
        for name in ("scale", "offset", "offset_kernel", "scale_kernel", "lbs_voxel_final", "grid_denorm"):
            other.register_buffer(name, getattr(self, name))
        other.resolution, other.ratio, other.bbox = self.resolution, self.ratio, self.bbox
        return other

    # -- one-time voxelisation of the skinning weights (deformer_torch.py:130-186)
    def switch_to_explicit(self, resolution=32, smpl_verts=None, smpl_weights=None, use_smpl=False):
        if not use_smpl:
            raise NotImplementedError("the path only uses SMPL-initialised weight voxels (use_smpl=True)")
        dev = self.device if self.device is not None else smpl_verts.device
        self.resolution = resolution
        d, h, w = resolution // 4, resolution, resolution
        self.ratio = h / d
        verts = smpl_verts.to(dev).float()
        lo, hi = verts.min(dim=1).values[0], verts.max(dim=1).values[0]
        centre = (lo + hi) * 0.5
        half = (hi - lo).max() / 2 * self.global_scale
        ext = torch.stack([half, half, half / self.ratio])
        self.bbox = torch.stack([centre - ext, centre + ext])
        inv = torch.stack([1.0 / half, 1.0 / half, self.ratio / half])
        for name, val in (("scale", half), ("offset", centre.view(1, 1, 3)),
                          ("offset_kernel", -centre.view(1, 1, 3)), ("scale_kernel", inv.view(1, 1, 3))):
            self.register_buffer(name, val.clone())
        # voxel centres in (d,h,w) raster order; x<->w, y<->h, z<->d
        zz, yy, xx = torch.meshgrid(torch.linspace(-1, 1, d, device=dev), torch.linspace(-1, 1, h, device=dev),
                                    torch.linspace(-1, 1, w, device=dev), indexing="ij")
        unit = torch.stack([xx, yy, zz], dim=-1).reshape(1, -1, 3)
        grid_denorm = self.denormalize(unit)
        vox = voxelise_skinning_weights(grid_denorm[0], verts[0], smpl_weights.to(dev).float()[0], (d, h, w))
        self.register_buffer("lbs_voxel_final", vox[None].contiguous())
        self.register_buffer("grid_denorm", grid_denorm)
        self._grid_c = None

    def lbs_voxel_channel_last(self):
        """the skinning-weight volume as [D,H,W,24] (made once, on first use: only the SMPL-refinement backward gathers the 24
        weights of single voxels; `precompute` streams the channel-major planes)"""
        cl = getattr(self, "_lbs_cl", None)
        if cl is None or cl.data_ptr() == 0 or getattr(self, "_lbs_cl_src", None) != self.lbs_voxel_final.data_ptr():
            cl = self._lbs_cl = self.lbs_voxel_final[0].permute(1, 2, 3, 0).contiguous()
            self._lbs_cl_src = self.lbs_voxel_final.data_ptr()
        return cl

    def normalize(self, x):
        y = (x - self.offset) / self.scale
        return torch.cat([y[..., :2], y[..., 2:] * self.ratio], dim=-1)

    def denormalize(self, x):
        y = torch.cat([x[..., :2], x[..., 2:] / self.ratio], dim=-1)
        return y * self.scale + self.offset

    def grid_desc(self):
        if self._grid_c is None:
            g = _lib.SnarfGrid()
            g.D, g.H, g.W = self.resolution // 4, self.resolution, self.resolution
            g.offset[:] = self.offset_kernel.reshape(3).float().cpu().tolist()  # init-time host read
            g.scale[:] = self.scale_kernel.reshape(3).float().cpu().tolist()
            self._grid_c = g
        return self._grid_c

    # -- per frame --------------------------------------------------------------
    def precompute(self, tfs, want_voxel_d=True, want_bbox=True):
        _lib.require_cuda(tfs, self.lbs_voxel_final)
        assert tfs.shape[0] == 1, "skinning-voxel kernels are batch-1 (so are the reference's, SURVEY 2.1)"
        d, h, w = self.resolution // 4, self.resolution, self.resolution
        fr = self._frame
        if fr is None or fr["J"].device != tfs.device or fr["J"].shape[0] != d:
            mk = lambda *s: torch.empty(s, device=tfs.device, dtype=torch.float32)
            fr = self._frame = dict(J=mk(d, h, w, 12), d=mk(1, 3, d, h, w), bbox=mk(6))
            # per-workgroup extrema of the bounding-box reduction (own buffer per deformer replica / stream)
            fr["ws"] = torch.empty(int(_lib.lib().ia_precompute_workspace_bytes(C.byref(self.grid_desc()))), dtype=torch.uint8,
                                   device=tfs.device)
        tfs_c = tfs.detach().float().contiguous()
        _lib.check(_lib.lib().ia_precompute_ws(_lib.ptr(self.lbs_voxel_final), _lib.ptr(tfs_c), _lib.ptr(fr["J"]),
                                               _lib.ptr(fr["d"]) if want_voxel_d else None, _lib.ptr(fr["bbox"]) if want_bbox else None,
                                               C.byref(self.grid_desc()), _lib.ptr(fr["ws"]), fr["ws"].numel(), _lib.stream()),
                   "ia_precompute_ws")
        # (voxel_d / bbox_deformed None: not computed for this frame -- SNARFDeformer.get_bbox_deformed recomputes on demand)
        self.voxel_J_cl, self.voxel_d, self.bbox_deformed = fr["J"], (fr["d"] if want_voxel_d else None), (fr["bbox"] if want_bbox else None)

    @property
    def voxel_J(self):
        return self.voxel_J_cl.permute(3, 0, 1, 2)[None]

    def broyden_cuda(self, xd_tgt, voxel, voxel_J_cl, tfs, cvg_thresh=1e-5, dvg_thresh=1e-1, want_J_inv=True):
        _lib.require_cuda(xd_tgt, voxel_J_cl, tfs)
        b, n, _ = xd_tgt.shape
        assert b == 1
        k = len(self.init_bones)
        mk = lambda shape, dt: torch.empty(shape, device=xd_tgt.device, dtype=dt)
        xc, valid = mk((1, n, k, 3), torch.float32), mk((1, n, k), torch.uint8)
        J_inv = mk((1, n, k, 3, 3), torch.float32) if want_J_inv else None
        xd_c, tfs_c = xd_tgt.detach().float().contiguous(), tfs.detach().float().contiguous()
        _lib.check(_lib.lib().ia_snarf_search(_lib.ptr(xd_c), n, _lib.ptr(voxel_J_cl), _lib.ptr(tfs_c), self._bones_c, k,
                                              C.byref(self.grid_desc()), cvg_thresh, dvg_thresh, _lib.ptr(xc),
                                              _lib.ptr(valid), None, _lib.ptr(J_inv), _lib.stream()), "ia_snarf_search")
        return {"result": xc, "valid_ids": valid.bool(), "J_inv": J_inv}

    def search(self, xd, cond, tfs, eval_mode=False, want_J_inv=True):
        with torch.no_grad():
            out = self.broyden_cuda(xd, self.voxel_d, self.voxel_J_cl, tfs, want_J_inv=want_J_inv)
        return out["result"], out

    def forward(self, xd, cond, tfs, eval_mode=False):
        """Canonical correspondences of xd [1,N,3] -> ([1,N,I,3], others).
        Training adds the implicit-differentiation term of deformer_torch.py:50-67
        (version 1) or the closed-form inverse skinning of :68-75 (version 2)."""
        need_grad = (not eval_mode) and tfs.requires_grad
        xc, others = self.search(xd, cond, tfs, eval_mode=True, want_J_inv=need_grad)
        if eval_mode:
            return xc, others
        mask = others["valid_ids"]
        if self.version != 1:
            # :68-75, closed-form inverse skinning of the roots (value AND gradient differ from version 1)
            if FUSED_IMPLICIT_DIFF and xc.is_cuda and tfs.shape[0] == 1:
                return _InverseSkinningFn.apply(tfs, xc.detach(), xd, mask, None, None, self).reshape(xc.shape), others
            T = torch.einsum("pn,nij->pij", self.query_weights(xc, cond, mask=mask)[mask], tfs[0])
            xd_rep = xd[..., None, :].expand(1, -1, len(self.init_bones), 3)[mask]
            out = torch.zeros_like(xc)
            out[mask] = ((xd_rep - T[:, :3, 3]).unsqueeze(-2) @ T[:, :3, :3]).squeeze(1)
            return out, others
        xc = xc.detach()  # invalid slots are already zero (ia_snarf_search writes them)
        if need_grad:
            # xc + 0 with d(xc) = -J_inv . d(skin(xc)): value unchanged, gradient to tfs
            if FUSED_IMPLICIT_DIFF and xc.is_cuda and tfs.shape[0] == 1:
                xc = _ImplicitDiffFn.apply(tfs, xc, others["J_inv"], mask, self)
            else:
                skinned = self.forward_skinning(xc, cond, tfs, mask=mask)
                delta = skinned - skinned.detach()
                xc = xc.clone()
                xc[mask] = xc[mask] + bmv(-others["J_inv"][mask], delta.unsqueeze(-1)).squeeze(-1)
        return xc, others

    def forward_skinning(self, xc, cond, tfs, mask=None):
        w = self.query_weights(xc, cond, mask=mask)
        return skinning_mask(xc[mask], w[mask], tfs)

    def query_weights(self, xc, cond=None, mask=None, mode="bilinear"):
        lead = xc.shape[:-1]
        g = self.normalize(xc.reshape(1, -1, 3))[:, :, None, None]
        w = F.grid_sample(self.lbs_voxel_final, g, align_corners=True, mode=mode, padding_mode="border")
        return w[0, :, :, 0, 0].T.reshape(*lead, -1)


#: a7 through the HIP kernel `ia_snarf_implicit_bwd`; False = the reference's torch-op formulation
#: (grid_sample + einsum + batched mat-vec under autograd), kept as the checker of the kernel
FUSED_IMPLICIT_DIFF = True


class _ImplicitDiffFn(torch.autograd.Function):
    """x_c* with the gradient of  x_c* - J_inv (d(x_c*) - sg[d(x_c*)])  w.r.t. the bone transforms."""

    @staticmethod
    def forward(ctx, tfs, xc, J_inv, mask, deformer):
        ctx.deformer = deformer
        ctx.save_for_backward(xc, J_inv, mask)
        ctx.tfs_shape = tfs.shape
        return xc.clone()

    @staticmethod
    def backward(ctx, g):
        xc, J_inv, mask = ctx.saved_tensors
        d = ctx.deformer
        L = _lib.lib()
        n = mask.numel()
        x = xc.reshape(-1, 3).float().contiguous()
        J = J_inv.reshape(-1, 9).float().contiguous()
        m = mask.reshape(-1).to(torch.uint8).contiguous()
        gg = g.reshape(-1, 3).float().contiguous()
        d_tfs = torch.zeros(ctx.tfs_shape, device=x.device)
        ws = torch.empty(int(L.ia_snarf_implicit_bwd_workspace_bytes(n)), dtype=torch.uint8, device=x.device)
        _lib.check(L.ia_snarf_implicit_bwd(_lib.ptr(x), _lib.ptr(J), _lib.ptr(m), _lib.ptr(gg), n,
                                           _lib.ptr(d.lbs_voxel_final), C.byref(d.grid_desc()), _lib.ptr(d_tfs),
                                           _lib.ptr(ws), ws.numel(), _lib.stream()), "ia_snarf_implicit_bwd")
        return d_tfs, None, None, None, None


class _InverseSkinningFn(torch.autograd.Function):
    """`version: 2` (deformer_torch.py:68-75): x_c = R^T (x_d - t), T = sum_n w_n(x_c*) tfs_n, as the kernels
    `ia_snarf_inverse_skinning` / `ia_snarf_inverse_skinning_bwd`.  Two layouts: dense (xc [1,P,I,3] with `mask` [1,P,I], the
    entry's target is xd[e // I]) and compact (xc [cap,3] with the live count `n_dev` and `cand_pt` [cap], the sample point of
    every candidate).  The roots carry no gradient (the reference searches under no_grad); tfs does, and so does the target
    x_d: in the refine configuration the sample points are rays in the SMPL-root frame, i.e. functions of w2s."""

    @staticmethod
    def forward(ctx, tfs, xc, xd, mask, cand_pt, n_dev, deformer):
        L = _lib.lib()
        x = xc.reshape(-1, 3).float().contiguous()
        t = xd.reshape(-1, 3).float().contiguous()
        m = mask.reshape(-1).to(torch.uint8).contiguous() if mask is not None else None
        n_init = xc.shape[-2] if mask is not None else 1
        tf = tfs.detach().reshape(-1, 4, 4).float().contiguous()
        out = torch.empty_like(x)
        _lib.check(L.ia_snarf_inverse_skinning(_lib.ptr(x), _lib.ptr(t), _lib.ptr(cand_pt), n_init, _lib.ptr(m), x.shape[0], _lib.ptr(n_dev),
                                               _lib.ptr(deformer.lbs_voxel_channel_last()), 1, C.byref(deformer.grid_desc()), _lib.ptr(tf),
                                               _lib.ptr(out), _lib.stream()), "ia_snarf_inverse_skinning")
        ctx.deformer, ctx.n_init, ctx.tfs_shape, ctx.xd_shape = deformer, n_init, tfs.shape, xd.shape
        ctx.save_for_backward(x, t, m, cand_pt, n_dev, tf)
        return out

    @staticmethod
    def backward(ctx, g):
        x, t, m, cand_pt, n_dev, tf = ctx.saved_tensors
        d, L = ctx.deformer, _lib.lib()
        gg = g.reshape(-1, 3).float().contiguous()
        d_tfs = torch.zeros(ctx.tfs_shape, device=x.device)
        want_dx = ctx.needs_input_grad[2]
        d_e = torch.empty_like(x) if want_dx else None
        ws = torch.empty(int(L.ia_snarf_implicit_bwd_workspace_bytes(x.shape[0])), dtype=torch.uint8, device=x.device)
        _lib.check(L.ia_snarf_inverse_skinning_bwd(_lib.ptr(x), _lib.ptr(t), _lib.ptr(cand_pt), ctx.n_init, _lib.ptr(m), _lib.ptr(gg), x.shape[0],
                                                   _lib.ptr(n_dev), _lib.ptr(d.lbs_voxel_channel_last()), 1, C.byref(d.grid_desc()), _lib.ptr(tf),
                                                   _lib.ptr(d_tfs), _lib.ptr(d_e), _lib.ptr(ws), ws.numel(), _lib.stream()), "ia_snarf_inverse_skinning_bwd")
        d_xd = None
        if want_dx:   # sum the entries of every point: the n_init slots of the dense layout, or the candidates that name the point
            if cand_pt is None:
                d_xd = d_e.reshape(-1, ctx.n_init, 3).sum(1).reshape(ctx.xd_shape)
            else:
                d_xd = torch.zeros((t.shape[0], 3), device=x.device).index_add_(0, cand_pt.long().clamp_(0, t.shape[0] - 1), d_e).reshape(ctx.xd_shape)
        return d_tfs, None, d_xd, None, None, None, None


class _ImplicitDiffCompactFn(torch.autograd.Function):
    """`_ImplicitDiffFn` over the compact candidate list of `ia_snarf_search_compact_jinv`: value = the roots,
    gradient to tfs through `ia_snarf_implicit_bwd_compact` (live count on the device, no mask, no host read)."""

    @staticmethod
    def forward(ctx, tfs, cand_xc, cand_Jinv, n_cand, deformer):
        ctx.deformer = deformer
        ctx.save_for_backward(cand_xc, cand_Jinv, n_cand)
        ctx.tfs_shape = tfs.shape
        return cand_xc.clone()

    @staticmethod
    def backward(ctx, g):
        xc, J_inv, n_cand = ctx.saved_tensors
        d = ctx.deformer
        L = _lib.lib()
        cap = xc.shape[0]
        gg = g.reshape(-1, 3).float().contiguous()
        d_tfs = torch.zeros(ctx.tfs_shape, device=xc.device)
        ws = torch.empty(int(L.ia_snarf_implicit_bwd_workspace_bytes(cap)), dtype=torch.uint8, device=xc.device)
        _lib.check(L.ia_snarf_implicit_bwd_compact(_lib.ptr(xc), _lib.ptr(J_inv), _lib.ptr(gg), cap, _lib.ptr(n_cand),
                                                   _lib.ptr(d.lbs_voxel_channel_last()), 1, C.byref(d.grid_desc()), _lib.ptr(d_tfs),
                                                   _lib.ptr(ws), ws.numel(), _lib.stream()), "ia_snarf_implicit_bwd_compact")
        return d_tfs, None, None, None, None


def skinning_mask(x, w, tfs, inverse=False):
    """x [P,3], w [P,24], tfs [1,24,4,4] -> skinned points [P,3]."""
    T = torch.einsum("pn,nij->pij", w, tfs[0])
    return (T[:, :3, :3] @ x[:, :, None])[:, :, 0] + T[:, :3, 3]


def bmv(m, v):
    return m @ v


def voxelise_skinning_weights(points, verts, vert_weights, dims):
    """deformer_torch.py:225-244 on the GPU: `ia_voxelise_weights` (exact brute-force 30-NN with
    the vertices staged in LDS + inverse-distance blend + 30 smoothing passes).  There is no CPU route:
    CPU tensors raise (the checkers of this kernel live in oracle/ and tests/)."""
    _lib.require_cuda(points)
    d, h, w = dims
    L = _lib.lib()
    pts, vs, vw = points.float().contiguous(), verts.float().contiguous(), vert_weights.float().contiguous()
    out = torch.empty((24, d, h, w), device=points.device)
    ws = torch.empty(int(L.ia_voxelise_workspace_bytes(d, h, w)), dtype=torch.uint8, device=points.device)
    _lib.check(L.ia_voxelise_weights(_lib.ptr(pts), _lib.ptr(vs), vs.shape[0], _lib.ptr(vw), d, h, w, SMOOTH_PASSES,
                                     _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.stream()), "ia_voxelise_weights")
    return out
