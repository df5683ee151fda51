"""SMPLDeformer plugin (drop-in for instant_avatar/deformers/smpl_deformer.py:20-137), the second
deformer behind the `confs/deformer` surface (`confs/deformer/smpl.yaml`): a point is moved to
canonical space by the inverse transform of its NEAREST posed SMPL vertex if that vertex is closer
than `threshold`.

    SMPLDeformer(model_path, gender, threshold=0.05, k=1)
    .initialize(betas, device)   .prepare_deformer(smpl_params)   .transform_rays_w2s(rays)
    .get_bbox_deformed()         .deform(pts) -> (pts_cano, valid)
    .deform_test / .deform_train / __call__(pts, model, eval_mode)
    attributes: bbox, vertices, w2s, T_inv, initialized, body_model

The nearest-vertex search (pytorch3d knn_points in the reference) and the per-point transform are
the HIP kernel `ia_smpl_nn_deform`; `deformer(pts, net)` is the fused `ia_smpl_deform_query`
(search + compaction + field + scatter, no host synchronisation) when `net` is an
instantavatar_amd NeRFNGPNet, and the reference's masked structure for any other callable.
Only k = 1 is supported (the reference squeezes the neighbour axis: "we use the nearest neighbor
only", smpl_deformer.py:98-100).
"""
import ctypes as C

import torch

from .. import _lib
from .smplx import SMPL
from .snarf_deformer import _abs_path, get_bbox_from_smpl


class SMPLDeformer():
    def __init__(self, model_path, gender, threshold=0.05, k=1, body_model=None) -> None:
        # body_model: optional pre-built SMPL (e.g. SMPL.from_dict(synthetic.make_body()))
        self.body_model = body_model if body_model is not None else SMPL(_abs_path(model_path), gender=gender)
        if k != 1:
            raise NotImplementedError("SMPLDeformer: only k = 1 (nearest vertex) is implemented, as used by the reference")
        self.k = k
        self.threshold = threshold
        self.strategy = "nearest_neighbor"
        self.initialized = False
        self._ws = None

    def initialize(self, betas, device):
        """smpl_deformer.py:32-45: template pose (legs apart), its per-vertex transforms and offsets."""
        batch_size = betas.shape[0]
        body_pose_t = torch.zeros((batch_size, 69), device=device)
        body_pose_t[:, 2] = torch.pi / 6
        body_pose_t[:, 5] = -torch.pi / 6
        out = self.body_model(betas=betas, body_pose=body_pose_t)
        self.bbox = get_bbox_from_smpl(out.vertices[0:1].detach())
        self.T_template = out.T
        self.vs_template = out.vertices
        self.pose_offset_t = out.pose_offsets
        self.shape_offset_t = out.shape_offsets

    def get_bbox_deformed(self):
        return get_bbox_from_smpl(self.vertices[0:1].detach())

    def prepare_deformer(self, smpl_params):
        """smpl_deformer.py:50-77"""
        device = smpl_params["betas"].device
        if next(self.body_model.buffers()).device != device:
            self.body_model = self.body_model.to(device)
        if not self.initialized:
            self.initialize(smpl_params["betas"], device)  # the reference re-initialises every frame (betas may change)
        out = self.body_model(betas=smpl_params["betas"], body_pose=smpl_params["body_pose"],
                              global_orient=smpl_params["global_orient"], transl=smpl_params["transl"], small_ops=True)
        s2w = out.A[:, 0]
        # (smpl_deformer.py:69-70 uses torch.inverse twice: the LU route reads its status back to the host and, for the 6 890
        # per-vertex transforms, is a chain of library kernels; both matrices are affine -- closed form, differentiable)
        from .snarf_deformer import affine_inverse
        from .smplx import small_matmul
        w2s = affine_inverse(s2w)
        T_inv = small_matmul(affine_inverse(out.T.float()), s2w[:, None])
        T_inv[..., :3, 3] += self.pose_offset_t - out.pose_offsets   # remove & re-apply the blend shapes
        T_inv[..., :3, 3] += self.shape_offset_t - out.shape_offsets
        self.T_inv = small_matmul(self.T_template, T_inv).float().contiguous()
        self.vertices = ((out.vertices @ w2s[:, :3, :3].permute(0, 2, 1)) + w2s[:, None, :3, 3]).float().contiguous()
        self.w2s = w2s

    def release_graph(self):
        """drop the autograd graph held by the per-frame attributes (see SNARFDeformer.release_graph)"""
        for name in ("T_inv", "vertices", "w2s"):
            v = getattr(self, name, None)
            if torch.is_tensor(v) and v.requires_grad:
                setattr(self, name, v.detach())

    def transform_rays_w2s(self, rays):
        """smpl_deformer.py:79-86 (same as SNARFDeformer.transform_rays_w2s): fused kernel."""
        from .snarf_deformer import SNARFDeformer
        return SNARFDeformer.transform_rays_w2s(self, rays)

    # ------------------------------------------------------------------ queries
    def _check_batch(self):
        if self.vertices.shape[0] != 1:
            raise NotImplementedError("SMPLDeformer: batch size 1 (one frame per step, peoplesnapshot.py:171)")

    def deform(self, pts):
        """smpl_deformer.py:88-110 -> (pts_cano [P,3], valid [P] bool)"""
        self._check_batch()
        _lib.require_cuda(pts)
        x = pts.detach().reshape(-1, 3).float().contiguous()
        P = x.shape[0]
        cano = torch.empty((P, 3), device=x.device)
        valid = torch.empty(P, dtype=torch.uint8, device=x.device)
        need_grad = torch.is_grad_enabled() and (self.T_inv.requires_grad or pts.requires_grad)
        idx = torch.empty(P, dtype=torch.int32, device=x.device) if need_grad else None
        if P:
            _lib.check(_lib.lib().ia_smpl_nn_deform(_lib.ptr(x), P, None, _lib.ptr(self.vertices.detach()), _lib.ptr(self.T_inv.detach()),
                                                    self.vertices.shape[1], float(self.threshold), _lib.ptr(cano),
                                                    _lib.ptr(valid), _lib.ptr(idx), _lib.stream()), "ia_smpl_nn_deform")
        if need_grad and P:
            # SMPL refinement with this deformer (SNARF_NGP_refine + deformer=smpl): the reference's pts_cano
            # (smpl_deformer.py:100-107) is differentiable w.r.t. the per-vertex transforms; the nearest-vertex
            # index comes from the kernel, the affine map is re-applied in torch so that autograd reaches T_inv
            # (index_select: its backward is an index_add, not the sort-based indexing backward of T_inv[0][idx] -- 250 us a call;
            # the 3 x 3 products as multiply + sum: a batched GEMM over ~10^5 tiny matrices is 260 us forward and twice that back)
            T = torch.index_select(self.T_inv[0], 0, idx.long())
            xh = pts.reshape(-1, 3).float()
            cano = (T[:, :3, :3] * xh[:, None, :]).sum(-1) + T[:, :3, 3]
        return cano, valid.bool()

    def _workspace(self, nbytes, device):
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != device:
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        return self._ws

    def _query_fused(self, pts, net, fill, nan_to_num):
        self._check_batch()
        x = pts.detach().reshape(-1, 3).float().contiguous()
        P = x.shape[0]
        rgb = torch.zeros((P, 3), device=x.device)
        sigma = torch.full((P,), float(fill), device=x.device)
        if P == 0:
            return rgb, sigma
        L = _lib.lib()
        ws = self._workspace(L.ia_smpl_query_workspace_bytes(P), x.device)
        _lib.check(L.ia_smpl_deform_query(_lib.ptr(x), P, None, _lib.ptr(self.vertices), _lib.ptr(self.T_inv),
                                          self.vertices.shape[1], float(self.threshold), C.byref(net.field_desc(P)),
                                          float(fill), int(nan_to_num), _lib.ptr(rgb), _lib.ptr(sigma), _lib.ptr(ws),
                                          ws.numel(), _lib.stream()), "ia_smpl_deform_query")
        return rgb, sigma

    @staticmethod
    def _native(model):
        from ..models.networks.ngp import NeRFNGPNet
        return isinstance(model, NeRFNGPNet)

    def deform_test(self, pts, model):
        """smpl_deformer.py:122-131: invalid points -> rgb 0, sigma 0"""
        if self._native(model) and not torch.is_grad_enabled():
            return self._query_fused(pts, model, 0.0, 0)
        from .. import dense_routes
        return dense_routes.deform_query_single(self, pts, model, eval_mode=True)

    def deform_train(self, pts, model):
        """smpl_deformer.py:112-120: invalid or non-finite -> rgb 0, sigma -1e5"""
        from .. import dense_routes
        return dense_routes.deform_query_single(self, pts, model, eval_mode=False)

    def __call__(self, pts, model, eval_mode=True):
        return self.deform_test(pts, model) if eval_mode else self.deform_train(pts, model)
