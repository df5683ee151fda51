"""DensityGrid plugin (drop-in for instant_avatar/models/structures/density_grid.py:17).

64^3 boolean occupancy grid.  `initialize` (per rendered frame, density_grid.py:95-110)
is ONE C-ABI call (`ia_density_grid_init`: 5 jittered probe sets -> fused deformer
query -> running max -> 1-exp(-0.01 s) -> 3^3 max-pool -> relative threshold ->
largest 26-connected component by union-find) when deformer/net are the native
plugins, and the same sequence through the `deformer(pts, net)` closure +
`ia_occupancy_from_density` otherwise.  `update` is the training-time EMA version
(density_grid.py:46-92).

Besides the reference's `density_field` (bool [G,G,G]) the grid keeps `occ_bits`,
the bit-packed copy (32 KB) the marcher kernels read, followed by one flag word
(1 = no border cell occupied, which lets the marcher reject samples outside the aabb
without a lookup).
"""
import ctypes as C

import torch
import torch.nn.functional as F

from ... import _lib


def get_aabb(vs, scale=1.2):
    lo, hi = vs.min(dim=0).values, vs.max(dim=0).values
    c, s = (lo + hi) * 0.5, (hi - lo) * 0.5
    return torch.stack([c - s * scale, c + s * scale], dim=0)


def denormalize(coords, aabb):
    return coords * (aabb[1] - aabb[0]) + aabb[0]


class DensityGrid(torch.nn.Module):
    def __init__(self, grid_size=64, aabb=None, smpl_init=False) -> None:
        super().__init__()
        self.grid_size = grid_size
        G = grid_size
        self.register_buffer("density_cached", torch.zeros(G, G, G))
        self.register_buffer("density_field", torch.zeros(G, G, G, dtype=torch.bool))
        self.register_buffer("occ_bits", torch.zeros(G * G * G // 32 + 8, dtype=torch.int32), persistent=False)   # + flag word + occupied-cell bounds (ia_occupancy_*)
        self.aabb = aabb
        self.initialized = False
        self.smpl_init = smpl_init
        self._coords = None
        self._ws = None
        self._bits_version = None

    # -- helpers ---------------------------------------------------------------
    @property
    def coords(self):
        """cell corners in [0,1)^3, [G,G,G,3] (density_grid.py:22-26)."""
        dev = self.density_cached.device
        if self._coords is None or self._coords.device != dev:
            idx = torch.arange(0, self.grid_size, device=dev)
            self._coords = torch.stack(torch.meshgrid((idx, idx, idx), indexing="ij"), dim=-1) / self.grid_size
        return self._coords

    @property
    def min_corner(self):
        return self.aabb[0]

    @property
    def max_corner(self):
        return self.aabb[1]

    def aabb_tensor(self):
        """contiguous [6] device tensor (min xyz, max xyz) for the kernels."""
        a = self.aabb
        if torch.is_tensor(a) and a.dim() == 2:
            return a.reshape(6).float().contiguous()
        return torch.cat([a[0].reshape(3), a[1].reshape(3)]).float().contiguous()

    def _workspace(self, nbytes):
        dev = self.density_cached.device
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != dev:
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        return self._ws

    def pack_bits(self):
        """refresh occ_bits from density_field (after loading a checkpoint / update)."""
        f8 = self.density_field.to(torch.uint8).contiguous()
        _lib.check(_lib.lib().ia_occupancy_pack(_lib.ptr(f8), self.grid_size, _lib.ptr(self.occ_bits), _lib.stream()),
                   "ia_occupancy_pack")

    def _postprocess(self, density):
        """density [G,G,G] -> density_field + occ_bits (density_grid.py:104-110)."""
        G = self.grid_size
        L = _lib.lib()
        ws = self._workspace(L.ia_occupancy_workspace_bytes(G))
        out8 = torch.empty((G, G, G), dtype=torch.bool, device=density.device)   # written as bytes 0 / 1 by the kernel
        _lib.check(L.ia_occupancy_from_density(_lib.ptr(density.contiguous()), G, _lib.ptr(self.occ_bits), _lib.ptr(out8),
                                               _lib.ptr(ws), ws.numel(), _lib.stream()), "ia_occupancy_from_density")
        self.density_field = out8

    # -- test-time grid (per frame) ----------------------------------------------
    batched_probes = True

    @torch.no_grad()
    def initialize(self, deformer, net, iters=5, jitter=None):
        """density_grid.py:95-110.  `jitter` ([iters,G^3,3] in [0,1)) may be injected
        for reproducible tests; by default it is drawn like the reference does
        (torch.rand_like, density_grid.py:100)."""
        G = self.grid_size
        bb = deformer.get_bbox_deformed()
        if (bb[0].is_contiguous() and bb[1].is_contiguous() and bb[0].numel() == 3 and bb[1].dtype == bb[0].dtype
                and bb[1].data_ptr() == bb[0].data_ptr() + 3 * bb[0].element_size()
                and bb[0].untyped_storage().data_ptr() == bb[1].untyped_storage().data_ptr()):
            # the two corners are the halves of ONE 6-float record (the voxel pass of the frame writes it: ia_precompute): `aabb` is a
            # [2,3] VIEW of that record -- no stack launch per frame.  It follows the deformer's prepared frame, as this grid does
            # (it is rebuilt by every render_image_fast); clone it to keep the box of an earlier frame.
            self.aabb = torch.as_strided(bb[0], (2, 3), (3, 1))
        else:
            self.aabb = torch.stack([bb[0], bb[1]])
        dev = self.density_cached.device
        if jitter is None:
            jitter = torch.rand((iters, G * G * G, 3), device=dev)
        jitter = jitter.to(dev).float().reshape(-1, G * G * G, 3).contiguous()
        iters = jitter.shape[0]  # an injected jitter tensor defines the number of probe sets
        from ...deformers.snarf_deformer import SNARFDeformer
        from ..networks.ngp import NeRFNGPNet
        if isinstance(deformer, SNARFDeformer) and isinstance(net, NeRFNGPNet):
            L = _lib.lib()
            k = len(deformer.deformer.init_bones)
            # all probe sets in one launch when the sizes allow (288 GB of HBM: ~1 GB of scratch); `batched_probes = False`
            # hands over the small workspace of `ia_density_init_workspace_bytes` only: one probe set per launch, same result
            need = L.ia_density_init_workspace_bytes(G, k)
            if self.batched_probes:
                need = max(need, L.ia_density_init_workspace_bytes_batched(G, k, iters))
                ws = self._workspace(need)
            else:
                ws = torch.empty(int(need), dtype=torch.uint8, device=dev)
            density = torch.empty((G, G, G), device=dev)
            out8 = torch.empty((G, G, G), dtype=torch.bool, device=dev)   # the kernel writes the bytes 0 / 1: a bool tensor's storage
            tfs = deformer.tfs.detach().float().contiguous()
            _lib.check(L.ia_density_grid_init(_lib.ptr(jitter), iters, G, _lib.ptr(self.aabb_tensor()),
                                              _lib.ptr(deformer.deformer.voxel_J_cl), _lib.ptr(tfs),
                                              deformer.deformer._bones_c, k, C.byref(deformer.deformer.grid_desc()),
                                              C.byref(net.field_desc(G * G * G * k * iters)), _lib.ptr(density), _lib.ptr(self.occ_bits),
                                              _lib.ptr(out8), _lib.ptr(ws), ws.numel(), _lib.stream()),
                       "ia_density_grid_init")
            self.density_field = out8
            self.density_probe = density
            return
        density = torch.zeros_like(self.coords[..., 0])
        for it in range(iters):
            pts = denormalize(self.coords + jitter[it].reshape(G, G, G, 3) / G, self.aabb)
            _, d = deformer(pts.reshape(-1, 3), net)
            density = torch.maximum(density, d.reshape(density.shape))
        self.density_probe = density
        self._postprocess(density)

    def mesh_signed_distance(self, vertices, faces):
        """signed distance [G,G,G] of the cell centres (coords + 0.5 / G, density_grid.py:55) to the triangle mesh
        (`ia_mesh_signed_distance`; the reference calls kaolin's point_to_mesh_distance and check_sign, :62-70)."""
        G = self.grid_size
        dev = self.density_cached.device
        v = vertices.detach().reshape(-1, 3).float().contiguous()
        f = faces.reshape(-1, 3).to(device=dev, dtype=torch.int32).contiguous()
        if f.numel() == 0:
            raise ValueError("smpl_init needs the body model's triangle faces (body_model.faces_tensor is empty)")
        _lib.require_cuda(v)
        L = _lib.lib()
        pts = torch.empty((G * G * G, 3), device=dev)
        sd = torch.empty(G * G * G, device=dev)
        _lib.check(L.ia_grid_cell_centres(G, _lib.ptr(self.aabb_tensor()), _lib.ptr(pts), _lib.stream()), "ia_grid_cell_centres")
        _lib.check(L.ia_mesh_signed_distance(_lib.ptr(pts), G * G * G, _lib.ptr(v), _lib.ptr(f), f.shape[0], _lib.ptr(sd), _lib.stream()),
                   "ia_mesh_signed_distance")
        return sd.reshape(G, G, G)

    # -- training-time grid ---------------------------------------------------------
    def update(self, deformer, net, step, reduce_hook=None, jitter=None, differentiable=True):
        """density_grid.py:46-92.  `reduce_hook(density_cached)` (optional) runs between the EMA update
        and the thresholding: data-parallel training MAX-reduces the cache there, so `density_field`,
        the returned `valid` mask and the occupancy bits all come from the reduced cache, once.
        `jitter` ([G,G,G,3] in [0,1)) may be injected for reproducible tests (reference: torch.rand_like).
        `differentiable=False`: the probe is not recorded for autograd (callers that drop the regulariser)."""
        G = self.grid_size
        if jitter is None:
            jitter = torch.rand_like(self.coords)
        coords = denormalize(self.coords + jitter.reshape(self.coords.shape).to(self.coords) / G, self.aabb)
        with (torch.enable_grad() if differentiable else torch.no_grad()):
            _, density = deformer(coords.reshape(-1, 3), net, eval_mode=False)
        density = density.clip(min=0).reshape(coords.shape[:-1])
        old = self.density_field
        if step < 500 and self.smpl_init:
            # density_grid.py:53-75: for the first 500 steps the grid is the posed SMPL mesh (+1 cm), set once
            if not self.initialized:
                sd = self.mesh_signed_distance(deformer.vertices, deformer.body_model.faces_tensor)
                self.density_field = sd < 0.01
                opacity = -torch.log(1 - self.density_field.float()) * 100
                self.density_cached = torch.maximum(self.density_cached * 0.8, opacity)
                self.pack_bits()
                self.initialized = True
        else:
            self.density_cached = torch.maximum(self.density_cached * 0.8, density.detach())
            if reduce_hook is not None:
                reduce_hook(self.density_cached)
            self._postprocess(self.density_cached)
        density = 1 - torch.exp(0.01 * -F.relu(density))
        valid = self.density_field if step < 500 else old
        return density, valid
