"""Raymarcher plugin (drop-in for instant_avatar/renderers/raymarcher_acc.py:49).

    Raymarcher(MAX_SAMPLES, MAX_BATCH_SIZE, smpl_init=False)
    .initialize(N_frames)  .idx  .density_grid_test  .density_grid_train
    __call__(rays, model, eval_mode, noise, bg_color) -> dict(rgb_coarse, depth_coarse,
        alpha_coarse, counter_coarse | weight_coarse)

Two test-time routes with identical results:
  * fused: `ia_render_test` -- the whole wave-front loop of raymarcher_acc.py:83-138
    (march -> deformer query -> composite -> alive compaction, device-side N_step
    schedule) enqueued without a single host sync.  Taken when the model closure
    is recognised as (SNARFDeformer, NeRFNGPNet) or after `bind_fused(deformer, net)`;
  * dense (instantavatar_amd/dense_routes.py): any other `model(pts, _)` callable -- the unfused
    kernels `ia_raymarch_test` / `ia_composite_test` driven from the host with the callable in between.
"""
import ctypes as C

import torch

from .. import _lib
from ..models.structures.density_grid import DensityGrid


def _find_native_pair(model):
    """Recognise `lambda x, _: deformer(x, net, eval_mode)` (models/DNeRF.py:66-67)."""
    from ..deformers.snarf_deformer import SNARFDeformer
    from ..models.networks.ngp import NeRFNGPNet
    pair = getattr(model, "ia_native_pair", None)
    if pair is not None:
        return pair
    cells = getattr(model, "__closure__", None) or ()
    objs = []
    for c in cells:
        try:
            objs.append(c.cell_contents)
        except ValueError:
            pass
    for o in list(objs):
        objs.extend(v for v in (getattr(o, "deformer", None), getattr(o, "net_coarse", None)) if v is not None)
    from ..deformers.smpl_deformer import SMPLDeformer
    d = next((o for o in objs if isinstance(o, (SNARFDeformer, SMPLDeformer))), None)
    n = next((o for o in objs if isinstance(o, NeRFNGPNet)), None)
    return (d, n) if d is not None and n is not None else None


class _RaySamplesFn(torch.autograd.Function):
    """The compact sample points of a training render as a function of the rays: pts = o + z d (raymarcher_acc.py:158; the depths
    z come from the marcher and are not differentiated).  Forward: the points the march kernel already wrote; backward:
    `ia_ray_samples_bwd` (per-ray sums, one wave per ray).  Only the SMPLDeformer's fit stage needs it: there the ray frame w2s is
    under optimisation and the reference's autograd runs through transform_rays_w2s (smpl_deformer.py:79-86)."""

    @staticmethod
    def forward(ctx, o, d, st):
        ctx.st = st
        ctx.shapes = (o.shape, d.shape)
        # (written by the march kernel from the same o, d.  A NEW tensor object on the same memory: autograd attaches the node to the
        # object a Function returns -- returning the dict's own entry would make `st["s_pts"]` a non-leaf whose grad_fn owns `st`: a
        # reference cycle that keeps the step's whole graph, and the AccumulateGrad nodes of the SMPL tables, alive into the next
        # step -- and a HIP-graph capture on another stream then dies in capture_end)
        return st["s_pts"].detach()

    @staticmethod
    def backward(ctx, d_pts):
        st = ctx.st
        n = st["n"]
        dev = d_pts.device
        d_o, d_d = torch.empty((n, 3), device=dev), torch.empty((n, 3), device=dev)
        g = d_pts.float().contiguous()
        _lib.check(_lib.lib().ia_ray_samples_bwd(_lib.ptr(st["ray_off"]), _lib.ptr(st["ray_cnt"]), _lib.ptr(st["s_z"]), _lib.ptr(g), n,
                                                 _lib.ptr(d_o), _lib.ptr(d_d), _lib.stream()), "ia_ray_samples_bwd")
        return d_o.reshape(ctx.shapes[0]), d_d.reshape(ctx.shapes[1]), None


class _SmplDeformCompactFn(torch.autograd.Function):
    """SMPLDeformer.deform on compact samples (smpl_deformer.py:88-110) with the valid points compacted: `ia_smpl_nn_compact` /
    `ia_smpl_nn_compact_bwd`.  Inputs: sample points [cap,3] (gradient -> rays), T_inv [1,V,4,4] (gradient -> body model)."""

    @staticmethod
    def forward(ctx, pts, T_inv, deformer, n_pts_dev, out):
        L = _lib.lib()
        x = pts.detach().reshape(-1, 3).float().contiguous()
        P = x.shape[0]
        V = deformer.vertices.shape[1]
        Ti = T_inv.detach().reshape(V, 4, 4).float().contiguous()
        dev = x.device
        cand_xc = torch.empty((P, 3), device=dev)
        cand_pt, idx = torch.empty(P, dtype=torch.int32, device=dev), torch.empty(P, dtype=torch.int32, device=dev)
        pt_off, pt_cnt = torch.empty(P, dtype=torch.int32, device=dev), torch.empty(P, dtype=torch.uint8, device=dev)
        n_cand = torch.empty(1, dtype=torch.int32, device=dev)
        _lib.check(L.ia_smpl_nn_compact(_lib.ptr(x), P, _lib.ptr(n_pts_dev), _lib.ptr(deformer.vertices.detach()), _lib.ptr(Ti), V,
                                        float(deformer.threshold), _lib.ptr(cand_xc), _lib.ptr(cand_pt), _lib.ptr(idx), _lib.ptr(pt_off),
                                        _lib.ptr(pt_cnt), _lib.ptr(n_cand), deformer.nn_grid_ptr(), _lib.stream()), "ia_smpl_nn_compact")
        out.update(pt_off=pt_off, pt_cnt=pt_cnt, n_cand=n_cand)
        ctx.save_for_backward(x, Ti, cand_pt, idx, n_cand)
        ctx.shapes = (pts.shape, T_inv.shape)
        ctx.need = (pts.requires_grad, T_inv.requires_grad)
        return cand_xc

    @staticmethod
    def backward(ctx, d_cand):
        x, Ti, cand_pt, idx, n_cand = ctx.saved_tensors
        P, V = x.shape[0], Ti.shape[0]
        dev = x.device
        need_p, need_T = ctx.need
        d_pts = torch.empty((P, 3), device=dev) if need_p else None
        d_T = torch.empty((V, 4, 4), device=dev) if need_T else None
        g = d_cand.float().contiguous()
        _lib.check(_lib.lib().ia_smpl_nn_compact_bwd(_lib.ptr(x), P, _lib.ptr(cand_pt), _lib.ptr(idx), _lib.ptr(n_cand), P, _lib.ptr(Ti), V,
                                                     _lib.ptr(g), _lib.ptr(d_T), _lib.ptr(d_pts), _lib.stream()), "ia_smpl_nn_compact_bwd")
        return (d_pts.reshape(ctx.shapes[0]) if need_p else None), (d_T.reshape(ctx.shapes[1]) if need_T else None), None, None, None


class _CompositeTrainFn(torch.autograd.Function):
    """composite() + render_train tail (raymarcher_acc.py:25-36,161-186) on compact samples:
    `ia_composite_train_fwd` / `ia_composite_train_bwd`."""

    @staticmethod
    def forward(ctx, cand_rgb, cand_sigma, st):
        L = _lib.lib()
        dev = cand_rgb.device
        n, S = st["n"], st["S"]
        cand_rgb, cand_sigma = cand_rgb.contiguous(), cand_sigma.contiguous()
        cap = st["s_z"].shape[0]
        from ..training import pooled_zeros
        color, depth, alpha = torch.empty((n, 3), device=dev), torch.empty(n, device=dev), torch.empty(n, device=dev)
        weights = pooled_zeros((n, S), dev)  # the kernel writes the occupied slots only
        # the zero-initialised gradient buffers of the backward pass, taken now from the step's zero pool (one fill launch for all)
        ctx.d_bufs = (pooled_zeros((cand_sigma.shape[0], 3), dev), pooled_zeros((cand_sigma.shape[0],), dev))
        ctx.set_materialize_grads(False)   # an unused output (depth) arrives as None, not as a freshly filled zero tensor
        sv = dict(arg=torch.empty(cap, dtype=torch.int32, device=dev), sigma=torch.empty(cap, device=dev),
                  alpha=torch.empty(cap, device=dev), T=torch.empty(cap, device=dev))
        _lib.check(L.ia_composite_train_fwd(_lib.ptr(cand_rgb), _lib.ptr(cand_sigma), cand_sigma.shape[0],
                                            _lib.ptr(st["pt_off"]), _lib.ptr(st["pt_cnt"]),
                                            st["n_init"], _lib.ptr(st["ray_off"]), _lib.ptr(st["ray_cnt"]), _lib.ptr(st["s_z"]),
                                            _lib.ptr(st["near"]), _lib.ptr(st["far"]), n, S, _lib.ptr(st["noise"]), st["noise_scale"],
                                            _lib.ptr(st["bg"]), _lib.ptr(color), _lib.ptr(depth), _lib.ptr(alpha), _lib.ptr(weights),
                                            _lib.ptr(st["s_slot"]), _lib.ptr(sv["arg"]), _lib.ptr(sv["sigma"]), _lib.ptr(sv["alpha"]),
                                            _lib.ptr(sv["T"]), _lib.stream()), "ia_composite_train_fwd")
        ctx.st, ctx.sv = st, sv
        ctx.save_for_backward(cand_rgb)
        ctx.n_cand = cand_sigma.shape[0]
        return color, depth, alpha, weights

    @staticmethod
    def backward(ctx, d_color, d_depth, d_alpha, d_weights):
        L = _lib.lib()
        st, sv = ctx.st, ctx.sv
        (cand_rgb,) = ctx.saved_tensors
        dev = cand_rgb.device
        c = lambda t: None if t is None else t.float().contiguous()
        d_color, d_depth, d_alpha, d_weights = c(d_color), c(d_depth), c(d_alpha), c(d_weights)
        bufs, ctx.d_bufs = getattr(ctx, "d_bufs", None), None
        if bufs is None:       # (a second backward through the same graph: fresh buffers)
            bufs = (torch.zeros((ctx.n_cand, 3), device=dev), torch.zeros(ctx.n_cand, device=dev))
        d_rgb, d_sig = bufs
        _lib.check(L.ia_composite_train_bwd(_lib.ptr(d_color), _lib.ptr(d_depth), _lib.ptr(d_alpha), _lib.ptr(d_weights),
                                            _lib.ptr(cand_rgb), _lib.ptr(st["ray_off"]), _lib.ptr(st["ray_cnt"]), _lib.ptr(st["s_z"]),
                                            _lib.ptr(st["near"]), _lib.ptr(st["far"]), st["n"], st["S"], _lib.ptr(st["bg"]),
                                            _lib.ptr(st["s_slot"]), _lib.ptr(sv["arg"]), _lib.ptr(sv["sigma"]), _lib.ptr(sv["alpha"]),
                                            _lib.ptr(sv["T"]), _lib.ptr(d_rgb), _lib.ptr(d_sig), _lib.stream()),
                   "ia_composite_train_bwd")
        return d_rgb, d_sig, None


class Raymarcher(torch.nn.Module):
    def __init__(self, MAX_SAMPLES: int, MAX_BATCH_SIZE: int, smpl_init: bool = False) -> None:
        super().__init__()
        self.MAX_SAMPLES = MAX_SAMPLES
        self.MAX_BATCH_SIZE = MAX_BATCH_SIZE
        self.register_buffer("aabb", torch.tensor([[-1.25, -1.55, -1.25], [1.25, 0.95, 1.25]]).float(),
                             persistent=False)
        self.density_grid_test = DensityGrid(64)
        self.smpl_init = smpl_init
        self.idx = 0
        self._fused = None
        self._ws = None
        self._iters_hint = 8     # loop iterations enqueued per call; adapted per frame
        self._n_alive_host = None
        self.last_iters = 0

    def initialize(self, N):
        """raymarcher_acc.py:66-70: one training grid (N with smpl_init: one per frame), held in a PLAIN list as in the
        reference -- not a sub-module: the training grids are not part of `state_dict()` (a checkpoint of the reference
        does not carry them, and its loaders are strict), they restart from zero on resume and are rebuilt by the updates."""
        n = N if self.smpl_init else 1
        dev = self.aabb.device
        self.density_grid_train_all = [DensityGrid(64, self.aabb, smpl_init=self.smpl_init).to(dev) for _ in range(n)]
        for g in self.density_grid_train_all:
            g.aabb = self.aabb

    def _apply(self, fn, *args, **kwargs):
        """`.to(device)` / `.cuda()` of the renderer also moves the training grids (they are not registered sub-modules)"""
        super()._apply(fn, *args, **kwargs)
        for g in self.__dict__.get("density_grid_train_all", []):
            g._apply(fn, *args, **kwargs)
            g.aabb = self.aabb
        return self

    def bind_fused(self, deformer, net):
        self._fused = (deformer, net)

    def __call__(self, rays, model, eval_mode=True, noise=0, bg_color=None):
        if eval_mode:
            return self.render_test(rays, model, bg_color)
        return self.render_train(rays, model, noise, bg_color)

    @property
    def density_grid_train(self):
        return self.density_grid_train_all[min(self.idx, len(self.density_grid_train_all) - 1)]

    def _occ_desc(self, grid):
        o = _lib.OccGrid()
        o.G = grid.grid_size
        a = grid.aabb_tensor().cpu().tolist()  # host read (closure path only)
        o.aabb_min[:] = a[:3]
        o.aabb_max[:] = a[3:]
        return o

    # ------------------------------------------------------------------ test
    @torch.no_grad()
    def render_test(self, rays, model, bg_color):
        pair = self._fused or _find_native_pair(model)
        from ..deformers.snarf_deformer import SNARFDeformer
        if pair is not None and rays.o.is_cuda and isinstance(pair[0], SNARFDeformer):
            return self.render_test_fused(rays, pair[0], pair[1], bg_color)
        from .. import dense_routes
        return dense_routes.render_test(self, rays, model, bg_color)     # any other callable: host-driven loop, dense blocks

    @torch.no_grad()
    def render_test_fused(self, rays, deformer, net, bg_color=None, sync=True):
        """raymarcher_acc.py:83-138 as `ia_render_test`.  The loop length is data
        dependent; `_iters_hint` iterations are enqueued (idle ones cost a few
        empty launches) and the device-side alive count is checked once at the
        end -- if rays are still alive the call is resumed.  Results are
        independent of the hint."""
        L = _lib.lib()
        dev = rays.o.device
        o = rays.o.reshape(-1, 3).float().contiguous()
        d = rays.d.reshape(-1, 3).float().contiguous()
        near = rays.near.reshape(-1).float().contiguous()
        far = rays.far.reshape(-1).float().contiguous()
        R = o.shape[0]
        grid = self.density_grid_test
        k = len(deformer.deformer.init_bones)
        need = L.ia_render_workspace_bytes(R, self.MAX_BATCH_SIZE, k)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(int(need), dtype=torch.uint8, device=dev)
            self._n_alive_dev = torch.zeros(2, dtype=torch.int32, device=dev)  # [alive after the last iteration, iterations executed]
        rgb = torch.empty((R, 3), device=dev)
        depth, alpha, counter = torch.empty(R, device=dev), torch.empty(R, device=dev), torch.empty(R, device=dev)
        bg = bg_color.reshape(-1, 3).float().contiguous() if bg_color is not None else None
        tfs = deformer.tfs.detach().float().contiguous()
        aabb = grid.aabb_tensor()

        def launch(n_iters, resume):
            _lib.check(L.ia_render_test(_lib.ptr(o), _lib.ptr(d), _lib.ptr(near), _lib.ptr(far), R, _lib.ptr(bg),
                                        _lib.ptr(grid.occ_bits), grid.grid_size, _lib.ptr(aabb),
                                        _lib.ptr(deformer.deformer.voxel_J_cl), _lib.ptr(tfs), deformer.deformer._bones_c,
                                        k, C.byref(deformer.deformer.grid_desc()),
                                        C.byref(net.field_desc(self.MAX_BATCH_SIZE * k)),
                                        self.MAX_SAMPLES, self.MAX_BATCH_SIZE, n_iters, resume, _lib.ptr(rgb),
                                        _lib.ptr(depth), _lib.ptr(alpha), _lib.ptr(counter),
                                        _lib.ptr(self._n_alive_dev), _lib.ptr(self._ws), self._ws.numel(),
                                        _lib.stream()), "ia_render_test")

        total = self._iters_hint
        launch(total, 0)
        if sync and not getattr(self, "_graph_capture", False):
            # one device->host read per frame (4 bytes) to validate the hint
            while int(self._n_alive_dev[0].item()) > 0 and total < 2 * self.MAX_SAMPLES:
                launch(4, total)   # resume = iterations already enqueued for this frame
                total += 4
                self._iters_hint = total
        self.last_iters = total
        return {
            "rgb_coarse": rgb.reshape(rays.o.shape),
            "depth_coarse": depth.reshape(rays.near.shape),
            "alpha_coarse": alpha.reshape(rays.near.shape),
            "counter_coarse": counter.reshape(rays.near.shape),
        }

    # ----------------------------------------------------------------- train
    def render_train_fused_smpl(self, rays, deformer, net, noise, bg_color):
        """render_train (raymarcher_acc.py:140-186) with the SMPLDeformer plugin (fit stage) over COMPACT samples: march + jitter +
        compaction, nearest-vertex deformation + compaction of the valid samples (`ia_smpl_nn_compact`), the field under autograd
        on them, compositing forward / backward as two kernels.  With the SMPL parameters under optimisation the gradient reaches
        the per-vertex transforms (`ia_smpl_nn_compact_bwd` -> `ia_smpl_lbs_bwd`) and, through the sample points, the rays
        (`ia_ray_samples_bwd` -> transform_rays_w2s's autograd -> w2s).  No host synchronisation."""
        L = _lib.lib()
        dev = rays.o.device
        o = rays.o.reshape(-1, 3).float().contiguous()
        d = rays.d.reshape(-1, 3).float().contiguous()
        near = rays.near.detach().reshape(-1).float().contiguous()
        far = rays.far.detach().reshape(-1).float().contiguous()
        n, S = o.shape[0], self.MAX_SAMPLES
        cap = n * S
        grid = self.density_grid_train
        occ = self._occ_desc_cached(grid)
        i32 = lambda *s: torch.empty(s, dtype=torch.int32, device=dev)
        st = dict(s_pts=torch.empty((cap, 3), device=dev), s_z=torch.empty(cap, device=dev), s_slot=i32(cap),
                  ray_off=i32(n), ray_cnt=i32(n), n_samples=i32(1), near=near, far=far, n=n, S=S)
        draws = getattr(self, "train_draws", None) or {}
        jitter = draws["ray_jitter"].to(dev).float().reshape(n, S).contiguous() if "ray_jitter" in draws else torch.rand((n, S), device=dev)  # :156
        with torch.no_grad():
            _lib.check(L.ia_march_train_compact(_lib.ptr(o.detach()), _lib.ptr(d.detach()), _lib.ptr(near), _lib.ptr(far), n, _lib.ptr(grid.occ_bits),
                                                C.byref(occ), S, _lib.ptr(jitter), _lib.ptr(st["s_pts"]), _lib.ptr(st["s_z"]),
                                                _lib.ptr(st["s_slot"]), _lib.ptr(st["ray_off"]), _lib.ptr(st["ray_cnt"]),
                                                _lib.ptr(st["n_samples"]), cap, _lib.stream()), "ia_march_train_compact")
        pts = st["s_pts"]
        if torch.is_grad_enabled() and (o.requires_grad or d.requires_grad):
            pts = _RaySamplesFn.apply(o, d, st)
        T_inv = deformer.T_inv
        from ..training import ZeroPool, field_autograd
        if torch.is_grad_enabled():
            ZeroPool.current = ZeroPool(8 * cap + n * S + 1024, dev)
        if torch.is_grad_enabled() and (pts.requires_grad or T_inv.requires_grad):
            cand = _SmplDeformCompactFn.apply(pts, T_inv, deformer, st["n_samples"], st)
        else:
            with torch.no_grad():
                cand = _SmplDeformCompactFn.apply(pts, T_inv, deformer, st["n_samples"], st)
        st.update(n_init=1, bg=bg_color.reshape(-1, 3).float().contiguous() if bg_color is not None else None,
                  noise=(draws["noise"].to(dev).float().reshape(n, S).contiguous() if "noise" in draws else torch.randn((n, S), device=dev))
                  if noise > 0 else None, noise_scale=float(noise))                # :167
        rgb_c, sig_c = field_autograd(net, cand, n_dev=st["n_cand"])
        self.train_overflow_flag = None      # one candidate per sample at most: the sample capacity bounds the candidates
        self.train_overflow_src = None
        color, depth, alpha, weights = _CompositeTrainFn.apply(rgb_c.float(), sig_c.float(), st)
        return {
            "rgb_coarse": color.reshape(rays.o.shape),
            "depth_coarse": depth.reshape(rays.near.shape),
            "alpha_coarse": alpha.reshape(rays.near.shape),
            "weight_coarse": weights.reshape(*rays.near.shape, -1),
        }

    def render_train_fused(self, rays, deformer, net, noise, bg_color):
        """render_train (raymarcher_acc.py:140-186) over COMPACT samples: march + jitter +
        compaction, candidate search + compaction, field under autograd on the surviving
        candidates, compositing forward/backward as two kernels.  No host synchronisation: all
        counts stay on the device."""
        L = _lib.lib()
        dev = rays.o.device
        o = rays.o.reshape(-1, 3).float().contiguous()
        d = rays.d.reshape(-1, 3).float().contiguous()
        near = rays.near.reshape(-1).float().contiguous()
        far = rays.far.reshape(-1).float().contiguous()
        n, S = o.shape[0], self.MAX_SAMPLES
        cap = n * S
        grid = self.density_grid_train
        occ = self._occ_desc_cached(grid)
        i32 = lambda *s: torch.empty(s, dtype=torch.int32, device=dev)
        k = len(deformer.deformer.init_bones)
        cand_cap = min(cap * k, self.train_cand_capacity)
        from ..training import ZeroPool, field_autograd, pooled_zeros
        if torch.is_grad_enabled():
            # ONE zero-fill for the step's zero-initialised work tensors (closed by training_step): the two device counters,
            # field outputs [V,3] + [V], dense weights [n,S], the compositor's candidate gradients [V,3] + [V], the loss values
            ZeroPool.current = ZeroPool(8 * cand_cap + n * S + 2048, dev)
        # the step's two device-side counters as ONE int32 pair: [samples, candidates] -- zeroed with the pool, copied to the host
        # in one transfer (`_train_counts_post`)
        counts = pooled_zeros((2,), dev).view(torch.int32)
        st = dict(s_pts=torch.empty((cap, 3), device=dev), s_z=torch.empty(cap, device=dev), s_slot=i32(cap),
                  ray_off=i32(n), ray_cnt=i32(n), n_samples=counts[0:1], near=near, far=far, n=n, S=S)
        draws = getattr(self, "train_draws", None) or {}                           # injected by reproducible tests
        jitter = draws["ray_jitter"].to(dev).float().reshape(n, S).contiguous() if "ray_jitter" in draws else torch.rand((n, S), device=dev)  # :156
        want_J_inv = deformer.tfs.requires_grad and torch.is_grad_enabled() and deformer.deformer.version == 1
        with torch.no_grad():
            _lib.check(L.ia_march_train_compact(_lib.ptr(o), _lib.ptr(d), _lib.ptr(near), _lib.ptr(far), n, _lib.ptr(grid.occ_bits),
                                                C.byref(occ), S, _lib.ptr(jitter), _lib.ptr(st["s_pts"]), _lib.ptr(st["s_z"]),
                                                _lib.ptr(st["s_slot"]), _lib.ptr(st["ray_off"]), _lib.ptr(st["ray_cnt"]),
                                                _lib.ptr(st["n_samples"]), cap, _lib.stream()), "ia_march_train_compact")
            sc = deformer.search_compact(st["s_pts"], n_pts_dev=st["n_samples"], cap=cand_cap,
                                         want_J_inv=want_J_inv, n_cand_out=counts[1:2])
        # No host read: the field runs on a capacity-sized candidate buffer with the device-side
        # count (kernels clamp to it).  The counts of step i are copied to pinned memory and looked
        # at during step i+1: a step whose candidates exceeded the capacity (they were dropped) is
        # counted in `train_overflow` and the capacity grows for the following steps.
        self._train_counts_check()
        st.update(pt_off=sc["pt_off"], pt_cnt=sc["pt_cnt"], n_init=k,
                  bg=bg_color.reshape(-1, 3).float().contiguous() if bg_color is not None else None,
                  noise=(draws["noise"].to(dev).float().reshape(n, S).contiguous() if "noise" in draws else torch.randn((n, S), device=dev))
                  if noise > 0 else None, noise_scale=float(noise))                # :167
        # (SMPL refinement: the candidates carry the implicit-differentiation gradient to tfs, deformer_torch.py:50-67)
        rgb_c, sig_c = field_autograd(net, deformer.candidates_with_grad(sc), n_dev=sc["n_cand"])
        self._train_counts_post(counts, cand_cap)
        # device-side overflow flag of THIS step: candidates past the capacity were dropped (in atomic-arrival order), so the
        # step's gradients are wrong -- `training_step` feeds the flag to the optimiser's found_inf, the update is skipped on
        # the device without a host read; the deferred count check then grows the capacity and the next steps are whole
        self.train_overflow_flag = None
        self.train_overflow_src = (sc["n_cand"], int(cand_cap))     # -> `training_step` (the loss kernel compares; `overflow_flag()` for others)
        color, depth, alpha, weights = _CompositeTrainFn.apply(rgb_c.float(), sig_c.float(), st)
        return {
            "rgb_coarse": color.reshape(rays.o.shape),
            "depth_coarse": depth.reshape(rays.near.shape),
            "alpha_coarse": alpha.reshape(rays.near.shape),
            "weight_coarse": weights.reshape(*rays.near.shape, -1),
        }

    #: capacity (candidates) of the training field call; 2^20 x 480 B of activations = 0.5 GB
    train_cand_capacity = 1 << 20
    train_overflow = 0
    train_overflow_flag = None     # device scalar > 0: the last training render dropped candidates (set by callers / tests)
    train_overflow_src = None      # (device int32 candidate counter, capacity) of the last fused training render

    def overflow_flag(self):
        """The overflow flag of the last training render as a device float scalar (None: the route cannot overflow)."""
        if self.train_overflow_flag is not None:
            return self.train_overflow_flag
        if self.train_overflow_src is not None:
            cnt, cap = self.train_overflow_src
            return (cnt.reshape(-1)[0] > cap).to(torch.float32).reshape(())
        return None

    def _train_counts_post(self, counts, cand_cap):
        """counts: device int32 [2] = [samples, candidates] of this step -> pinned host pair, one asynchronous copy"""
        if not hasattr(self, "_tc_host"):
            self._tc_host = torch.zeros(2, dtype=torch.int32).pin_memory()
        self._tc_host.copy_(counts, non_blocking=True)
        if getattr(self, "_graph_capture", False):
            return   # recorded into a graph: the replaying caller marks the copy with `_train_counts_posted`
        self._train_counts_posted(cand_cap)

    def _train_counts_posted(self, cand_cap):
        self._tc_event = torch.cuda.Event()
        self._tc_event.record()
        self._tc_cap = cand_cap

    def _train_counts_peek(self, cand_cap):
        """Graph-replay variant of the deferred check: look at whatever counts the device has copied to the pinned
        pair so far -- those of a step one or two replays back -- without waiting for anything."""
        if not hasattr(self, "_tc_host"):
            return
        self.last_train_counts = (int(self._tc_host[0]), int(self._tc_host[1]))
        if self.last_train_counts[1] > cand_cap:
            self.train_overflow += 1
            self.train_cand_capacity = max(self.train_cand_capacity, 2 * self.last_train_counts[1])
            self._tc_host[1] = 0   # counted once

    def _train_counts_check(self):
        """Deferred look at the previous step's counts (no stall: that step has long finished)."""
        ev = getattr(self, "_tc_event", None)
        if ev is None or getattr(self, "_graph_capture", False):
            return
        ev.synchronize()
        self._tc_event = None
        self.last_train_counts = (int(self._tc_host[0]), int(self._tc_host[1]))
        if self.last_train_counts[1] > self._tc_cap:
            self.train_overflow += 1
            self.train_cand_capacity = max(self.train_cand_capacity, 2 * self.last_train_counts[1])

    def iters_executed(self):
        """Wave-front iterations of the last fused test render that had rays to process (host read)."""
        return int(self._n_alive_dev[1].item())

    def _occ_desc_cached(self, grid):
        key = id(grid.aabb)
        if getattr(self, "_occ_key", None) != key:
            self._occ_cache, self._occ_key = self._occ_desc(grid), key
        return self._occ_cache

    def render_train(self, rays, model, noise, bg_color):
        """raymarcher_acc.py:140-186."""
        pair = self._fused or _find_native_pair(model)
        if pair is not None and rays.o.is_cuda and pair[0].fused_train_route():
            from ..deformers.smpl_deformer import SMPLDeformer
            if isinstance(pair[0], SMPLDeformer):
                return self.render_train_fused_smpl(rays, pair[0], pair[1], noise, bg_color)
            return self.render_train_fused(rays, pair[0], pair[1], noise, bg_color)
        from .. import dense_routes
        return dense_routes.render_train(self, rays, model, noise, bg_color)   # any other callable / the dense deformer route
