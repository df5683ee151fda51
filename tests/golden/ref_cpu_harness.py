"""Harness that runs the REFERENCE's Python (instant_avatar.*) on the CPU of the build container.

The reference cannot run here as it is: its three JIT-compiled CUDA extensions, tiny-cuda-nn, pytorch3d, kaolin, hydra,
pytorch_lightning, cv2 and a GPU are all missing.  None of those is the code this harness is after.  What it exercises is
the reference's own Python -- SMPL / LBS, SNARFDeformer, ForwardDeformer, NeRFNGPNet, DensityGrid, Raymarcher,
DNeRFModel.{forward, render_image_fast, update_density_grid} -- with the native pieces replaced by adapters around the
CPU oracle (oracle/ia_oracle.c), which is itself pinned to the reference's CUDA kernels on the MI355X
(tests/test_ref_pin.py).  Goldens produced this way pin the oracle's restatement of the Python glue (N_step schedule,
alive-ray bookkeeping, masks and fills, occupancy post-processing, EMA / step-500 switch, regulariser, ...) to the
reference executing, not to a reading of it.

Test infrastructure only: imported by tests/golden/make_*.py, never by the product.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"


class Opt(dict):
    """omegaconf-like: attribute access + .get"""
    __getattr__ = dict.__getitem__


def _np(t):
    return t.detach().numpy() if torch.is_tensor(t) else np.asarray(t)


def torch_encoder(xn, enc, hd, n_levels=16):
    """fp32 restatement of tcnn's NetworkWithInputEncoding (HashGrid -> 32-64-16 MLP) on unit coordinates, differentiable
    w.r.t. xn and the flat parameter vector enc = [W1 64x2L | W2 16x64 | table]: what tcnn's backward computes (its
    input gradient is the derivative of the trilinear weights, floor() has none), without its fp16 rounding."""
    n1 = 64 * 2 * n_levels
    W1, W2, table = enc[:n1].view(64, 2 * n_levels), enc[n1:n1 + 1024].view(16, 64), enc[n1 + 1024:].view(-1, 2)
    feats = []
    for l in range(n_levels):
        scale, res = float(hd.scale[l]), int(hd.res[l])
        off, size = int(hd.offset[l]), int(hd.offset[l + 1] - hd.offset[l])
        pos = xn * scale + 0.5
        g = pos.detach().floor()
        w = pos - g
        g = g.long()
        acc = 0
        for idx in range(8):
            c = [g[:, d] + ((idx >> d) & 1) for d in range(3)]
            wt = 1
            for d in range(3):
                wt = wt * (w[:, d] if (idx >> d) & 1 else 1 - w[:, d])
            if res ** 3 <= size:
                index = (c[0] + c[1] * res + c[2] * res * res) % size
            else:
                index = ((c[0] * 1) ^ ((c[1] * 2654435761) & 0xffffffff) ^ ((c[2] * 805459861) & 0xffffffff)) % size
            acc = acc + wt[:, None] * table[off + index]
        feats.append(acc)
    h1 = torch.relu(torch.cat(feats, dim=1) @ W1.t())
    return h1 @ W2.t()


def torch_color(in15, col):
    """fp32 restatement of tcnn's colour Network (15 -> pad 16 with 1 -> 64 -> 64 -> 16, sigmoid, 3 used)."""
    Wc1, Wc2, Wc3 = col[:1024].view(64, 16), col[1024:5120].view(64, 64), col[5120:6144].view(16, 64)
    cin = torch.cat([in15, torch.ones_like(in15[:, :1])], dim=1)
    c2 = torch.relu(torch.relu(cin @ Wc1.t()) @ Wc2.t())
    return torch.sigmoid((c2 @ Wc3.t())[:, :3])


def install(oracle, field_holder, differentiable=False):
    """Install the stubs and return the imported reference modules.  field_holder: dict with key "field" (oracle Field
    struct) used by the tinycudann stand-in.  differentiable: the two tcnn stand-ins carry REAL flat parameter vectors
    (field_holder["fp"], the dict of synthetic.make_field) and a backward pass -- forward values still come from the
    oracle's C restatement (fp16 rounding points of tcnn), gradients from the fp32 formula above, which is what the
    training-step goldens (make_refine_golden.py) need."""
    sys.dont_write_bytecode = True
    C = oracle.C
    L = oracle.lib()
    P = oracle._p

    def deformer_init(offset_kernel, scale_kernel, D, H, W, lbs=None):
        d = dict(offset_kernel=np.ascontiguousarray(_np(offset_kernel).reshape(3), np.float32),
                 scale_kernel=np.ascontiguousarray(_np(scale_kernel).reshape(3), np.float32), D=D, H=H, W=W)
        if lbs is not None:
            d["lbs_voxel"] = np.ascontiguousarray(_np(lbs).reshape(24, D, H, W), np.float32)
        return d

    # ---- the three deformer extensions (deformer_torch.py:9-20) ----
    def precompute(lbs_voxel_final, tfs, voxel_d, voxel_J, offset_kernel, scale_kernel):
        _, _, D, H, W = lbs_voxel_final.shape
        vJ, vd = oracle.precompute(deformer_init(offset_kernel, scale_kernel, D, H, W, lbs_voxel_final), _np(tfs)[0])
        voxel_J[0].copy_(torch.as_tensor(vJ))
        voxel_d[0].copy_(torch.as_tensor(vd))

    def fuse_broyden(xc, xd, voxel, voxel_J, tfs, bones, align_corners, J_inv, is_valid, offset_kernel, scale_kernel, cvg, dvg):
        _, _, D, H, W = voxel_J.shape
        x, Ji, valid = oracle.broyden(_np(xd)[0], np.ascontiguousarray(_np(voxel_J)[0]), _np(tfs)[0],
                                      deformer_init(offset_kernel, scale_kernel, D, H, W), _np(bones).astype(np.int32), cvg, dvg)
        xc[0].copy_(torch.as_tensor(x))
        J_inv[0].copy_(torch.as_tensor(Ji))
        is_valid[0].copy_(torch.as_tensor(valid.astype(bool)))

    def filt(x, mask):
        keep = oracle.filter_dup(np.ascontiguousarray(_np(x)[0]), np.ascontiguousarray(_np(mask)[0].astype(np.uint8)))
        return torch.as_tensor(keep.astype(bool))[None]

    # ---- the ray-march extension (raymarcher_acc.py:13-16) ----
    def f32(t):
        assert t.dtype == torch.float32 and t.is_contiguous(), (t.dtype, t.is_contiguous())
        return t.detach().numpy()   # (rays carry a gradient to w2s when the SMPL parameters are optimised; the kernels have none)

    def raymarch_test(rays_o, rays_d, near, far, alive, density_field, scale, offset, step_size, N_step):
        Na = alive.shape[0]
        pts = torch.zeros((Na, N_step, 3))
        dn = torch.zeros((Na, N_step))
        zn = torch.zeros((Na, N_step))
        occ8 = np.ascontiguousarray(_np(density_field).astype(np.uint8))
        G = occ8.shape[0]
        al = np.ascontiguousarray(_np(alive), np.int64)
        o, d = np.ascontiguousarray(f32(rays_o.contiguous())), np.ascontiguousarray(f32(rays_d.contiguous()))
        L.orc_raymarch_test(P(o), P(d), P(f32(near)), P(np.ascontiguousarray(f32(far.contiguous()))), P(al), C.c_long(Na), P(occ8), G,
                            P(np.ascontiguousarray(f32(scale.contiguous()))), P(np.ascontiguousarray(f32(offset.contiguous()))),
                            P(f32(step_size)), N_step, P(pts.numpy()), P(dn.numpy()), P(zn.numpy()))   # writes near in place (raymarcher.cu:72)
        return pts, dn, zn

    def composite_test(rgb_vals, sigma_vals, d_new, z_new, alive, color, depth, no_hit, thresh):
        Na, Ns = sigma_vals.shape
        al = np.ascontiguousarray(_np(alive), np.int64)
        L.orc_composite_test(P(f32(rgb_vals)), P(f32(sigma_vals)), P(f32(d_new)), P(f32(z_new)), P(al), C.c_long(Na), Ns,
                             P(f32(color)), P(f32(depth)), P(f32(no_hit)), C.c_float(thresh))

    def raymarch_train(rays_o, rays_d, near, far, density_field, scale, offset, step_size, N_step):
        N = rays_o.shape[0]
        z = torch.zeros((N, N_step))
        occ8 = np.ascontiguousarray(_np(density_field).astype(np.uint8))
        c = lambda t: np.ascontiguousarray(f32(t.contiguous()))
        L.orc_raymarch_train(P(c(rays_o)), P(c(rays_d)), P(c(near)), P(c(far)), C.c_long(N), P(occ8), occ8.shape[0], P(c(scale)),
                             P(c(offset)), P(c(step_size)), N_step, P(z.numpy()))
        return z

    ext = {
        "fuse_cuda": types.SimpleNamespace(fuse_broyden=fuse_broyden),
        "filter": types.SimpleNamespace(filter=filt),
        "precompute": types.SimpleNamespace(precompute=precompute),
        "raymarch_kernel": types.SimpleNamespace(raymarch_test=raymarch_test, composite_test=composite_test, raymarch_train=raymarch_train),
    }
    import torch.utils.cpp_extension as cpp
    cpp.load = lambda name, **kw: ext[name]

    # ---- pytorch3d KNN (deformer_torch.py:227) through the oracle's KNN (pinned to the reference's knn_cpu.cpp) ----
    def knn_points(a, b, K=30):
        d, i = oracle.knn(_np(a)[0], _np(b)[0], K)
        return torch.as_tensor(d)[None], torch.as_tensor(i)[None], None
    p3d, ops = types.ModuleType("third_parties.pytorch3d"), types.ModuleType("third_parties.pytorch3d.ops")
    ops.knn_points = knn_points
    p3d.ops = ops
    sys.modules["third_parties.pytorch3d"], sys.modules["third_parties.pytorch3d.ops"] = p3d, ops

    # ---- tiny-cuda-nn: the two modules of ngp.py:26-57 ----
    class _Tcnn(torch.nn.Module):
        def __init__(self, n_params):
            super().__init__()
            self.params = torch.nn.Parameter(torch.zeros(n_params))

    class NetworkWithInputEncoding(_Tcnn):
        def __init__(self, n_input_dims, n_output_dims, encoding_config, network_config):
            super().__init__(1)
            assert (n_input_dims, n_output_dims, encoding_config["n_levels"]) == (3, 16, 16)

        def forward(self, x):
            return torch.as_tensor(oracle.tcnn_encoder(field_holder["field"], _np(x.float())))

    class Network(_Tcnn):
        def __init__(self, n_input_dims, n_output_dims, network_config):
            super().__init__(1)
            assert (n_input_dims, n_output_dims) == (15, 3)

        def forward(self, x):
            return torch.as_tensor(oracle.tcnn_color(field_holder["field"], _np(x.float())))
    if differentiable:
        fp = field_holder["fp"]
        flat = lambda *ks: torch.cat([torch.as_tensor(np.asarray(fp[k], np.float32).reshape(-1)) for k in ks])

        def current_field(enc, col):
            """oracle Field struct of the CURRENT parameter values (rebuilt after every optimiser step)"""
            key = (enc._version, col._version)
            if field_holder.get("key") != key:
                n1 = 64 * 2 * fp["n_levels"]
                e, c = enc.detach().numpy(), col.detach().numpy()
                d = dict(fp)
                d.update(sig_w1=e[:n1].astype(np.float16), sig_w2=e[n1:n1 + 1024].astype(np.float16),
                         table=e[n1 + 1024:].astype(np.float16), col_w1=c[:1024].astype(np.float16),
                         col_w2=c[1024:5120].astype(np.float16), col_w3=c[5120:6144].astype(np.float16))
                field_holder["field"], field_holder["keep"] = oracle.make_field(d)
                field_holder["key"] = key
            return field_holder["field"]

        class _EncFn(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, enc):
                ctx.save_for_backward(x, enc)
                return torch.as_tensor(oracle.tcnn_encoder(current_field(enc, field_holder["col"]), _np(x.float())))

            @staticmethod
            def backward(ctx, g):
                x, enc = ctx.saved_tensors
                with torch.enable_grad():
                    x2, e2 = x.detach().requires_grad_(True), enc.detach().requires_grad_(True)
                    out = torch_encoder(x2, e2, field_holder["field"].hash, fp["n_levels"])
                    gx, ge = torch.autograd.grad(out, [x2, e2], g)
                return gx, ge

        class _ColFn(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, col):
                ctx.save_for_backward(x, col)
                return torch.as_tensor(oracle.tcnn_color(current_field(field_holder["enc"], col), _np(x.float())))

            @staticmethod
            def backward(ctx, g):
                x, col = ctx.saved_tensors
                with torch.enable_grad():
                    x2, c2 = x.detach().requires_grad_(True), col.detach().requires_grad_(True)
                    gx, gc = torch.autograd.grad(torch_color(x2, c2), [x2, c2], g)
                return gx, gc

        class NetworkWithInputEncoding(torch.nn.Module):   # noqa: F811
            def __init__(self, n_input_dims, n_output_dims, encoding_config, network_config):
                super().__init__()
                assert (n_input_dims, n_output_dims, encoding_config["n_levels"]) == (3, 16, 16)
                self.params = torch.nn.Parameter(flat("sig_w1", "sig_w2", "table"))
                field_holder["enc"] = self.params

            def forward(self, x):
                return _EncFn.apply(x, self.params)

        class Network(torch.nn.Module):                    # noqa: F811
            def __init__(self, n_input_dims, n_output_dims, network_config):
                super().__init__()
                assert (n_input_dims, n_output_dims) == (15, 3)
                self.params = torch.nn.Parameter(flat("col_w1", "col_w2", "col_w3"))
                field_holder["col"] = self.params

            def forward(self, x):
                return _ColFn.apply(x, self.params)

    tcnn = types.ModuleType("tinycudann")
    tcnn.NetworkWithInputEncoding, tcnn.Network = NetworkWithInputEncoding, Network
    sys.modules["tinycudann"] = tcnn

    # ---- hydra / lightning / kaolin / cv2: imported by the modules, not used on this path ----
    hydra = types.ModuleType("hydra")
    hydra.utils = types.SimpleNamespace(to_absolute_path=lambda p: p, instantiate=None)
    sys.modules["hydra"] = hydra
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(torch.nn.Module):
        global_step = 0
    pl.LightningModule = LightningModule
    sys.modules["pytorch_lightning"] = pl
    kaolin = types.ModuleType("kaolin")
    kaolin.ops = types.ModuleType("kaolin.ops")
    kaolin.ops.mesh = types.ModuleType("kaolin.ops.mesh")
    kaolin.ops.mesh.index_vertices_by_faces = None
    sys.modules["kaolin"], sys.modules["kaolin.ops"], sys.modules["kaolin.ops.mesh"] = kaolin, kaolin.ops, kaolin.ops.mesh
    sys.modules["cv2"] = types.ModuleType("cv2")

    # ---- no GPU: .cuda() is the identity ----
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self

    if REF not in sys.path:
        sys.path.insert(0, REF)
    import instant_avatar.deformers.snarf_deformer as snarf
    import instant_avatar.models.networks.ngp as ngp
    import instant_avatar.renderers.raymarcher_acc as ray
    import instant_avatar.models.structures.density_grid as dgrid
    # DNeRF.py:15 opens logging.FileHandler("DNeRF.log") in the working directory at import time: import it from a
    # temporary directory so that the harness leaves nothing behind in the repository
    import tempfile
    _cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as _tmp:
        os.chdir(_tmp)
        try:
            import instant_avatar.models.DNeRF as dnerf
        finally:
            os.chdir(_cwd)
    from instant_avatar.deformers.smplx.body_models import SMPL
    from instant_avatar.deformers.smplx.utils import Struct
    return types.SimpleNamespace(snarf=snarf, ngp=ngp, ray=ray, dgrid=dgrid, dnerf=dnerf, SMPL=SMPL, Struct=Struct)


class SeededDraws:
    """torch.rand_like / torch.randn_like replaced by draws from a numpy RandomState, so that the oracle can be handed
    the very same numbers: a test re-creates them from the seed in the same order."""

    def __init__(self):
        self.rs = None
        self._rand, self._randn = torch.rand_like, torch.randn_like

    def seed(self, s):
        self.rs = np.random.RandomState(s)

    def __enter__(self):
        torch.rand_like = lambda t, **kw: torch.as_tensor(self.rs.rand(*t.shape).astype(np.float32))
        torch.randn_like = lambda t, **kw: torch.as_tensor(self.rs.randn(*t.shape).astype(np.float32))
        return self

    def __exit__(self, *a):
        torch.rand_like, torch.randn_like = self._rand, self._randn


def build_reference_model(R, body, fp, resolution=32, max_batch=291600):
    """deformer / network / renderer / DNeRFModel instances of the reference, on the synthetic body and field."""
    V = body["v_template"].shape[0]
    kintree = np.stack([np.asarray(body["parents"], np.int64), np.arange(24)])
    kintree[0, 0] = 2 ** 32 - 1
    struct = R.Struct(v_template=np.asarray(body["v_template"], np.float32), shapedirs=np.asarray(body["shapedirs"], np.float32),
                      posedirs=np.ascontiguousarray(np.asarray(body["posedirs"], np.float32).T.reshape(V, 3, -1)),
                      J_regressor=np.asarray(body["J_regressor"], np.float32), weights=np.asarray(body["lbs_weights"], np.float32),
                      kintree_table=kintree, f=np.array([[0, 1, 2]], np.int64))
    R.snarf.SMPL = lambda model_path, gender: R.SMPL(model_path, data_struct=struct, gender=gender)
    deformer = R.snarf.SNARFDeformer("", "neutral", Opt(cano_pose="a_pose", resolution=resolution, version=1))
    net = R.ngp.NeRFNGPNet(Opt(center=[float(v) for v in fp["center"]], scale=[float(v) for v in fp["scale"]]))
    renderer = R.ray.Raymarcher(MAX_SAMPLES=256, MAX_BATCH_SIZE=max_batch)
    renderer.initialize(1)
    renderer.idx = 0
    model = R.dnerf.DNeRFModel.__new__(R.dnerf.DNeRFModel)
    torch.nn.Module.__init__(model)
    model.net_coarse, model.deformer, model.renderer = net, deformer, renderer
    model.opt = Opt(optimize_SMPL=Opt(enable=False, is_refine=False))
    return model
