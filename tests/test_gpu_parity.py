"""GPU parity suite (-m gpu): every HIP kernel, called through the C ABI, against
the CPU oracle on identical seeded inputs.

Tolerances (north_star: RGB/alpha within 1e-3 abs, fp32):
  * integer / index / byte results (validity masks, occupancy bits, hash-grid
    features in fp16, ray-march sample depths)             -> bit exact
  * fp32 chains whose only difference is FMA contraction    -> 1e-5 .. 1e-4
  * rendered rgb / alpha                                    -> 1e-3
"""
import ctypes as C

import numpy as np
import pytest
import torch

from instantavatar_amd import _lib, synthetic as syn
from instantavatar_amd.models.structures.utils import Rays
from instantavatar_amd.pipeline import make_batch

import world as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def gpu_world(oracle):
    model, body, fp, init = W.build(DEV, 64, 16)
    poses, tr = W.poses()
    return model, body, fp, init, poses, tr


def _prepare(model, poses, tr, i, res=64):
    batch = make_batch(DEV, res, poses[i], tr[i])
    model.deformer.prepare_deformer(batch)
    return batch


def test_native_library_is_loaded():
    import os
    maps = open("/proc/self/maps").read()
    _lib.lib()
    maps = open("/proc/self/maps").read()
    assert "libinstantavatar_hip.so" in maps
    assert torch.cuda.is_available() and "gfx950" in torch.cuda.get_device_properties(0).gcnArchName


def test_smpl_tfs_kernel(oracle, gpu_world):
    model, body, fp, init, poses, tr = gpu_world
    for i in (0, 3, 5):
        _prepare(model, poses, tr, i)
        tfs, w2s = oracle.prepare_deformer(body, init, np.zeros(10, np.float32), poses[i, 3:], poses[i, :3], tr[i])
        assert np.abs(model.deformer.tfs[0].cpu().numpy() - tfs).max() < 2e-5
        assert np.abs(model.deformer.w2s[0].cpu().numpy() - w2s).max() < 2e-5


def test_precompute_kernel(oracle, gpu_world):
    model, body, fp, init, poses, tr = gpu_world
    _prepare(model, poses, tr, 2)
    tfs = model.deformer.tfs[0].cpu().numpy()
    vJ, vd = oracle.precompute(init, tfs)
    fd = model.deformer.deformer
    assert np.abs(fd.voxel_J[0].cpu().numpy() - vJ).max() < 1e-5
    assert np.abs(fd.voxel_d[0].cpu().numpy() - vd).max() < 1e-5
    bb = torch.cat(model.deformer.get_bbox_deformed()).cpu().numpy()
    vdg = fd.voxel_d[0].reshape(3, -1)
    assert np.array_equal(bb[:3], vdg.min(1).values.cpu().numpy()) and np.array_equal(bb[3:], vdg.max(1).values.cpu().numpy())


def test_precompute_bbox_routes_agree(gpu_world):
    """ia_precompute (float atomics on bbox) and ia_precompute_ws (per-workgroup extrema + k_bbox_reduce) must
    produce the same six numbers and the same J / voxel_d, bit for bit (min / max are order independent)."""
    import ctypes as C
    from instantavatar_amd import _lib
    model, body, fp, init, poses, tr = gpu_world
    _prepare(model, poses, tr, 3)
    fd = model.deformer.deformer
    tfs = model.deformer.tfs.detach().float().contiguous()
    L = _lib.lib()
    J1, d1, b1 = torch.empty_like(fd.voxel_J_cl), torch.empty_like(fd.voxel_d), torch.zeros(6, device=DEV)
    _lib.check(L.ia_precompute(_lib.ptr(fd.lbs_voxel_final), _lib.ptr(tfs), _lib.ptr(J1), _lib.ptr(d1), _lib.ptr(b1),
                               C.byref(fd.grid_desc()), _lib.stream()), "ia_precompute")
    assert torch.equal(J1, fd.voxel_J_cl) and torch.equal(d1, fd.voxel_d)
    assert torch.equal(b1, fd.bbox_deformed), (b1, fd.bbox_deformed)
    nb = int(L.ia_precompute_workspace_bytes(C.byref(fd.grid_desc())))
    assert nb > 0
    small = torch.empty(nb - 4, dtype=torch.uint8, device=DEV)
    rc = L.ia_precompute_ws(_lib.ptr(fd.lbs_voxel_final), _lib.ptr(tfs), _lib.ptr(J1), _lib.ptr(d1), _lib.ptr(b1),
                            C.byref(fd.grid_desc()), _lib.ptr(small), small.numel(), _lib.stream())
    assert rc != 0 and b"workspace" in L.ia_last_error()


def _query_points(model, n, seed):
    rng = np.random.RandomState(seed)
    vd = model.deformer.deformer.voxel_d[0].reshape(3, -1).cpu().numpy()
    pts = vd[:, rng.randint(0, vd.shape[1], n)].T + rng.randn(n, 3).astype(np.float32) * 0.02
    return np.ascontiguousarray(pts, np.float32)


def test_broyden_search_and_filter(oracle, gpu_world):
    model, body, fp, init, poses, tr = gpu_world
    _prepare(model, poses, tr, 4)
    fd = model.deformer.deformer
    tfs = model.deformer.tfs[0].cpu().numpy()
    vJ = np.ascontiguousarray(fd.voxel_J[0].cpu().numpy())
    pts = _query_points(model, 20011, 7)  # ragged: not a multiple of 64
    x_o, Ji_o, raw_o = oracle.broyden(pts, vJ, tfs, init, syn.INIT_BONES)
    keep_o = oracle.filter_dup(x_o, raw_o)
    out = fd.broyden_cuda(torch.as_tensor(pts, device=DEV)[None], None, fd.voxel_J_cl, model.deformer.tfs)
    x_g, keep_g = out["result"][0].cpu().numpy(), out["valid_ids"][0].cpu().numpy()
    Ji_g = out["J_inv"][0].cpu().numpy()
    # convergence flags are branchy: allow a vanishing fraction of flips (FMA contraction)
    flips = (keep_g != keep_o.astype(bool)).mean()
    assert flips < 2e-4, flips
    both = keep_g & keep_o.astype(bool)
    assert both.sum() > 1000
    assert np.abs(x_g[both] - x_o[both]).max() < 1e-4  # both within cvg=1e-5 of the root; one may stop an iteration earlier
    # J_inv is the matrix before the last rank-1 update (Q4): it moves by O(1) when the two
    # sides stop one iteration apart, so compare the bulk, not the max
    assert (np.abs(Ji_g[both] - Ji_o[both]).max(axis=(1, 2)) > 5e-3).mean() < 1e-3
    # Q1: non-converged / invalid slots stay exactly zero
    raw_g = np.abs(x_g).sum(-1) > 0
    assert (x_g[~raw_g] == 0).all()
    # empty input is fine
    e = fd.broyden_cuda(torch.zeros((1, 0, 3), device=DEV), None, fd.voxel_J_cl, model.deformer.tfs)
    assert e["result"].shape == (1, 0, 13, 3)


def test_hashgrid_features_bit_exact(oracle, gpu_world):
    model, body, fp, init, poses, tr = gpu_world
    field, keep = oracle.make_field(fp)
    rng = np.random.RandomState(11)
    bb = init["bbox"]
    x = (rng.rand(50003, 3) * (bb[1] - bb[0]) * 1.1 + bb[0] - 0.05 * (bb[1] - bb[0])).astype(np.float32)  # incl. clamped
    x[:3] = [bb[0], bb[1], (bb[0] + bb[1]) / 2]
    ref = oracle.hashgrid(field, x).view(np.uint16)
    got = model.net_coarse.encode(torch.as_tensor(x, device=DEV)).cpu().numpy().view(np.uint16)
    mism = (ref != got).any(1).mean()
    # pos = x*scale+0.5 is an FMA on the GPU and mul+add on the host: a cell flip at an
    # exact cell boundary is continuous in value but not bitwise -> allow 1e-4 of samples
    assert mism < 1e-4, mism
    d = np.abs(ref.view(np.float16).astype(np.float32) - got.view(np.float16).astype(np.float32)).max()
    assert d < 2e-3, d


def test_field_forward(oracle, gpu_world):
    model, body, fp, init, poses, tr = gpu_world
    field, keep = oracle.make_field(fp)
    rng = np.random.RandomState(12)
    bb = init["bbox"]
    for n in (1, 63, 64, 65, 40007):
        x = (rng.rand(n, 3) * (bb[1] - bb[0]) + bb[0]).astype(np.float32)
        rgb_o, sig_o = oracle.field_fwd(field, x)
        with torch.no_grad():
            rgb_g, sig_g = model.net_coarse(torch.as_tensor(x, device=DEV), None)
        rgb_g, sig_g = rgb_g.cpu().numpy(), sig_g.cpu().numpy()
        # fp16-rounded outputs: equal except where the fp32 sum order flips a rounding
        assert np.abs(rgb_g - rgb_o).max() < 2e-3
        tol = 2e-3 * np.maximum(1.0, np.abs(sig_o))
        assert (np.abs(sig_g - sig_o) <= tol).all()
        if n > 1000:
            assert (sig_g == sig_o).mean() > 0.97 and (rgb_g == rgb_o).mean() > 0.97
            assert sig_o.max() > 50 and sig_o.min() < -50  # the synthetic field is not trivial


def test_xcd_sharded_encoding_equals_fused_kernel(gpu_world):
    """The level-sharded encoding (one hashed level per XCD L2, x-neighbour pair loads) must
    produce bit-identical features, and the two-pass field bit-identical rgb / sigma."""
    model, body, fp, init, poses, tr = gpu_world
    rng = np.random.RandomState(21)
    bb = init["bbox"]
    for n_levels in (16, 8):
        net = model.net_coarse if n_levels == 16 else W.build(DEV, 64, 8)[0].net_coarse
        for n in (1, 1023, 1025, 8192, 100003):
            x = (rng.rand(n, 3) * (bb[1] - bb[0]) * 1.1 + bb[0] - 0.05 * (bb[1] - bb[0])).astype(np.float32)
            xt = torch.as_tensor(x, device=DEV)
            row = net.encode(xt).view(torch.int32)                       # [V, L] packed half2
            for coherent in (True, False):                               # ia_field.enc_split: 3 / 2 tiles of four to XCDs 0-3
                net.sample_coherence(coherent)
                planes = net.encode_planes(xt)                           # [L, V]
                assert torch.equal(planes.t().contiguous(), row), (n_levels, n, coherent)
            net.sample_coherence(True)
            with torch.no_grad():
                saved = net.max_encode_workspace_bytes
                try:
                    net.max_encode_workspace_bytes = 0
                    net._enc_ws_samples, net._enc_ws, net._desc = 0, None, None
                    rgb0, sig0 = net(xt, None)                           # fused single kernel
                finally:
                    net.max_encode_workspace_bytes = saved
                net._desc = None
                rgb1, sig1 = net(xt, None)                               # sharded two-pass (n >= 8192)
            assert torch.equal(rgb0, rgb1) and torch.equal(sig0, sig1), (n_levels, n)


def test_deform_query_fused_vs_oracle(oracle, gpu_world):
    model, body, fp, init, poses, tr = gpu_world
    _prepare(model, poses, tr, 1)
    ow = W.oracle_world(oracle, body, fp, init, poses[1], tr[1])
    # feed the oracle the GPU's transforms so that only the stage under test differs
    ow["tfs"] = model.deformer.tfs[0].cpu().numpy()
    ow["voxel_J"] = np.ascontiguousarray(model.deformer.deformer.voxel_J[0].cpu().numpy())
    pts = _query_points(model, 30001, 5)
    rgb_o, sig_o = oracle.deform_query(pts, ow, True)
    rgb_g, sig_g = model.deformer(torch.as_tensor(pts, device=DEV), model.net_coarse, True)
    rgb_g, sig_g = rgb_g.cpu().numpy(), sig_g.cpu().numpy()
    bad = np.abs(sig_g - sig_o) > 2e-3 * np.maximum(1, np.abs(sig_o))
    assert bad.mean() < 5e-4, bad.mean()
    ok = ~bad & (sig_o > 0)
    assert np.abs(rgb_g[ok] - rgb_o[ok]).max() < 2e-3
    assert (sig_g >= 0).all()  # Q11: invalid candidates contribute 0 at test time
    # closure route == fused route
    rgb_c, sig_c = model.deformer(torch.as_tensor(pts, device=DEV), lambda x, d: model.net_coarse(x, d), True)
    assert torch.equal(torch.as_tensor(sig_g), sig_c.cpu())


def test_occupancy_postprocess_bit_exact(oracle, gpu_world):
    model = gpu_world[0]
    G = 64
    rng = np.random.RandomState(2)
    dens = np.zeros((G, G, G), np.float32)
    dens[10:30, 20:40, 5:50] = rng.rand(20, 20, 45) * 300
    dens[40:44, 40:44, 40:44] = 200           # separate small component
    dens[50, 50, 50] = 1e-3                     # below threshold
    ref = oracle.occupancy_from_density(dens, G)
    grid = model.renderer.density_grid_test
    grid._postprocess(torch.as_tensor(dens, device=DEV))
    got = grid.density_field.cpu().numpy()
    assert np.array_equal(got, ref.astype(bool))
    bits = grid.occ_bits.cpu().numpy().view(np.uint32)[:G ** 3 // 32]  # (+1 border-flag word)
    unpacked = ((bits[:, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(G, G, G).astype(bool)
    assert np.array_equal(unpacked, got)
    grid._postprocess(torch.zeros((G, G, G), device=DEV))   # empty grid edge case
    assert grid.density_field.sum().item() == 0


def test_occupancy_components_adversarial_cases(oracle, gpu_world):
    """The occupancy post-process (density_grid.py:104-125) on adversarial inputs: two components of EQUAL size
    (torch.mode keeps the smaller label), diagonal-only (26-) connectivity, a component touching the border (border
    flag), sparse noise with hundreds of components, large blobs, an almost full grid, runs at the column ends; 64^3
    grids take the run-based union kernels, other sizes the cell-based one."""
    model = gpu_world[0]
    G = 64
    grid = model.renderer.density_grid_test
    rng = np.random.RandomState(5)

    def check(dens, what):
        ref = oracle.occupancy_from_density(dens, G).astype(bool)
        grid._postprocess(torch.as_tensor(dens, device=DEV))
        got = grid.density_field.cpu().numpy()
        assert np.array_equal(got, ref), (what, int(got.sum()), int(ref.sum()))
        words = grid.occ_bits.cpu().numpy().view(np.uint32)
        unpacked = ((words[:G ** 3 // 32, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(G, G, G).astype(bool)
        assert np.array_equal(unpacked, got), what
        border = got.copy(); border[1:-1, 1:-1, 1:-1] = False
        assert int(words[G ** 3 // 32]) == (0 if border.any() else 1), what   # flag word: 1 = no border cell occupied
        occ = np.argwhere(got)                                                # the six words behind it: bounds of the occupied cells
        want = [v for c in range(3) for v in ((int(occ[:, c].min()), int(occ[:, c].max())) if len(occ) else (0x7fffffff, -1))]
        assert words[G ** 3 // 32 + 1:G ** 3 // 32 + 7].view(np.int32).tolist() == want, (what, want)
        return int(got.sum())

    d = np.zeros((G, G, G), np.float32)
    d[10:14, 10:14, 10:14] = 100; d[40:44, 40:44, 40:44] = 100          # equal sizes: the component with the smaller label wins
    n = check(d, "tie")
    assert n == 6 ** 3
    d = np.zeros((G, G, G), np.float32)
    for k in range(20):
        d[20 + k, 20 + k, 20 + k] = 50                                     # a diagonal chain (dilated by the max-pool)
    d[5:8, 50:60, 30:33] = 80
    check(d, "diagonal")
    d = np.zeros((G, G, G), np.float32)
    d[0:6, 30:36, 28:40] = 120                                            # touches x = 0
    assert check(d, "border") > 0
    d = (rng.rand(G, G, G) > 0.9995).astype(np.float32) * 70                # ~130 isolated seeds -> many 27-cell components
    d[30:36, 30:36, 30:36] = 90
    check(d, "noise")
    for side in (22, 24):                                                    # (side + 2)^3 cells after dilation
        d = np.zeros((G, G, G), np.float32)
        d[8:8 + side, 8:8 + side, 8:8 + side] = rng.rand(side, side, side) * 200 + 50
        d[50:53, 50:53, 50:53] = 60
        assert check(d, "blob %d" % side) == (side + 2) ** 3
    d = rng.rand(G, G, G).astype(np.float32) * 100                          # untrained-field-like: most of the grid occupied
    check(d, "dense")
    d = np.zeros((G, G, G), np.float32)                                      # runs that only touch diagonally across columns / at z = 0, 63
    d[10, 10, 0:3] = 50; d[13, 13, 5:9] = 50; d[16, 15, 60:64] = 50; d[19, 17, 0:64] = 50; d[22, 20, 31:33] = 50
    check(d, "column ends")
    # other grid sizes take the cell-based union kernel (the run-based one needs a z-column to be exactly one wave)
    from instantavatar_amd.models.structures.density_grid import DensityGrid
    g32 = DensityGrid(32).to(DEV)
    d = np.zeros((32, 32, 32), np.float32)
    d[4:14, 6:20, 3:9] = rng.rand(10, 14, 6) * 100 + 20
    d[24:27, 24:27, 24:27] = 70
    g32._postprocess(torch.as_tensor(d, device=DEV))
    assert np.array_equal(g32.density_field.cpu().numpy(), oracle.occupancy_from_density(d, 32).astype(bool))
    # flag + bounds of the occupied cells behind the bit grid on other grid sizes (32: the power-of-two route of k_occ_bounds, 24 and
    # 40: the generic one -- a word of the bit grid spans several z-columns there)
    for Gx in (32, 24, 40):
        gx = DensityGrid(Gx).to(DEV)
        d = np.zeros((Gx, Gx, Gx), np.float32)
        d[3:9, 5:Gx - 4, 2:7] = rng.rand(6, Gx - 9, 5) * 100 + 20
        gx._postprocess(torch.as_tensor(d, device=DEV))
        got = gx.density_field.cpu().numpy()
        assert np.array_equal(got, oracle.occupancy_from_density(d, Gx).astype(bool)) and got.any()
        occ = np.argwhere(got)
        tail = gx.occ_bits.cpu().numpy()[Gx ** 3 // 32:Gx ** 3 // 32 + 7].tolist()
        assert tail == [1] + [v for c in range(3) for v in (int(occ[:, c].min()), int(occ[:, c].max()))], (Gx, tail)


def test_raymarch_and_composite_kernels(oracle, gpu_world):
    model = gpu_world[0]
    G = 64
    rng = np.random.RandomState(4)
    occ = np.zeros((G, G, G), np.uint8)
    occ[20:44, 10:54, 24:40] = 1
    occ[rng.rand(G, G, G) > 0.97] = 1
    aabb = np.array([[-1.0, -1.2, -0.6], [1.0, 0.9, 0.7]], np.float32)
    N = 3001
    o = np.tile(np.array([[0.0, 0.0, -4.0]], np.float32), (N, 1)) + rng.randn(N, 3).astype(np.float32) * 0.01
    d = rng.randn(N, 3).astype(np.float32) * 0.15 + np.array([0, 0, 1], np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    dist = np.linalg.norm(o, axis=1).astype(np.float32)
    near, far = dist - 1, dist + 1
    step = ((far - near) / 256).astype(np.float32)
    alive = np.sort(rng.choice(N, 2000, replace=False)).astype(np.int64)
    for Ns in (1, 7, 256):
        near_o = near.copy()
        pts = np.empty((len(alive), Ns, 3), np.float32); dn = np.empty((len(alive), Ns), np.float32); zn = np.empty_like(dn)
        oracle.lib().orc_raymarch_test(*[a.ctypes.data_as(C.c_void_p) for a in (o, d, near_o, far, alive)], C.c_long(len(alive)),
                                       occ.ctypes.data_as(C.c_void_p), G, (aabb[1] - aabb[0]).ctypes.data_as(C.c_void_p),
                                       aabb[0].ctypes.data_as(C.c_void_p), step.ctypes.data_as(C.c_void_p), Ns,
                                       pts.ctypes.data_as(C.c_void_p), dn.ctypes.data_as(C.c_void_p), zn.ctypes.data_as(C.c_void_p))
        t = lambda a: torch.as_tensor(a, device=DEV)
        bits = torch.zeros(G ** 3 // 32 + 8, dtype=torch.int32, device=DEV)  # + border flag + occupied-cell bounds (8 tail words)
        tocc = t(occ)
        _lib.check(_lib.lib().ia_occupancy_pack(_lib.ptr(tocc), G, _lib.ptr(bits), _lib.stream()))
        og = _lib.OccGrid(); og.G = G; og.aabb_min[:] = aabb[0].tolist(); og.aabb_max[:] = aabb[1].tolist()
        near_g = t(near.copy()); pg = torch.empty((len(alive), Ns, 3), device=DEV); dg = torch.empty((len(alive), Ns), device=DEV); zg = torch.empty_like(dg)
        to, td, tf, ta, ts = t(o), t(d), t(far), t(alive), t(step)
        _lib.check(_lib.lib().ia_raymarch_test(_lib.ptr(to), _lib.ptr(td), _lib.ptr(near_g), _lib.ptr(tf), _lib.ptr(ta), len(alive),
                                               _lib.ptr(bits), C.byref(og), _lib.ptr(ts), Ns, _lib.ptr(pg), _lib.ptr(dg), _lib.ptr(zg),
                                               _lib.stream()))
        # depths are produced by the same sequence of float adds -> bit exact; positions differ by FMA only
        assert np.array_equal(zg.cpu().numpy(), zn) and np.array_equal(dg.cpu().numpy(), dn)
        assert np.array_equal(near_g.cpu().numpy(), near_o)
        assert np.abs(pg.cpu().numpy() - pts).max() < 1e-6
        # composite on random field values
        rgbv = rng.rand(len(alive), Ns, 3).astype(np.float32); sig = (rng.randn(len(alive), Ns) * 60).astype(np.float32)
        col = np.zeros((N, 3), np.float32); dep = np.zeros(N, np.float32); nh = np.ones(N, np.float32)
        oracle.lib().orc_composite_test(*[a.ctypes.data_as(C.c_void_p) for a in (rgbv, sig, dn, zn, alive)], C.c_long(len(alive)), Ns,
                                        col.ctypes.data_as(C.c_void_p), dep.ctypes.data_as(C.c_void_p), nh.ctypes.data_as(C.c_void_p), C.c_float(0.01))
        cg = torch.zeros((N, 3), device=DEV); dpg = torch.zeros(N, device=DEV); nhg = torch.ones(N, device=DEV)
        trgb, tsig = t(rgbv), t(sig)  # keep the device copies alive across the call
        _lib.check(_lib.lib().ia_composite_test(_lib.ptr(trgb), _lib.ptr(tsig), _lib.ptr(dg), _lib.ptr(zg), _lib.ptr(ta), len(alive), Ns,
                                                _lib.ptr(cg), _lib.ptr(dpg), _lib.ptr(nhg), 0.01, _lib.stream()))
        assert np.abs(cg.cpu().numpy() - col).max() < 2e-5 and np.abs(nhg.cpu().numpy() - nh).max() < 2e-5
        assert np.abs(dpg.cpu().numpy() - dep).max() < 2e-4


def _frame_parity(oracle, model, body, fp, init, pose, transl, res, jit_seed, iters=2):
    G = 64
    jit = np.random.RandomState(jit_seed).rand(iters, G ** 3, 3).astype(np.float32)
    batch = make_batch(DEV, res, pose, transl)
    rgb, depth, alpha, counter = model.render_image_fast(batch, (res, res), jitter=torch.as_tensor(jit, device=DEV))
    ow = W.oracle_world(oracle, body, fp, init, pose, transl)
    ro, rd = syn.make_camera_rays(res)
    ref = oracle.render_image_fast(ow, ro, rd, jit)
    rgb, alpha = rgb.reshape(-1, 3).cpu().numpy(), alpha.reshape(-1).cpu().numpy()
    occ_g = model.renderer.density_grid_test.density_field.cpu().numpy()
    return rgb, alpha, depth.reshape(-1).cpu().numpy(), counter.reshape(-1).cpu().numpy(), occ_g, ref


def test_render_frame_parity_full_pipeline(oracle, gpu_world):
    """DNeRFModel.render_image_fast end to end (SMPL chain -> precompute -> occupancy
    build -> fused render loop) vs the oracle: rgb / alpha within 1e-3."""
    model, body, fp, init, poses, tr = gpu_world
    for i, res in ((2, 64), (6, 96)):
        rgb, alpha, depth, counter, occ_g, ref = _frame_parity(oracle, model, body, fp, init, poses[i], tr[i], res, 100 + i)
        W.cells_within(occ_g, ref["occ"].astype(bool), "frame %dx%d occupancy" % (res, res))
        cov = (ref["alpha"] > 0.5).mean()
        assert cov > 0.02
        err_rgb = np.abs(rgb - ref["rgb"]).max(1)
        err_a = np.abs(alpha - ref["alpha"])
        # discontinuities (occupancy cell flips, alpha<0.01 skips) may move single rays
        W.rays_within(err_rgb, "frame %dx%d rgb" % (res, res))
        W.rays_within(err_a, "frame %dx%d alpha" % (res, res))
        assert np.median(err_rgb[ref["alpha"] > 0.5]) < 1e-4
        assert abs(counter.mean() - ref["counter"].mean()) < 0.02 * max(1.0, ref["counter"].mean())


@pytest.mark.parametrize("which", ["own", "other"])
def test_flat_tcnn_checkpoint_loads_and_renders_like_the_oracle(oracle, tmp_path, monkeypatch, which):
    """VERDICT r03 missing 4 (second half): the first real checkpoint arrives as a Lightning dict whose field is two FLAT
    tcnn vectors -- `net_coarse.encoder.params` = [W1 64x32 | W2 16x64 | grid] and `net_coarse.color_net.params` =
    [64x16 | 64x64 | 16x64] (ngp.py:27-58, animate.py:92-95 loads it) -- in whichever of the two level-3 layouts that tcnn
    build produced.  Written here from a synthetic field in BOTH layouts, loaded through drivers/checkpoint.load_checkpoint
    into a model built for this host's default layout, rendered, and compared with the oracle evaluating the same field."""
    import os
    from instantavatar_amd.drivers.checkpoint import load_checkpoint
    from instantavatar_amd.pipeline import build_synthetic_model
    own = int(os.environ.get("IA_TCNN_LEVEL3_RES", "54"))
    r3 = own if which == "own" else (55 if own == 54 else 54)
    _, body, _, init = W.build(DEV, 64, 16)
    model, _, _ = build_synthetic_model(DEV, resolution=64, n_levels=16)       # default layout; its own field is overwritten below
    assert int(model.net_coarse.hash_desc.res[3]) == own
    smpl = model.deformer.body_model
    cano = smpl(betas=torch.zeros(1, 10, device=DEV), body_pose=torch.as_tensor(syn.cano_pose("A_pose"), device=DEV)[None],
                return_verts=False).joints[0].cpu().numpy()
    monkeypatch.setenv("IA_TCNN_LEVEL3_RES", str(r3))     # the generator and the oracle follow the environment
    fp = syn.make_field(cano, model.deformer.bbox.cpu().numpy(), seed=7, n_levels=16)
    assert int(fp["level_res"][3]) == r3
    flat = lambda *ks: torch.from_numpy(np.concatenate([np.asarray(fp[k], np.float32).reshape(-1) for k in ks]))
    sd = {"net_coarse.encoder.params": flat("sig_w1", "sig_w2", "table"), "net_coarse.color_net.params": flat("col_w1", "col_w2", "col_w3"),
          "net_coarse.center": torch.as_tensor(fp["center"])[None], "net_coarse.scale": torch.as_tensor(fp["scale"])[None],
          "loss_fn.lpips.net.slice1.0.weight": torch.zeros(3)}      # something a Lightning checkpoint carries that is not on the path
    sd = {k: (v.reshape(model.state_dict()[k].shape) if k in model.state_dict() and v.numel() == model.state_dict()[k].numel() else v) for k, v in sd.items()}
    path = str(tmp_path / "lightning.ckpt")
    torch.save({"state_dict": sd, "global_step": 1234, "epoch": 3, "pytorch-lightning_version": "1.5.7"}, path)
    missing, unexpected = load_checkpoint(model, path)
    assert int(model.net_coarse.hash_desc.res[3]) == r3 and model.global_step == 1234
    assert "loss_fn.lpips.net.slice1.0.weight" in unexpected
    assert model.tcnn_self_check["level3_res"] == r3 and model.tcnn_self_check["finite"]
    poses, tr = W.poses()
    rgb, alpha, depth, counter, occ_g, ref = _frame_parity(oracle, model, body, fp, init, poses[2], tr[2], 64, 321)
    err_rgb, err_a = np.abs(rgb - ref["rgb"]).max(1), np.abs(alpha - ref["alpha"])
    print("flat tcnn checkpoint, level-3 resolution %d: cov %.3f, rays > 1e-3: rgb %.5f alpha %.5f, max %.2e, occupancy flips %.2e" % (
        r3, (ref["alpha"] > 0.5).mean(), (err_rgb > 1e-3).mean(), (err_a > 1e-3).mean(), err_rgb.max(), (occ_g != ref["occ"].astype(bool)).mean()))
    assert (ref["alpha"] > 0.5).mean() > 0.02
    W.cells_within(occ_g, ref["occ"].astype(bool), "flat tcnn checkpoint (level-3 %d) occupancy" % r3)
    W.rays_within(err_rgb, "flat tcnn checkpoint (level-3 %d) rgb" % r3)
    W.rays_within(err_a, "flat tcnn checkpoint (level-3 %d) alpha" % r3)
    assert np.median(err_rgb[ref["alpha"] > 0.5]) < 1e-4


def test_render_closure_route_equals_fused_route(gpu_world):
    model, body, fp, init, poses, tr = gpu_world
    res = 64
    batch = make_batch(DEV, res, poses[3], tr[3])
    jit = torch.rand((2, 64 ** 3, 3), device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
    rgb_f, depth_f, alpha_f, cnt_f = model.render_image_fast(batch, (res, res), jitter=jit)
    rays = Rays(o=batch["rays_o"], d=batch["rays_d"], near=batch["near"], far=batch["far"])
    model.deformer.transform_rays_w2s(rays)
    from instantavatar_amd import dense_routes
    out = dense_routes.render_test(model.renderer, rays, lambda x, _: model.deformer(x, lambda p, d: model.net_coarse(p, d), True), None)
    assert torch.allclose(out["rgb_coarse"].reshape(-1, 3), rgb_f.reshape(-1, 3), atol=1e-6)
    assert torch.allclose(out["alpha_coarse"].reshape(-1), alpha_f.reshape(-1), atol=1e-6)
    assert torch.equal(out["counter_coarse"].reshape(-1), cnt_f.reshape(-1))


def test_config0_identity_pose_8_levels(oracle):
    """BASELINE configs[0]: 128x128, canonical pose (identity deformer), 8-level grid."""
    model, body, fp, init = W.build(DEV, 64, 8)
    pose = np.concatenate([np.zeros(3, np.float32), syn.cano_pose("A_pose")])
    pose[0] = np.pi  # face the camera
    transl = np.array([0, 0.15, 5], np.float32)
    rgb, alpha, depth, counter, occ_g, ref = _frame_parity(oracle, model, body, fp, init, pose, transl, 128, 9)
    W.rays_within(np.abs(rgb - ref["rgb"]).max(1), "config 0 (128^2, canonical pose, 8 levels) rgb")
    W.rays_within(np.abs(alpha - ref["alpha"]), "config 0 (128^2, canonical pose, 8 levels) alpha")
    assert (ref["alpha"] > 0.5).mean() > 0.03


def test_render_is_idempotent_and_background_linear(gpu_world):
    """Size-independent properties at the full 512x512 bench size."""
    model, body, fp, init, poses, tr = gpu_world
    res = 512
    batch = make_batch(DEV, res, poses[0], tr[0])
    jit = torch.rand((5, 64 ** 3, 3), device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    a = model.render_image_fast(batch, (res, res), jitter=jit)
    b = model.render_image_fast(make_batch(DEV, res, poses[0], tr[0]), (res, res), jitter=jit)
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
    batch2 = make_batch(DEV, res, poses[0], tr[0])
    batch2["bg_color"] = torch.zeros((1, res * res, 3), device=DEV)
    c = model.render_image_fast(batch2, (res, res), jitter=jit)
    T = (1 - a[2]).reshape(-1, 1)
    assert torch.allclose(a[0].reshape(-1, 3) - c[0].reshape(-1, 3), T.expand(-1, 3), atol=1e-6)
    assert 0.03 < (a[2] > 0.5).float().mean().item() < 0.5
    assert (a[0] >= 0).all() and (a[0] <= 1 + 1e-5).all()


def test_hip_graph_replay_equals_eager(gpu_world):
    """The captured per-frame HIP graph must reproduce the eager launches bit for bit (same
    jitter stream is not available across the two modes, so the occupancy jitter is fixed)."""
    from instantavatar_amd.pipeline import GraphedRenderer
    model, body, fp, init, poses, tr = gpu_world
    res = 128
    grid = model.renderer.density_grid_test
    orig = grid.initialize
    jit = torch.rand((5, 64 ** 3, 3), device=DEV, generator=torch.Generator(device=DEV).manual_seed(11))
    grid.initialize = lambda deformer, net, iters=5, jitter=None: orig(deformer, net, iters=iters, jitter=jit)
    try:
        g = GraphedRenderer(model, make_batch(DEV, res, poses[0], tr[0]), (res, res), sync_check=True)
        for i in (1, 4, 6):
            b = make_batch(DEV, res, poses[i], tr[i])
            out_g = [t.clone() for t in g(b)]
            out_e = model.render_image_fast(make_batch(DEV, res, poses[i], tr[i]), (res, res))
            for a, e in zip(out_g, out_e):
                assert torch.equal(a, e)
            assert (out_e[2] > 0.5).float().mean() > 0.02
        assert g.finish() == 0
    finally:
        grid.initialize = orig


def test_replays_with_too_few_wavefront_iterations_are_reported_and_rerendered(gpu_world):
    """A captured frame holds a FIXED number of wave-front iterations (what the probe frames needed + margin); a frame whose rays
    are still alive after them is incomplete.  The renderers must report exactly those calls (`incomplete_calls`, global call
    numbers, whichever replica of a PipelinedRenderer rendered them -- least-loaded schedule included) so that the caller renders
    them again (drivers/animate.py, bench.py): here the graphs are captured one iteration short of what the longest frame needs;
    exactly the calls that need more are reported, every other call must equal the eager frame bit for bit."""
    from instantavatar_amd.pipeline import GraphedRenderer, PipelinedRenderer
    import instantavatar_amd.pipeline as P
    model, body, fp, init, poses, tr = gpu_world
    res = 96
    jit = torch.rand((5, 64 ** 3, 3), device=DEV, generator=torch.Generator(device=DEV).manual_seed(31))
    frames = (1, 4, 6, 2, 7, 3, 5)
    eager = {i: [t.clone() for t in model.render_image_fast(make_batch(DEV, res, poses[i], tr[i]), (res, res), jitter=jit)] for i in frames}
    need = {}
    for i in frames:
        model.render_image_fast(make_batch(DEV, res, poses[i], tr[i]), (res, res), jitter=jit)
        need[i] = model.renderer.iters_executed()
    hint_before = model.renderer._iters_hint
    model.render_image_fast(make_batch(DEV, res, poses[0], tr[0]), (res, res), jitter=jit)
    need0 = model.renderer.iters_executed()               # what the renderers' warm-up on this batch will measure
    margin = max(need.values()) - 1 - need0               # one iteration less than the longest frame needs: those frames are incomplete
    try:
        for make in ("graphed", "pipelined"):
            if make == "graphed":
                r = GraphedRenderer(model, make_batch(DEV, res, poses[0], tr[0]), (res, res), margin=margin, jitter=jit)
                budget = [model.renderer._iters_hint]
            else:
                r = PipelinedRenderer(model, make_batch(DEV, res, poses[0], tr[0]), (res, res), n_in_flight=3, margin=margin, jitter=jit)
                assert r.schedule == "least_loaded"
                budget = [m.renderer._iters_hint for m in r.replicas]
            outs = []
            for i in frames:
                b = make_batch(DEV, res, poses[i], tr[i])
                if make == "graphed":
                    outs.append([t.clone() for t in r(b)])
                else:
                    r(b, consume=lambda out, k: outs.append([t.clone() for t in out]))
            if make == "pipelined":
                r.synchronize()
            n_bad = r.finish()
            bad = sorted(r.incomplete_calls)
            assert n_bad == len(bad) and all(0 <= c < len(frames) for c in bad), (make, n_bad, bad)
            # a call is reported exactly when its frame needs more iterations than its graph holds (every replica holds the same number)
            assert len(set(budget)) == 1
            want = [c for c, i in enumerate(frames) if need[i] > budget[0]]
            assert bad == want and len(bad) > 0, (make, bad, want, need, budget)
            for c, i in enumerate(frames):
                if c in bad:
                    continue
                for a, e in zip(outs[c], eager[i]):
                    assert torch.equal(a, e), (make, c, i)
    finally:
        model.renderer._iters_hint = hint_before


def test_march_empty_space_skip_is_exact(gpu_world):
    """k_march_compact skips the occupancy arithmetic of the steps that cannot be occupied when the grid's border flag says that
    no border cell is occupied (the clamped cell of a point outside the grid is a border cell, raymarcher.cu:49-51): the part of a
    ray outside the box of the interior cells advances by the same float adds only.  Exactness: the same frame rendered with the
    flag as the occupancy post-process computed it (1) and with the flag word cleared (0 = every step tested, rounds 1-5) must be
    identical bit for bit -- on real poses, and on an adversarial grid whose occupied cells fill the whole interior box (rays
    graze its faces, edges and corners one cell inside the border layer)."""
    model, body, fp, init, poses, tr = gpu_world
    res = 160
    grid = model.renderer.density_grid_test
    G = grid.grid_size

    FLAG = G ** 3 // 32      # tail of the bit grid: [flag, x_min, x_max, y_min, y_max, z_min, z_max, spare]

    def both(b, what):
        flag = int(grid.occ_bits[FLAG])
        outs = []
        for f in (flag, 0):
            grid.occ_bits[FLAG] = f
            with torch.no_grad():
                d = model.forward(b, eval_mode=True)
            outs.append({k: v.clone() for k, v in d.items() if torch.is_tensor(v)})
        grid.occ_bits[FLAG] = flag
        for k in outs[0]:
            assert torch.equal(outs[0][k], outs[1][k]), (what, k, int((outs[0][k] != outs[1][k]).sum()))
        return flag, outs[0]

    flags = []
    for i in (0, 3, 5, 7):
        b = make_batch(DEV, res, poses[i], tr[i])
        model.render_image_fast(b, (res, res))          # prepares the deformer and builds the pose's occupancy grid (+ flag)
        flag, out = both(b, "pose %d" % i)
        flags.append(flag)
        occ = grid.density_field.nonzero()
        if flag == 1:     # the bounds the post-process left behind the flag are those of the occupied cells
            want = [v for c in range(3) for v in (int(occ[:, c].min()), int(occ[:, c].max()))]
            assert grid.occ_bits[FLAG + 1:FLAG + 7].tolist() == want, (grid.occ_bits[FLAG:FLAG + 7].tolist(), want)
        assert float(out["alpha_coarse"].max()) > 0.5
    assert 1 in flags, flags     # (a body that touched its bounding box in every pose would leave the skip untested)
    # adversarial grid: every interior cell occupied, the border layer empty
    b = make_batch(DEV, res, poses[3], tr[3])
    model.render_image_fast(b, (res, res))
    keep = (grid.density_field.clone(), grid.occ_bits.clone())
    try:
        f = torch.zeros((G, G, G), dtype=torch.bool, device=DEV)
        f[1:G - 1, 1:G - 1, 1:G - 1] = True
        grid.density_field = f
        grid.pack_bits()
        assert grid.occ_bits[FLAG:FLAG + 7].tolist() == [1, 1, G - 2, 1, G - 2, 1, G - 2]
        flag, out = both(b, "full interior box")
        assert float(out["counter_coarse"].max()) > 10
        f[0, G // 2, G // 2] = True                      # one occupied border cell: the flag must drop, nothing may be skipped
        grid.density_field = f
        grid.pack_bits()
        assert grid.occ_bits[FLAG:FLAG + 3].tolist() == [0, 0, G - 2]
        # a thin diagonal set of cells: rays cross the bounds box at every angle, most of them without a hit
        f = torch.zeros((G, G, G), dtype=torch.bool, device=DEV)
        ar = torch.arange(8, G - 8, device=DEV)
        f[ar, ar, (ar * 3) % (G - 16) + 8] = True
        grid.density_field = f
        grid.pack_bits()
        assert int(grid.occ_bits[FLAG]) == 1 and grid.occ_bits[FLAG + 1:FLAG + 5].tolist() == [8, G - 9, 8, G - 9]
        both(b, "diagonal cells")
    finally:
        grid.density_field, _ = keep
        grid.occ_bits.copy_(keep[1])


@pytest.mark.parametrize("n_in_flight,schedule", [(2, "round_robin"), (3, "round_robin"), (3, "least_loaded")])
def test_pipelined_renderer_frames_in_flight_equal_eager(gpu_world, n_in_flight, schedule):
    """PipelinedRenderer: two / three replicas (shared weights, own workspaces), one captured graph and one stream each.  Every
    frame must equal the eager render bit for bit whichever replica rendered it, and both replicas must see a weight
    update (the network parameters are shared by reference, not copied)."""
    from instantavatar_amd.pipeline import PipelinedRenderer, clone_for_stream
    model, body, fp, init, poses, tr = gpu_world
    res = 96
    jit = torch.rand((5, 64 ** 3, 3), device=DEV, generator=torch.Generator(device=DEV).manual_seed(21))
    clone = clone_for_stream(model)
    assert clone.net_coarse.encoder.params is model.net_coarse.encoder.params
    assert clone.deformer.deformer.lbs_voxel_final.data_ptr() == model.deformer.deformer.lbs_voxel_final.data_ptr()
    origs = []
    try:
        pr = None
        # fixed jitter for every replica's occupancy grid (the graphs read it from a static buffer)
        def patch(m):
            g = m.renderer.density_grid_test
            o = g.initialize
            g.initialize = lambda deformer, net, iters=5, jitter=None, _o=o: _o(deformer, net, iters=iters, jitter=jit)
            origs.append((g, o))
        patch(model)
        import instantavatar_amd.pipeline as P
        real_clone = P.clone_for_stream

        def patched_clone(m):
            c = real_clone(m)
            patch(c)
            return c
        P.clone_for_stream = patched_clone
        try:
            pr = PipelinedRenderer(model, make_batch(DEV, res, poses[0], tr[0]), (res, res), n_in_flight=n_in_flight, schedule=schedule)
        finally:
            P.clone_for_stream = real_clone
        assert pr.priorities == ([0, 0] if n_in_flight == 2 else [-1, 0, 0])   # (the first replica's stream at high priority from three on)
        outs = []
        for i in (1, 4, 6, 2):
            o, k = pr(make_batch(DEV, res, poses[i], tr[i]), consume=lambda out, k: outs.append([t.clone() for t in out]))
        pr.synchronize()
        assert pr.calls == 4 and pr.finish() == 0
        for n, i in enumerate((1, 4, 6, 2)):
            ref = model.render_image_fast(make_batch(DEV, res, poses[i], tr[i]), (res, res))
            for a, e in zip(outs[n], ref):
                assert torch.equal(a, e), (n, i, float((a.float() - e.float()).abs().max()), int((a != e).sum()))
        # shared weights: an in-place change of the master parameters reaches both replicas
        with torch.no_grad():
            model.net_coarse.color_net.params.mul_(0.5)
        pr.refresh_weights()
        ab = []      # (cloned on the replica's stream: with the least-loaded schedule both calls may land on the same replica)
        for _ in range(n_in_flight):
            pr(make_batch(DEV, res, poses[1], tr[1]), consume=lambda out, k: ab.append(out[0].clone()))
        pr.synchronize()
        assert all(torch.equal(ab[0], x) for x in ab[1:]) and not torch.equal(ab[0], outs[0][0])
        assert sum(pr.frames_per_replica) == pr.calls == 4 + n_in_flight and sorted(c for ids in pr._call_ids for c in ids) == list(range(pr.calls))
    finally:
        with torch.no_grad():
            model.net_coarse.color_net.params.mul_(2.0)
        model.net_coarse.refresh()
        torch.cuda.synchronize()
        for g, o in origs:
            g.initialize = o
