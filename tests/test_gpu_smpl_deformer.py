"""GPU parity of the SMPLDeformer plugin: nearest-vertex kernel, per-frame
preparation, fused field query and a rendered frame through the renderer's closure route, each
against the CPU oracle on identical inputs."""
import numpy as np
import pytest
import torch

from instantavatar_amd import synthetic as syn
from instantavatar_amd.pipeline import make_batch

import world as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", params=[False, True], ids=["zero-blendshapes", "blendshapes"])
def sw(request):
    """both bodies: zero shape / pose directions and the blend-shape body with synthetic.BLEND_BETAS -- the pose
    offsets enter T_inv (smpl_deformer.py:66-75) and the per-step `SMPL.forward(small_ops=True)` multiplies non-zero operands"""
    global BETAS
    model, body, fp = W.build_smpl_deformer_world(DEV, blend=request.param)
    BETAS = syn.BLEND_BETAS if request.param else np.zeros(10, np.float32)
    poses, tr = W.poses()
    return model, body, fp, poses, tr


BETAS = np.zeros(10, np.float32)


def _prep(oracle, sw, i, res=64):
    model, body, fp, poses, tr = sw
    batch = make_batch(DEV, res, poses[i], tr[i], betas=BETAS)
    model.deformer.prepare_deformer(batch)
    prep = oracle.smpl_deformer_prepare(body, BETAS, poses[i][3:], poses[i][:3], tr[i])
    return batch, prep


def test_prepare_matches_oracle(oracle, sw):
    model = sw[0]
    _, prep = _prep(oracle, sw, 2)
    d = model.deformer
    assert np.abs(d.vertices[0].cpu().numpy() - prep["vertices"]).max() < 2e-5
    assert np.abs(d.T_inv[0].cpu().numpy() - prep["T_inv"]).max() < 5e-5
    assert np.abs(d.w2s[0].cpu().numpy() - prep["w2s"]).max() < 1e-5
    assert np.abs(d.bbox.cpu().numpy() - prep["bbox"]).max() < 1e-5


def test_nearest_vertex_kernel_bit_exact(oracle, sw):
    """Same vertices / transforms on both sides: index, validity and canonical point must be equal."""
    model = sw[0]
    _prep(oracle, sw, 3)
    d = model.deformer
    v = d.vertices[0].cpu().numpy()
    T = d.T_inv[0].cpu().numpy()
    rng = np.random.RandomState(5)
    for n in (1, 255, 257, 40013):
        pts = (v[rng.randint(0, len(v), n)] + rng.randn(n, 3).astype(np.float32) * 0.04).astype(np.float32)
        pts[: min(n, 3)] = [[9, 9, 9], [0, 0, 0], [-9, 0, 3]][: min(n, 3)]
        cano_o, valid_o, _ = oracle.smpl_nn_deform(pts, v, T, d.threshold)
        cano_g, valid_g = d.deform(torch.as_tensor(pts, device=DEV))
        assert np.array_equal(valid_g.cpu().numpy(), valid_o)
        assert np.array_equal(cano_g.cpu().numpy(), cano_o)
    e, ev = d.deform(torch.zeros((0, 3), device=DEV))
    assert e.shape == (0, 3) and ev.shape == (0,)


def test_fused_query_matches_oracle_and_generic_route(oracle, sw):
    model, body, fp, poses, tr = sw
    _, prep = _prep(oracle, sw, 1)
    d, net = model.deformer, model.net_coarse
    field, keep = oracle.make_field(fp)
    v = d.vertices[0].cpu().numpy()
    rng = np.random.RandomState(6)
    pts = (v[rng.randint(0, len(v), 30011)] + rng.randn(30011, 3).astype(np.float32) * 0.05).astype(np.float32)
    prep_g = dict(prep, vertices=v, T_inv=d.T_inv[0].cpu().numpy())  # identical inputs for the query
    rgb_o, sig_o = oracle.smpl_deform_query(pts, prep_g, field, eval_mode=True)
    x = torch.as_tensor(pts, device=DEV)
    with torch.no_grad():
        rgb_g, sig_g = d(x, net, eval_mode=True)                       # fused kernel route
        rgb_c, sig_c = d(x, lambda p, dd: net(p, dd), eval_mode=True)   # reference structure (mask + scatter)
    assert torch.equal(rgb_g, rgb_c) and torch.equal(sig_g, sig_c)
    rgb_g, sig_g = rgb_g.cpu().numpy(), sig_g.cpu().numpy()
    assert np.abs(rgb_g - rgb_o).max() < 2e-3
    assert (np.abs(sig_g - sig_o) <= 2e-3 * np.maximum(1.0, np.abs(sig_o))).all()
    assert (sig_g == sig_o).mean() > 0.97 and (sig_o != 0).mean() > 0.2
    rgb_t, sig_t = d(x, net, eval_mode=False)                          # deform_train: invalid -> -1e5
    _, valid_o, _ = oracle.smpl_nn_deform(pts, v, prep_g["T_inv"], d.threshold)
    sig_t = sig_t.detach().cpu().numpy()
    assert ((sig_t == -1e5) == ~valid_o).all() and 0.05 < (~valid_o).mean() < 0.95
    assert np.array_equal(sig_t[valid_o], sig_g[valid_o])


def test_vertex_grid_query_equals_brute_force(oracle, sw):
    """The fused queries with the per-frame vertex grid (`ia_smpl_nn_grid_build`: a point is tested against the vertices of its
    27 cells) against the brute-force search over all 6 890 vertices: identical validity, identical nearest vertex and canonical
    position for every valid point, identical (rgb, sigma) -- on points around the body, far away, on the grid's faces, at vertices
    (distance 0: ties between coincident ring vertices are broken by the lowest index) and non-finite."""
    import ctypes as C
    from instantavatar_amd import _lib
    model = sw[0]
    _prep(oracle, sw, 4)
    d, net = model.deformer, model.net_coarse
    assert d.nn_grid_ptr() is not None
    v = d.vertices[0].cpu().numpy()
    rng = np.random.RandomState(12)
    n = 60011
    pts = (v[rng.randint(0, len(v), n)] + rng.randn(n, 3).astype(np.float32) * 0.04).astype(np.float32)
    pts[:2000] = v[rng.randint(0, len(v), 2000)]                                   # exactly on vertices
    pts[2000:4000] = rng.uniform(-3, 3, (2000, 3)).astype(np.float32)              # mostly far from the body
    lo, hi = v.min(0), v.max(0)
    pts[4000:5000] = (lo + (hi - lo) * rng.randint(0, 2, (1000, 3))) + rng.randn(1000, 3).astype(np.float32) * 0.06   # around the box corners / faces
    pts[5000] = [np.nan, 0, 0]; pts[5001] = [np.inf, 0, 0]; pts[5002] = [1e30, -1e30, 0]
    x = torch.as_tensor(pts, device=DEV)
    res = {}
    for use in (True, False):
        d.use_nn_grid = use
        try:
            with torch.no_grad():
                rgb, sig = d(x, net, eval_mode=True)
            P, V = n, d.vertices.shape[1]
            out = dict(cand=torch.empty((P, 3), device=DEV), cand_pt=torch.empty(P, dtype=torch.int32, device=DEV), idx=torch.empty(P, dtype=torch.int32, device=DEV),
                       off=torch.empty(P, dtype=torch.int32, device=DEV), cnt=torch.empty(P, dtype=torch.uint8, device=DEV), n=torch.empty(1, dtype=torch.int32, device=DEV))
            _lib.check(_lib.lib().ia_smpl_nn_compact(_lib.ptr(x), P, None, _lib.ptr(d.vertices), _lib.ptr(d.T_inv), V, float(d.threshold), _lib.ptr(out["cand"]),
                                                     _lib.ptr(out["cand_pt"]), _lib.ptr(out["idx"]), _lib.ptr(out["off"]), _lib.ptr(out["cnt"]), _lib.ptr(out["n"]),
                                                     d.nn_grid_ptr(), _lib.stream()), "ia_smpl_nn_compact")
            assert (d.nn_grid_ptr() is not None) == use
        finally:
            d.use_nn_grid = True
        res[use] = (rgb.cpu(), sig.cpu(), {k: t.cpu().numpy() for k, t in out.items()})
    (r1, s1, o1), (r0, s0, o0) = res[True], res[False]
    assert torch.equal(r1, r0) and torch.equal(s1, s0)
    valid = o0["cnt"].astype(bool)
    assert np.array_equal(o1["cnt"], o0["cnt"]) and int(o1["n"][0]) == int(o0["n"][0]) == int(valid.sum())
    assert 0.3 < valid.mean() < 0.95 and not valid[5000:5003].any()
    assert np.array_equal(o1["idx"][valid], o0["idx"][valid]) and (o1["idx"][~valid] == -1).all()
    assert np.array_equal(o1["cand"][o1["off"][valid]], o0["cand"][o0["off"][valid]])
    assert np.array_equal(o1["cand_pt"][o1["off"][valid]], np.flatnonzero(valid))
    # ... and the brute-force reference of both is the oracle's nearest-vertex search
    _, valid_o, idx_o = oracle.smpl_nn_deform(pts, v, d.T_inv[0].cpu().numpy(), d.threshold)
    assert np.array_equal(valid_o, valid) and np.array_equal(idx_o[valid], o1["idx"][valid])


def test_rendered_frame_matches_oracle(oracle, sw):
    """render_image_fast with the SMPLDeformer plugin (occupancy build + wave-front loop through the
    renderer's generic closure route) against the oracle: rgb / alpha within 1e-3."""
    model, body, fp, poses, tr = sw
    res, G = 64, 64
    i = 2
    jit = np.random.RandomState(31).rand(2, G ** 3, 3).astype(np.float32)
    batch = make_batch(DEV, res, poses[i], tr[i], betas=BETAS)
    rgb, depth, alpha, counter = model.render_image_fast(batch, (res, res), jitter=torch.as_tensor(jit, device=DEV))
    d = model.deformer
    prep = oracle.smpl_deformer_prepare(body, BETAS, poses[i][3:], poses[i][:3], tr[i])
    prep = dict(prep, vertices=d.vertices[0].cpu().numpy(), T_inv=d.T_inv[0].cpu().numpy())
    field, keep = oracle.make_field(fp)
    query = lambda p: oracle.smpl_deform_query(p, prep, field, eval_mode=True)
    aabb = oracle.get_bbox_from_smpl(prep["vertices"])
    idx = np.arange(G, dtype=np.float32)
    cx, cy, cz = np.meshgrid(idx, idx, idx, indexing="ij")
    coords0 = (np.stack([cx, cy, cz], -1).reshape(-1, 3) / np.float32(G)).astype(np.float32)
    density = np.zeros(G ** 3, np.float32)
    for it in range(len(jit)):
        coords = (coords0 + jit[it] / np.float32(G)) * (aabb[1] - aabb[0]) + aabb[0]
        density = np.maximum(density, query(coords.astype(np.float32))[1])
    occ = oracle.occupancy_from_density(density, G)
    ro, rd = syn.make_camera_rays(res)
    o, dd, near, far = oracle.transform_rays_w2s(ro, rd, prep["w2s"])
    ref = oracle.render_test(o, dd, near, far, occ, aabb, query)
    occ_g = model.renderer.density_grid_test.density_field.cpu().numpy()
    W.cells_within(occ_g, occ.astype(bool), "SMPLDeformer frame occupancy")
    rgb, alpha = rgb.reshape(-1, 3).cpu().numpy(), alpha.reshape(-1).cpu().numpy()
    assert (ref["alpha"] > 0.5).mean() > 0.02
    err_rgb, err_a = np.abs(rgb - ref["rgb"]).max(1), np.abs(alpha - ref["alpha"])
    W.rays_within(err_rgb, "SMPLDeformer frame rgb")
    W.rays_within(err_a, "SMPLDeformer frame alpha")
