"""CPU suite (-m "not gpu"): the oracle against the golden vectors, host logic,
and that the C-ABI library loads and exports every declared symbol."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest
import torch

from instantavatar_amd import synthetic as syn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


# ---------------------------------------------------------------- C ABI / build
#: goldens that involve the field exist once per hash-grid layout (tcnn level-3 resolution 54 / 55)
_LAYOUT = "_l55" if os.environ.get("IA_TCNN_LEVEL3_RES", "54") == "55" else ""


def test_cabi_exports_every_declared_symbol():
    from instantavatar_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "instantavatar_hip.h")).read()
    declared = set(re.findall(r"\b(ia_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    l = C.CDLL(_lib.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(l, s)]
    assert not missing, "symbols declared in include/*.h but not exported: %s" % missing
    assert set(_lib.EXPORTED) <= declared
    assert _lib.lib().ia_version() >= 100


def test_library_manifest_equals_checkout_and_names_device_code():
    """The library says what it was built from (ia_source_manifest): the source hashes must be the checkout's (a stale
    prebuilt .so is an error, build.py), and every translation unit with kernels carries a device-code hash."""
    from instantavatar_amd import _lib, build as ia_build
    assert ia_build.library_manifest() == ia_build.source_manifest(), "libinstantavatar_hip.so is stale: run __graft_entry__.build()"
    assert not ia_build.needs_build()
    dev = ia_build.device_manifest()
    for tu in ("ia_snarf.hip", "ia_field.hip", "ia_render.hip"):
        assert re.fullmatch(r"[0-9a-f]{16}", dev[tu]), dev
    # the loaded library reports the same string the file carries
    m = _lib.lib().ia_source_manifest().decode()
    assert dict(kv.split("=") for kv in m.split(";")) == ia_build._raw_manifest()


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_rebuild_on_another_path_reproduces_the_device_code(tmp_path):
    """the counter summaries under profiles/ are keyed on the device code, and a forced rebuild of
    the same sources in ANOTHER directory must give the very same hashes (the .so bytes used to differ through the
    path-derived __hip_cuid_*; build.py now names the cuid after the file)."""
    import shutil
    import subprocess
    from instantavatar_amd import build as ia_build
    os.makedirs(tmp_path / "elsewhere" / "instantavatar_amd")
    shutil.copytree(os.path.join(ROOT, "include"), tmp_path / "elsewhere" / "include")
    shutil.copytree(os.path.join(ROOT, "instantavatar_amd", "csrc"), tmp_path / "elsewhere" / "instantavatar_amd" / "csrc",
                    ignore=shutil.ignore_patterns("*.o", ".stamps"))
    shutil.copy(os.path.join(ROOT, "instantavatar_amd", "build.py"), tmp_path / "elsewhere" / "instantavatar_amd" / "build.py")
    subprocess.check_call([sys.executable, "instantavatar_amd/build.py", "--force"], cwd=tmp_path / "elsewhere",
                          stdout=subprocess.DEVNULL)
    other = str(tmp_path / "elsewhere" / "instantavatar_amd" / "libinstantavatar_hip.so")
    assert ia_build.device_manifest(other) == ia_build.device_manifest()
    assert ia_build.library_manifest(other) == ia_build.library_manifest()


def test_bench_quotes_a_counter_summary_only_for_the_device_code_it_runs(tmp_path, monkeypatch):
    import json
    import bench
    from instantavatar_amd import build as ia_build
    dev = ia_build.device_manifest()
    os.makedirs(tmp_path / "profiles")
    json.dump({"device_code": dev, "x": 1}, open(tmp_path / "profiles" / "r04_pmc_search.json", "w"))
    json.dump({"device_code": dict(dev, **{"ia_field.hip": "0" * 16}), "x": 2}, open(tmp_path / "profiles" / "r04_pmc_encode.json", "w"))
    json.dump({"so_sha256": "abc", "x": 3}, open(tmp_path / "profiles" / "r03_pmc_mfma.json", "w"))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    j, src = bench._profile_json("pmc_search", ("ia_snarf.hip",))
    assert j["x"] == 1 and "r04_pmc_search.json" in src
    j, src = bench._profile_json("pmc_encode", ("ia_field.hip",))
    assert j is None and "other device code of ia_field.hip" in src
    j, src = bench._profile_json("pmc_encode", ("ia_snarf.hip",))     # the changed unit is not the one this summary needs
    assert j["x"] == 2
    j, src = bench._profile_json("pmc_mfma", ("ia_field.hip",))
    assert j is None and "no device-code hashes" in src
    j, src = bench._profile_json("pmc_hgbwd", ("ia_field.hip",))
    assert j is None and "no PMC summary" in src


def test_ensure_current_rebuilds_a_stale_library_and_only_then(tmp_path, monkeypatch):
    """`build.ensure_current()` (called by conftest, bench.py and __graft_entry__.smoke before anything loads the library): nothing
    happens for a current library or while an A/B variant is run on purpose; a library that does not match the checkout is rebuilt
    when the compiler is there -- the product then runs the HIP library of THIS checkout, never an older one and never anything else."""
    from instantavatar_amd import build
    assert not build.needs_build() and build.ensure_current() is False
    calls = []
    monkeypatch.setattr(build, "needs_build", lambda: True)
    monkeypatch.setattr(build, "build", lambda force=False, verbose=False: calls.append(force) or build.OUT)
    monkeypatch.setenv("IA_ALLOW_STALE_LIB", "1")
    assert build.ensure_current() is False and calls == []          # a variant run: hands off
    monkeypatch.delenv("IA_ALLOW_STALE_LIB")
    monkeypatch.setattr(build, "have_compiler", lambda: False)
    assert build.ensure_current() is False and calls == []          # no compiler: `_lib.lib()` will refuse the stale library
    monkeypatch.setattr(build, "have_compiler", lambda: True)
    assert build.ensure_current() is True and calls == [False]      # stale + compiler: rebuilt (what changed only)


def test_no_cpu_fallback_fails_loudly():
    from instantavatar_amd import _lib
    from instantavatar_amd.models.networks.ngp import NeRFNGPNet
    net = NeRFNGPNet(dict(center=[0, 0, 0], scale=[1, 1, 1]), n_levels=8, log2_hashmap_size=12)
    with pytest.raises(_lib.IAError):
        with torch.no_grad():
            net(torch.zeros(4, 3), None)  # CPU tensor: must raise, not fall back


def test_product_does_not_import_oracle():
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "instantavatar_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b|libia_oracle|orc_[a-z_]+\(", src, re.M):
                    bad.append(f)
    assert not bad, "product files reference the oracle: %s" % bad


# ---------------------------------------------------------------- half helpers
def test_half_conversion_matches_ieee(oracle):
    rng = np.random.RandomState(0)
    x = np.concatenate([rng.randn(20000).astype(np.float32) * s for s in (1e-8, 1e-5, 1e-3, 1, 100, 7e4)])
    x = np.concatenate([x, np.array([0, -0.0, np.inf, -np.inf, 65504, 65520, 65519.99, 5.96e-8, 2.98e-8, 2.99e-8], np.float32)])
    y = np.empty(len(x), np.uint16)
    oracle.lib().orc_f32_to_f16(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), C.c_long(len(x)))
    ref = x.astype(np.float16).view(np.uint16)
    assert np.array_equal(y, ref)
    allh = np.arange(65536, dtype=np.uint16)
    back = np.empty(65536, np.float32)
    oracle.lib().orc_f16_to_f32(allh.ctypes.data_as(C.c_void_p), back.ctypes.data_as(C.c_void_p), C.c_long(65536))
    ref = allh.view(np.float16).astype(np.float32)
    ok = (back == ref) | (np.isnan(back) & np.isnan(ref))
    assert ok.all()


# ---------------------------------------------------------------- tcnn level table
def test_hash_level_table_pinned(oracle):
    """tcnn-v1.6 level table with glibc float32 arithmetic (parity unpinned vs tcnn itself;
    this pins OUR three implementations to each other and to the frozen numbers)."""
    from instantavatar_amd import _lib
    hd_o = oracle.hash_desc()
    hd_p = _lib.make_hash_desc()
    sc, res, off = syn.hash_level_table()
    r3 = int(os.environ.get("IA_TCNN_LEVEL3_RES", 54))   # the suite can be run under either level-3 layout
    expect_res = [16, 24, 36, r3, 81, 122, 183, 274, 411, 616, 923, 1384, 2076, 3114, 4671, 7007]
    assert list(hd_o.res) == expect_res == list(hd_p.res) == list(res)
    assert list(hd_o.offset) == list(hd_p.offset) == list(off)
    assert hd_o.offset[16] == (6513496 if r3 == 54 else 6522408)  # => encoder.params = 3072 + 13026992 (54) / 13044816 (55)
    assert np.allclose(np.array(hd_o.scale[:]), sc, rtol=0, atol=0)
    assert list(hd_o.offset[:5]) == [0, 4096, 17920, 64576, 222040 if r3 == 54 else 230952]


# ---------------------------------------------------------------- LBS golden (reference lbs.py)
def test_lbs_oracle_matches_reference_golden(oracle):
    g = np.load(os.path.join(HERE, "golden", "lbs_golden.npz"))
    body = dict(v_template=g["v_template"], shapedirs=g["shapedirs"], posedirs=g["posedirs"],
                J_regressor=g["J_regressor"], lbs_weights=g["lbs_weights"], parents=g["parents"])
    for i in range(int(g["n_cases"])):
        pose = g["pose%d" % i][0]
        out = oracle.smpl_forward(body, g["betas%d" % i], pose[3:], pose[:3], g["transl%d" % i])
        assert np.abs(out["A"] - g["A%d" % i][0]).max() < 2e-5
        assert np.abs(out["vertices"] - g["verts%d" % i][0]).max() < 2e-5
        assert np.abs(out["joints"] - g["joints%d" % i][0]).max() < 2e-5
        assert np.abs(oracle.batch_rodrigues(pose.reshape(-1, 3)) - g["rot%d" % i]).max() < 1e-6
        assert np.abs(out["T"] - g["T%d" % i][0]).max() < 2e-5                      # consumed by SMPLDeformer
        assert np.abs(out["shape_offsets"] - g["shape_offsets%d" % i][0]).max() < 1e-6
        assert np.abs(out["pose_offsets"] - g["pose_offsets%d" % i][0]).max() < 1e-6


def test_lbs_product_torch_matches_reference_golden():
    from instantavatar_amd.deformers.smplx import SMPL
    g = np.load(os.path.join(HERE, "golden", "lbs_golden.npz"))
    smpl = SMPL.from_dict(dict(v_template=g["v_template"], shapedirs=g["shapedirs"], posedirs=g["posedirs"],
                               J_regressor=g["J_regressor"], lbs_weights=g["lbs_weights"], parents=g["parents"]))
    for i in range(int(g["n_cases"])):
        pose = torch.as_tensor(g["pose%d" % i])
        for small_ops in (False, True):     # library GEMMs (initialisation) / broadcast multiply + sum (the per-step SMPLDeformer path)
            out = smpl(torch.as_tensor(g["betas%d" % i]), pose[:, 3:], pose[:, :3], torch.as_tensor(g["transl%d" % i]), small_ops=small_ops)
            assert (out.A - torch.as_tensor(g["A%d" % i])).abs().max() < 2e-5
            assert (out.vertices - torch.as_tensor(g["verts%d" % i])).abs().max() < 2e-5
            assert (out.T - torch.as_tensor(g["T%d" % i])).abs().max() < 2e-5
            assert (out.shape_offsets - torch.as_tensor(g["shape_offsets%d" % i])).abs().max() < 1e-6
            assert (out.pose_offsets - torch.as_tensor(g["pose_offsets%d" % i])).abs().max() < 1e-6


# ---------------------------------------------------------------- oracle self-consistency
@pytest.fixture(scope="module")
def small_world(oracle):
    body = syn.make_body()
    init = oracle.deformer_initialize(body, np.zeros(10, np.float32), syn.cano_pose("A_pose"), resolution=32, n_smooth=30)
    fp = syn.make_field(init["cano_joints"], init["bbox"])
    poses, tr = syn.procedural_pose_track(8)
    world = oracle.make_world(body, init, fp, np.zeros(10, np.float32), poses[1, 3:], poses[1, :3], tr[1], syn.INIT_BONES)
    return body, init, fp, world


@pytest.fixture(scope="module")
def small_world_blend(oracle):
    """the same world on the blend-shape body (non-zero shapedirs / posedirs, dense J_regressor) with synthetic.BLEND_BETAS:
    what a real SMPL pickle + a data set's betas are (VERDICT r05 missing 3)"""
    body = syn.make_body(blendshapes=True)
    init = oracle.deformer_initialize(body, syn.BLEND_BETAS, syn.cano_pose("A_pose"), resolution=32, n_smooth=30)
    fp = syn.make_field(init["cano_joints"], init["bbox"])
    poses, tr = syn.procedural_pose_track(8)
    world = oracle.make_world(body, init, fp, syn.BLEND_BETAS, poses[1, 3:], poses[1, :3], tr[1], syn.INIT_BONES)
    return body, init, fp, world


def test_mlp_half_accumulate_mode_sizes_the_tcnn_deviation(oracle, small_world):
    """DESIGN.md section 2's one documented deviation from tiny-cuda-nn v1.6: fp32 MLP accumulators here, __half wmma
    accumulators there.  The oracle can run both (`set_mlp_half_accumulate`): the default mode is untouched, the half mode
    rounds the running sum after every 16-wide k block, and on the synthetic field the two differ by a few half ulps of
    the activations -- the per-sample size of the deviation (its per-RAY size on the 512^2 bench frame is measured by
    tools/size_mlp_accumulation.py on the GPU box and quoted in DESIGN.md)."""
    body, init, fp, world = small_world
    rng = np.random.RandomState(3)
    bb = init["bbox"]
    x = (rng.rand(20000, 3).astype(np.float32) * (bb[1] - bb[0]) + bb[0]).astype(np.float32)
    assert oracle.set_mlp_half_accumulate(False) is False
    rgb0, sig0 = oracle.field_fwd(world["field"], x)
    rgb0b, sig0b = oracle.field_fwd(world["field"], x)
    assert np.array_equal(rgb0, rgb0b) and np.array_equal(sig0, sig0b)
    try:
        assert oracle.set_mlp_half_accumulate(True) is False
        rgb1, sig1 = oracle.field_fwd(world["field"], x)
    finally:
        assert oracle.set_mlp_half_accumulate(False) is True
    rgb2, sig2 = oracle.field_fwd(world["field"], x)
    assert np.array_equal(rgb0, rgb2) and np.array_equal(sig0, sig2)        # the switch leaves nothing behind
    d_rgb, d_sig = np.abs(rgb1 - rgb0).max(1), np.abs(sig1 - sig0)
    rel_sig = d_sig / np.maximum(np.abs(sig0), 1.0)
    print("half vs fp32 accumulate: rgb differs on %.3f of the samples (max %.2e, > 1e-3 on %.4f); sigma rel max %.2e, > 1e-3 on %.4f" % (
        (d_rgb > 0).mean(), d_rgb.max(), (d_rgb > 1e-3).mean(), rel_sig.max(), (rel_sig > 1e-3).mean()))
    assert (d_rgb > 0).any() or (d_sig > 0).any(), "the two modes must not be the same arithmetic"
    assert d_rgb.max() < 2e-2 and rel_sig.max() < 5e-2      # a few half ulps (2^-11 relative) through three / two layers


def test_weight_voxels_are_a_partition_of_unity(small_world):
    _, init, _, _ = small_world
    w = init["lbs_voxel"]
    assert w.shape == (24, 8, 32, 32)
    assert np.abs(w.sum(0) - 1).max() < 1e-5 and w.min() >= 0


def test_broyden_roots_satisfy_forward_skinning(oracle, small_world):
    """Converged roots x_c must map back to x_d under d(x) = J(x) [x,1] within cvg."""
    body, init, fp, world = small_world
    rng = np.random.RandomState(1)
    vd = world["voxel_d"].reshape(3, -1)
    pts = (vd[:, rng.randint(0, vd.shape[1], 4000)].T + rng.randn(4000, 3).astype(np.float32) * 0.01).astype(np.float32)
    x, Ji, valid, iters = oracle.broyden(pts, world["voxel_J"], world["tfs"], init, world["bone_ids"], want_iters=True)
    assert valid.any() and iters.min() >= 2 and iters.max() <= 11
    # identity pose: every valid root equals the query point
    w_id = oracle.make_world(body, init, fp, np.zeros(10, np.float32), syn.cano_pose("A_pose"), np.zeros(3, np.float32),
                             np.zeros(3, np.float32), syn.INIT_BONES)
    assert np.abs(w_id["tfs"] - np.eye(4)).max() < 1e-5
    x2, _, v2 = oracle.broyden(pts, w_id["voxel_J"], w_id["tfs"], init, w_id["bone_ids"])
    m = v2.astype(bool)
    assert m.any()
    assert np.abs(x2[m] - np.repeat(pts[:, None], 13, 1)[m]).max() < 1e-4
    # filter: at most one survivor per cluster; survivors are a subset of valid
    keep = oracle.filter_dup(x2, v2)
    assert (keep <= v2).all() and (keep.sum(1) <= 1 + 0 * keep.sum(1)).all()


def test_filter_keeps_last_duplicate(oracle):
    x = np.zeros((1, 13, 3), np.float32)
    mask = np.zeros((1, 13), np.uint8)
    x[0, 2] = [0.1, 0.2, 0.3]; x[0, 7] = [0.1, 0.2, 0.30005]; x[0, 9] = [0.5, 0, 0]
    mask[0, [2, 7, 9]] = 1
    out = oracle.filter_dup(x, mask)
    assert out[0].tolist() == [0, 0, 0, 0, 0, 0, 0, 1, 0, 1, 0, 0, 0]  # Q6: the LAST of a cluster survives


def test_occupancy_keeps_largest_component(oracle):
    G = 16
    dens = np.zeros((G, G, G), np.float32)
    dens[2:6, 2:6, 2:6] = 500.0     # big blob
    dens[12:13, 12:13, 12:13] = 500.0  # small blob, not 26-connected to the big one after max-pool
    occ = oracle.occupancy_from_density(dens, G)
    assert occ[3, 3, 3] and not occ[12, 12, 12]
    assert occ.sum() == 6 ** 3  # 4^3 dilated by the 3^3 max-pool
    assert oracle.occupancy_from_density(np.zeros((G, G, G), np.float32), G).sum() == 0


def test_composite_threshold_and_early_stop(oracle):
    n, S = 1, 6
    rgb = np.ones((n, S, 3), np.float32)
    sigma = np.array([[0.5, 100, 1e4, 50, 50, 50]], np.float32)
    delta = np.full((n, S), 2 / 256, np.float32); delta[0, 5] = 0
    depth = np.arange(S, dtype=np.float32)[None] + 1
    alive = np.zeros(1, np.int64)
    color = np.zeros((1, 3), np.float32); dep = np.zeros(1, np.float32); nohit = np.ones(1, np.float32)
    oracle.lib().orc_composite_test(*[a.ctypes.data_as(C.c_void_p) for a in (rgb, sigma, delta, depth, alive)],
                                    C.c_long(1), S, color.ctypes.data_as(C.c_void_p), dep.ctypes.data_as(C.c_void_p),
                                    nohit.ctypes.data_as(C.c_void_p), C.c_float(0.01))
    a1 = 1 - np.exp(-100 * 2 / 256)          # sample 0 (alpha<0.01) skipped without attenuation
    T = (1 - a1) * np.exp(-1e4 * 2 / 256)    # sample 2 drives T below 1e-4 -> loop stops
    assert abs(nohit[0] - T) < 1e-7 and T < 1e-4
    assert abs(color[0, 0] - (a1 + (1 - a1) * (1 - np.exp(-1e4 * 2 / 256)))) < 1e-6


def test_render_frame_small(oracle, small_world):
    body, init, fp, world = small_world
    ro, rd = syn.make_camera_rays(32)
    jit = np.random.RandomState(3).rand(2, 64 ** 3, 3).astype(np.float32)
    out = oracle.render_image_fast(world, ro, rd, jit)
    cov = (out["alpha"] > 0.5).mean()
    assert 0.02 < cov < 0.5 and out["occ"].sum() > 100
    assert np.isfinite(out["rgb"]).all() and out["rgb"].min() >= 0 and out["rgb"].max() <= 1 + 1e-5
    # linearity of the background term (Q12): rgb(bg) - rgb(white) = T * (bg - 1)
    o, d, near, far = oracle.transform_rays_w2s(ro, rd, world["w2s"])
    bg = np.zeros((len(ro), 3), np.float32)
    out2 = oracle.render_test(o, d, near, far, out["occ"], out["aabb"], lambda p: oracle.deform_query(p, world, True), bg=bg)
    assert np.abs((out["rgb"] - out2["rgb"]) - (1 - out["alpha"])[:, None]).max() < 1e-6


def test_smpl_deformer_oracle_properties(oracle):
    """SMPLDeformer restatement (smpl_deformer.py:32-110): T_inv of vertex v maps the posed vertex v
    (SMPL-root frame) onto the template vertex v; the nearest-vertex search equals numpy brute force."""
    body = syn.make_body(42)
    poses, tr = syn.procedural_pose_track(8)
    pose, transl = poses[3], tr[3]
    prep = oracle.smpl_deformer_prepare(body, np.zeros(10, np.float32), pose[3:], pose[:3], transl)
    pose_t = np.zeros((1, 69), np.float32)
    pose_t[:, 2], pose_t[:, 5] = np.pi / 6, -np.pi / 6
    tmpl = oracle.smpl_forward(body, np.zeros(10, np.float32), pose_t)["vertices"]
    v = prep["vertices"]
    back = np.einsum("vij,vj->vi", prep["T_inv"][:, :3, :3], v) + prep["T_inv"][:, :3, 3]
    assert np.abs(back - tmpl).max() < 2e-5
    rng = np.random.RandomState(4)
    pts = (v[rng.randint(0, len(v), 3000)] + rng.randn(3000, 3).astype(np.float32) * 0.04).astype(np.float32)
    cano, valid, idx = oracle.smpl_nn_deform(pts, v, prep["T_inv"], 0.05)
    d2 = ((pts[:, None, :] - v[None]) ** 2).sum(-1)
    assert (d2.argmin(1) == idx).mean() > 0.999          # ties / last-bit order only
    near = np.abs(d2.min(1) - 0.05 ** 2) > 1e-6
    assert ((d2.min(1) < 0.05 ** 2) == valid)[near].all()
    assert 0.2 < valid.mean() < 0.95
    ref = np.einsum("pij,pj->pi", prep["T_inv"][idx, :3, :3], pts) + prep["T_inv"][idx, :3, 3]
    assert np.abs(cano - ref).max() < 1e-5


def test_animate_and_eval_writers_agree_on_the_channel_order(tmp_path):
    """the model's channels are in cv2's (B, G, R) order; animate / novel_view frames (cv2.imwrite of BGRA in the
    reference, animate.py:113) and eval's panels (cv2.imwrite of BGR, evaluation.write_png_bgr) must put the same colour
    into the file."""
    from PIL import Image
    from instantavatar_amd import evaluation as ev
    from instantavatar_amd.drivers import animate
    frame = np.zeros((4, 6, 4), np.uint8)
    frame[..., 0], frame[..., 1], frame[..., 2], frame[..., 3] = 10, 20, 30, 255     # B = 10, G = 20, R = 30
    animate.write_frames([frame, frame], str(tmp_path / "a"), gif="x.gif")
    a = np.asarray(Image.open(tmp_path / "a" / "0.png"))
    assert a.shape == (4, 6, 4) and tuple(a[0, 0]) == (30, 20, 10, 255)           # what cv2.imwrite would have stored
    ev.write_png_bgr(str(tmp_path / "e.png"), torch.as_tensor(frame[..., :3].astype(np.float32) / 255))
    e = np.asarray(Image.open(tmp_path / "e.png"))
    assert np.array_equal(e, a[..., :3])
    g = Image.open(tmp_path / "a" / "x.gif")
    assert g.n_frames == 2 or getattr(g, "n_frames", 1) >= 1                        # identical frames may be merged by the encoder
    assert tuple(np.asarray(g.convert("RGB"))[0, 0]) == (30, 20, 10)


def test_checkpoint_layout_switch_keeps_the_parameter_an_optimizer_holds(tmp_path):
    """drivers/train.py --resume builds optimiser and scheduler BEFORE load_checkpoint; when the
    checkpoint has the other tcnn level-3 layout the hash-table parameter must be resized IN PLACE (same Parameter object),
    stale moments dropped, and the checkpoint's own optimiser state restored -- the optimiser then keeps training the live
    table."""
    from instantavatar_amd.drivers.checkpoint import load_checkpoint, save_checkpoint
    from instantavatar_amd.models.networks.ngp import NeRFNGPNet

    class M(torch.nn.Module):
        def __init__(self, r3):
            super().__init__()
            self.net_coarse = NeRFNGPNet(dict(center=[0, 0, 0], scale=[1, 1, 1]), n_levels=5, log2_hashmap_size=19, level3_res=r3)
    src, dst = M(55), M(54)
    assert src.net_coarse.encoder.params.numel() != dst.net_coarse.encoder.params.numel()
    so = torch.optim.Adam(src.parameters(), lr=1e-2)
    for p in src.parameters():
        p.grad = torch.full_like(p, 0.5)
    so.step()
    save_checkpoint(src, str(tmp_path / "c.ckpt"), optimizer=so)
    p = dst.net_coarse.encoder.params
    opt = torch.optim.Adam(dst.parameters(), lr=1e-2)
    for q in dst.parameters():
        q.grad = torch.ones_like(q)
    opt.step()                                       # the optimiser now holds moments of the OLD shape
    load_checkpoint(dst, str(tmp_path / "c.ckpt"), optimizer=opt)
    assert dst.net_coarse.encoder.params is p and p.numel() == src.net_coarse.encoder.params.numel()
    assert torch.equal(p.detach(), src.net_coarse.encoder.params.detach())
    assert opt.state[p]["exp_avg"].shape == p.shape  # the checkpoint's moments, new shape
    p.grad = torch.ones_like(p)
    before = p.detach().clone()
    opt.step()
    assert (p.detach() != before).all()              # the LIVE table moves


def _write_smpl_pickle(path, body, chumpy_key=None):
    """the synthetic body in the key layout of the licensed SMPL_<GENDER>.pkl (smplx/body_models.py:108-154 reads
    v_template, shapedirs [V,3,>=10], posedirs [V,3,207], a scipy-sparse J_regressor [24,V], kintree_table [2,24] with
    uint32(-1) as the root's parent, weights [V,24], f [F,3])"""
    import pickle
    import scipy.sparse as sp
    V = body["v_template"].shape[0]
    posedirs = np.asarray(body["posedirs"], np.float64)
    if posedirs.ndim == 2:                      # [207, V*3] -> the file's [V, 3, 207]
        posedirs = posedirs.T.reshape(V, 3, -1)
    parents = np.asarray(body["parents"]).astype(np.int64)
    kt = np.stack([parents, np.arange(24)]).astype(np.uint32)      # kt[0, 0] = 4294967295
    d = {"v_template": np.asarray(body["v_template"], np.float64), "shapedirs": np.asarray(body["shapedirs"], np.float64),
         "posedirs": posedirs, "J_regressor": sp.csc_matrix(np.asarray(body["J_regressor"], np.float64)),
         "kintree_table": kt, "weights": np.asarray(body["lbs_weights"], np.float64), "f": np.asarray(body["f"]).astype(np.uint32),
         "bs_type": "lrotmin", "bs_style": "lbs", "J": np.zeros((24, 3)), "vert_sym_idxs": np.arange(V)}
    if chumpy_key:
        import sys
        import types
        mod = types.ModuleType("chumpy"); sub = types.ModuleType("chumpy.ch")
        Ch = type("Ch", (), {"__module__": "chumpy.ch", "__init__": lambda self, x: setattr(self, "x", x)})
        sub.Ch = Ch; mod.ch = sub
        sys.modules["chumpy"], sys.modules["chumpy.ch"] = mod, sub
        try:
            d[chumpy_key] = Ch(d[chumpy_key])
            with open(path, "wb") as f:
                pickle.dump(d, f, protocol=2)
        finally:
            del sys.modules["chumpy"], sys.modules["chumpy.ch"]
        return
    with open(path, "wb") as f:
        pickle.dump(d, f, protocol=2)


def test_smpl_loads_the_licensed_pickle_layout_and_names_chumpy(tmp_path):
    """VERDICT r03 missing 4: the first real run goes through SMPL._load (model_path + gender -> SMPL_<GENDER>.pkl with
    kintree_table / weights / posedirs [V,3,207] / sparse J_regressor), which no test executed.  A pickle written in that
    layout from the synthetic body must give the same module as SMPL.from_dict(body); a pickle that needs chumpy to
    unpickle gets an error that says so and how to convert it."""
    from instantavatar_amd.deformers.smplx import SMPL
    body = syn.make_body(42)
    body = dict(body)
    body.setdefault("f", np.stack([np.arange(0, 300), np.arange(1, 301), np.arange(2, 302)], 1))
    d = tmp_path / "smpl"
    d.mkdir()
    _write_smpl_pickle(str(d / "SMPL_NEUTRAL.pkl"), body)
    a = SMPL(str(d), "neutral")                      # directory + gender (confs/deformer/fast_snarf.yaml: model_path)
    b = SMPL(str(d / "SMPL_NEUTRAL.pkl"), "neutral")  # or the file itself
    ref = SMPL.from_dict(body)
    for m in (a, b):
        for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "parents", "faces_tensor"):
            assert torch.equal(getattr(m, k), getattr(ref, k)), k
        assert m.parents_list[0] == -1 and m.posedirs.shape == (207, 6890 * 3)
    betas = torch.zeros(1, 10)
    pose = torch.as_tensor(syn.procedural_pose_track(4)[0][2][None])
    oa, orf = a(betas=betas, body_pose=pose[:, 3:], global_orient=pose[:, :3]), ref(betas=betas, body_pose=pose[:, 3:], global_orient=pose[:, :3])
    assert torch.equal(oa.vertices, orf.vertices) and torch.equal(oa.A, orf.A)
    with pytest.raises(FileNotFoundError):
        SMPL(str(d), "female")
    c = tmp_path / "ch"
    c.mkdir()
    _write_smpl_pickle(str(c / "SMPL_NEUTRAL.pkl"), body, chumpy_key="v_template")
    with pytest.raises(ImportError) as e:
        SMPL(str(c), "neutral")
    assert "chumpy" in str(e.value) and "npz" in str(e.value)
    # the conversion the message describes: an .npz next to the .pkl is picked up first
    np.savez(str(c / "SMPL_NEUTRAL.npz"), v_template=body["v_template"], shapedirs=body["shapedirs"], posedirs=body["posedirs"],
             J_regressor=body["J_regressor"], kintree_table=np.stack([np.asarray(body["parents"]), np.arange(24)]).astype(np.int64),
             weights=body["lbs_weights"], f=body["f"])
    n = SMPL(str(c), "neutral")
    assert torch.equal(n.v_template, ref.v_template) and torch.equal(n.lbs_weights, ref.lbs_weights)


def test_driver_config_and_checkpoint_io(tmp_path):
    """Hydra-free config loading (confs/ groups, ${...} interpolation, _target_ instantiation) and the
    Lightning-layout checkpoint round trip of the PL-free drivers."""
    from instantavatar_amd.drivers import checkpoint as ck, config as cfg
    from instantavatar_amd.pipeline import AvatarModel
    confs = os.path.join(ROOT, "confs")
    d = cfg.load_group(confs, "deformer", "fast_snarf", {"dataset": {"gender": "female"}, "train": {"precision": 32}})
    assert d["gender"] == "female" and d["opt"]["precision"] == 32 and d["opt"]["resolution"] == 128
    assert cfg.resolve("out/${a.b}/x${a.c}", {"a": {"b": "s", "c": 3}}) == "out/s/x3"
    # every deformer group the reference ships (confs/deformer/{fast_snarf,fast_snarf_debug,smpl}.yaml), re-pointed `_target_`
    body_model = None
    from instantavatar_amd.deformers.smplx import SMPL
    body_model = SMPL.from_dict(syn.make_body())
    for name, cls, version in (("fast_snarf", "SNARFDeformer", 1), ("fast_snarf_debug", "SNARFDeformer", 2), ("smpl", "SMPLDeformer", None)):
        dd = cfg.load_group(confs, "deformer", name, {"dataset": {"gender": "male"}, "train": {"precision": 32}})
        assert dd["_target_"].startswith("instantavatar_amd.deformers.") and dd["gender"] == "male"
        obj = cfg.instantiate(dd, body_model=body_model)            # (a pre-built body model: the SMPL pickles are not shipped)
        assert type(obj).__name__ == cls
        if version is not None:
            assert dd["opt"].get("version", 1) == version and obj.deformer.version == version      # deformer_torch.py:32 `opt.get("version", 1)`
    net = cfg.instantiate(cfg.load_group(confs, "network", "ngp", {}))
    ren = cfg.instantiate(cfg.load_group(confs, "renderer", "raymarcher_acc", {}))
    assert type(net).__name__ == "NeRFNGPNet" and ren.MAX_BATCH_SIZE == 291600
    ren.initialize(1)
    model = AvatarModel(None, net, ren)
    model.global_step = 1234
    with torch.no_grad():
        net.encoder.params[:100] = torch.arange(100.0)
    path = ck.save_checkpoint(model, str(tmp_path / "last.ckpt"), epoch=3)
    sd = torch.load(path, weights_only=False)
    assert set(sd) >= {"state_dict", "global_step", "epoch"} and sd["global_step"] == 1234
    assert "net_coarse.encoder.params" in sd["state_dict"] and "net_coarse.color_net.params" in sd["state_dict"]
    sd["state_dict"]["loss_fn.lpips.net.weight"] = torch.zeros(3)         # a module that is not on the path
    torch.save(sd, path)
    net2 = cfg.instantiate(cfg.load_group(confs, "network", "ngp", {}))
    ren2 = cfg.instantiate(cfg.load_group(confs, "renderer", "raymarcher_acc", {}))
    ren2.initialize(1)
    model2 = AvatarModel(None, net2, ren2)
    missing, unexpected = ck.load_checkpoint(model2, path)
    assert unexpected == ["loss_fn.lpips.net.weight"] and not missing and model2.global_step == 1234
    assert torch.equal(net2.encoder.params[:100], torch.arange(100.0))
    sd["state_dict"]["net_coarse.color_net.params"] = torch.zeros(5)
    torch.save(sd, path)
    with pytest.raises(ValueError):
        ck.load_checkpoint(model2, path)


def test_implicit_differentiation_formula_vs_finite_differences(oracle, small_world):
    """Row a7 on the CPU: the closed-form gradient of the implicitly differentiated roots w.r.t. the
    bone transforms (oracle.implicit_diff_grad) against central finite differences of the Broyden
    roots themselves under a perturbation of tfs (the voxel transforms are rebuilt for every probe)."""
    body, init, fp, world = small_world
    tfs, vJ, vd = world["tfs"], world["voxel_J"], world["voxel_d"]
    rng = np.random.RandomState(3)
    v = vd.reshape(3, -1)
    sel = rng.randint(0, v.shape[1], 1500)
    xd = (v[:, sel].T + 0.005 * rng.randn(1500, 3)).astype(np.float32)
    x, Jinv, valid = oracle.broyden(xd, vJ, tfs, init, syn.INIT_BONES)
    valid = valid.astype(bool)
    r = rng.randn(*x.shape).astype(np.float32)
    grad = oracle.implicit_diff_grad(init, x, Jinv, valid, r)
    assert valid.mean() > 0.05 and np.isfinite(grad).all()
    # finite differences on the entries with the largest analytic gradient
    flat = np.abs(grad[:, :3, :]).reshape(-1)
    fd, an = [], []
    for o in np.argsort(-flat)[:5]:
        n, c, k = o // 12, (o % 12) // 4, o % 4
        vals = []
        for sgn in (+1, -1):
            t2 = tfs.copy()
            t2[n, c, k] += sgn * 2e-3
            vJ2, _ = oracle.precompute(init, t2)
            x2, _, v2 = oracle.broyden(xd, vJ2, t2, init, syn.INIT_BONES)
            vals.append((x2, v2.astype(bool)))
        both = valid & vals[0][1] & vals[1][1]
        fd.append((((vals[0][0] - vals[1][0]) * r)[both].sum() / 4e-3))
        an.append(oracle.implicit_diff_grad(init, x, Jinv, both, r)[n, c, k])
    fd, an = np.array(fd), np.array(an)
    assert float((fd * an).sum() / (np.linalg.norm(fd) * np.linalg.norm(an))) > 0.98, (fd, an)


def test_render_train_oracle_is_consistent(oracle, small_world):
    """oracle.render_train (a15): weights of a ray sum to 1 - T_end, empty slots carry no weight, the
    background term is linear, and a frame rendered with jitter 0.5 resembles the test-time render."""
    body, init, fp, world = small_world
    res = 24
    ro, rd = syn.make_camera_rays(res)
    jit = np.random.RandomState(7).rand(2, 64 ** 3, 3).astype(np.float32)
    aabb, density, occ = oracle.density_grid_initialize(world, jit)
    o, d, near, far = oracle.transform_rays_w2s(ro, rd, world["w2s"])
    query = lambda p: oracle.deform_query(p, world, eval_mode=False)
    half = np.full((len(o), 256), 0.5, np.float32)
    a = oracle.render_train(o, d, near, far, occ, aabb, query, half)
    b = oracle.render_train(o, d, near, far, occ, aabb, query, half, bg=np.zeros((len(o), 3), np.float32))
    assert a["n_field"] > 200 and (a["alpha"] > 0.5).mean() > 0.02
    assert np.abs(a["weights"].sum(1) - a["alpha"]).max() < 1e-6
    assert np.abs((a["rgb"] - b["rgb"]) - (1 - a["alpha"])[:, None]).max() < 2e-5      # rgb = sum w c + T_end * bg
    t = oracle.render_test(o, d, near, far, occ, aabb, lambda p: oracle.deform_query(p, world, True))
    assert np.abs(a["alpha"] - t["alpha"]).mean() < 0.02


# ---------------------------------------------------------------- a20: K-nearest-neighbour pin
def _knn_inputs():
    body = syn.make_body()
    verts = np.ascontiguousarray(body["v_template"], np.float32)
    rng = np.random.RandomState(3)
    lo, hi = verts.min(0) - 0.2, verts.max(0) + 0.2
    pts = (rng.rand(1500, 3) * (hi - lo) + lo).astype(np.float32)
    pts[:200] = verts[rng.randint(0, len(verts), 200)] + rng.randn(200, 3).astype(np.float32) * 1e-3  # near-surface queries
    return pts, verts


def test_knn_oracle_matches_pytorch3d_reference(oracle):
    """oracle.knn (the K = 30 neighbour search inside oracle.deformer_initialize / orc_query_weights_smpl) against the
    reference's own pytorch3d KNearestNeighborIdxCpu (third_parties/pytorch3d/cuda/knn_cpu.cpp:13-69), compiled
    unmodified into oracle/_ref/ref_knn.so by oracle/build_ref.py -- and against the frozen golden of the same run."""
    pts, verts = _knn_inputs()
    d, i = oracle.knn(pts, verts, 30)
    gpath = os.path.join(os.path.dirname(__file__), "golden", "knn_golden.npz")
    g = np.load(gpath)
    assert np.array_equal(i[:256], g["idx"]) and np.array_equal(d[:256], g["dist"])   # frozen reference outputs
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "ref_knn.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/ref_knn.so not built (needs /root/reference); golden vectors were checked")
    from oracle import build_ref
    ref = build_ref.load_ext("ref_knn")
    n1, n2 = torch.tensor([len(pts)]), torch.tensor([len(verts)])
    ri, rd = ref.knn_points_idx_cpu(torch.from_numpy(pts)[None], torch.from_numpy(verts)[None], n1, n2, 2, 30)
    assert np.array_equal(ri[0].numpy(), i), "neighbour sets / order differ from pytorch3d"
    assert np.array_equal(rd[0].numpy(), d), "squared distances differ from pytorch3d"


# ---------------------------------------------------------------- a17: training-time occupancy update
def test_density_grid_update_oracle_semantics(oracle, small_world):
    """oracle.density_grid_update restates density_grid.py:46-92: EMA 0.8 with torch.maximum, thresholding on the
    CACHE (not on the fresh densities), `valid` = new field before step 500 and the previous field afterwards,
    and DNeRF.py:99-110's regulariser."""
    body, init, fp, world = small_world
    G = 16
    rng = np.random.RandomState(0)
    cached = np.zeros((G, G, G), np.float32)
    field = np.zeros((G, G, G), bool)
    aabb = np.stack([world["voxel_d"].reshape(3, -1).min(1), world["voxel_d"].reshape(3, -1).max(1)]).astype(np.float32)
    outs = []
    for step in (0, 20, 520):
        jit = rng.rand(G ** 3, 3).astype(np.float32)
        out = oracle.density_grid_update(world, cached, field, jit, step, G=G, aabb=aabb)
        fresh = -100.0 * np.log1p(-out["density"].astype(np.float64))   # invert 1 - exp(-0.01 d)
        assert np.allclose(out["density_cached"], np.maximum(cached * np.float32(0.8), fresh), rtol=2e-3, atol=1e-3)
        assert (out["density_cached"] >= cached * np.float32(0.8) - 1e-6).all()
        assert np.array_equal(out["density_field"], oracle.occupancy_from_density(out["density_cached"].reshape(-1), G).reshape(G, G, G).astype(bool))
        assert np.array_equal(out["valid"], out["density_field"] if step < 500 else field)
        reg = oracle.update_density_grid_reg(out["density"], out["valid"], step)
        expect = 20 * out["density"][~out["valid"]].astype(np.float64).mean() + (0.5 * out["density"].astype(np.float64).mean() if step < 500 else 0)
        assert abs(reg - expect) < 1e-9
        cached, field = out["density_cached"], out["density_field"]
        outs.append(out)
    assert outs[0]["density_field"].any() and (outs[0]["density"] >= 0).all() and (outs[0]["density"] < 1).all()


def test_no_memset_nodes_on_capturable_paths():
    """Every stream-ordered zero-fill of the library is a kernel: hipMemsetAsync recorded into a HIP graph becomes a memset node, and
    a graph with memset nodes replayed back-to-back faulted on ROCm 7.2 (fit stage, round 6; NOTES.md).  Only the profiling hooks
    (ia_prof.hip: synchronous hipMemset, never captured) may call the runtime's memset."""
    import glob, os, re
    from instantavatar_amd import build
    src_dir = os.path.dirname(build.__file__) + "/csrc"
    for path in sorted(glob.glob(src_dir + "/*.hip") + glob.glob(src_dir + "/*.h") + glob.glob(src_dir + "/*.cpp")):
        text = re.sub(r"//[^\n]*", "", open(path).read())
        assert "hipMemsetAsync" not in text and "hipMemset2D" not in text, path
        if not path.endswith("ia_prof.hip"):
            assert "hipMemset(" not in text, path


def test_voxeliser_has_no_cpu_route():
    from instantavatar_amd import _lib
    from instantavatar_amd.deformers.fast_snarf.forward_deformer import voxelise_skinning_weights
    with pytest.raises(_lib.IAError):
        voxelise_skinning_weights(torch.zeros(8, 3), torch.zeros(10, 3), torch.zeros(10, 24), (2, 2, 2))


# ---------------------------------------------------------------- tcnn level-3 resolution (54 vs 55)
def test_tcnn_param_vector_sizes_and_loader_messages(oracle, monkeypatch):
    """The one quantity of the tcnn restatement that cannot be settled offline: level 3 of the hash grid has an exact
    scale of 53.0, so its resolution is 54 or 55 depending on the last bit of exp2f.  Both layouts are supported
    explicitly; a parameter vector of the other layout is rejected with a precise message, never loaded shifted."""
    from instantavatar_amd import _lib
    from instantavatar_amd.models.networks.ngp import NeRFNGPNet
    monkeypatch.delenv("IA_TCNN_LEVEL3_RES", raising=False)
    sizes = NeRFNGPNet.tcnn_encoder_sizes()
    assert sizes[55] - sizes[54] == 2 * 8912                     # 55^3 = 166 375 -> 166 376 vs 54^3 = 157 464 entries
    assert sizes[55] == 3072 + 2 * 6522408                        # 13 044 816 grid parameters
    for r3 in (54, 55):
        hd = _lib.make_hash_desc(16, 19, 16, 1.5, level3_res=r3)
        oh = oracle.hash_desc(16, 19, level3_res=r3)
        assert list(hd.res[:16]) == list(oh.res[:16]) and list(hd.offset[:17]) == list(oh.offset[:17])
        assert hd.res[3] == r3 and all(int(hd.offset[l + 1]) - int(hd.offset[l]) == 524288 for l in range(4, 16))
    net = NeRFNGPNet(dict(center=[0, 0, 0], scale=[1, 1, 1]), level3_res=54)
    enc55 = torch.zeros(sizes[55])
    with pytest.raises(ValueError) as e:
        net.load_tcnn_params(enc55, torch.zeros(6144))
    msg = str(e.value)
    assert str(sizes[55]) in msg and str(sizes[54]) in msg and "level3_res=55" in msg
    with pytest.raises(ValueError):
        net.load_tcnn_params(torch.zeros(sizes[54] - 1), torch.zeros(6144))
    with pytest.raises(ValueError):
        net.load_tcnn_params(torch.zeros(sizes[54]), torch.zeros(6143))
    net.load_tcnn_params(torch.full((sizes[54],), 0.25), torch.full((6144,), -0.5))
    assert float(net.encoder.params.min()) == 0.25 and float(net.color_net.params.max()) == -0.5
    net55 = NeRFNGPNet(dict(center=[0, 0, 0], scale=[1, 1, 1]), level3_res=55)
    net55.load_tcnn_params(enc55, torch.zeros(6144))
    # a loader that is handed a vector of the other layout can FOLLOW it instead: decided by the size alone
    net_a = NeRFNGPNet(dict(center=[0, 0, 0], scale=[1, 1, 1]), level3_res=54)
    assert net_a.adopt_tcnn_layout(sizes[55]) == 55 and net_a.encoder.params.numel() == sizes[55] and int(net_a.hash_desc.res[3]) == 55
    net_a.load_tcnn_params(enc55, torch.zeros(6144))
    assert net_a.adopt_tcnn_layout(sizes[54]) == 54 and net_a.encoder.params.numel() == sizes[54]
    with pytest.raises(ValueError):
        net_a.adopt_tcnn_layout(sizes[54] + 2)
    monkeypatch.setenv("IA_TCNN_LEVEL3_RES", "55")             # the suite-wide switch
    assert NeRFNGPNet(dict(center=[0, 0, 0], scale=[1, 1, 1])).encoder.params.numel() == sizes[55]
    assert int(oracle.hash_desc().res[3]) == 55


# ---------------------------------------------------------------- f4: data-side checker
def test_data_oracle_erode_dilate_match_scipy_and_sampler_semantics():
    """oracle/data_oracle.py: the cv2.erode / cv2.dilate restatement against scipy.ndimage (an independent box
    min / max filter with the same anchor and ignored borders), and the sampler index rules of sampler.py:22-41,56-73."""
    from scipy import ndimage
    from oracle import data_oracle as do
    rng = np.random.RandomState(0)
    m = np.zeros((60, 47), np.float32)
    m[15:40, 10:30] = 1
    m[5:9, 35:44] = 1
    m += (rng.rand(60, 47) > 0.995)
    for k in (3, 4, 16):
        e = ndimage.minimum_filter(m, size=k, mode="constant", cval=np.inf)
        d = ndimage.maximum_filter(m, size=k, mode="constant", cval=-np.inf)
        assert np.array_equal(do.erode(m, k), e) and np.array_equal(do.dilate(m, k), d), k
    draws = rng.rand(200).astype(np.float32)
    idx = do.edge_sampler_indices(m, draws, num_sample=200, ratio_mask=0.6, ratio_edge=0.3, kernel_size=4)
    col = m.reshape(-1, 1)                      # the reference flattens the mask before the morphology (sampler.py:23-27)
    flat, band = m.reshape(-1), (do.dilate(col, 4) - do.erode(col, 4)).reshape(-1)
    assert (flat[idx[:120]] != 0).all() and (band[idx[120:180]] != 0).all() and len(idx) == 200
    # anchors of the mask branch: distinct, and a patch of size P anchored there is centred on a mask pixel
    x, y = do.patch_sampler_corners(m, np.r_[0.0, rng.rand(8)].astype(np.float32), num_patch=4, patch_size=8, ratio_mask=1)
    assert len(set(zip(x.tolist(), y.tolist()))) == 4 and all(m[a + 4, b + 4] > 0 for a, b in zip(x, y))
    x, y = do.patch_sampler_corners(m, np.r_[0.99, rng.rand(8)].astype(np.float32), num_patch=4, patch_size=8, ratio_mask=0.5)
    assert (x >= 0).all() and (x < 60 - 8).all() and (y < 47 - 8).all()
    K = np.array([[900.0, 0, 23.5], [0, 900.0, 30.0], [0, 0, 1]])
    o, d = do.make_rays(K, np.eye(4), 60, 47)
    assert o.shape == (60, 47, 3) and np.allclose(np.linalg.norm(d, axis=-1), 1, atol=1e-6) and abs(d[30, 23, 2] - 1) < 1e-3


# ---------------------------------------------------------------- smpl_init: mesh signed distance
def _cube_mesh(h=0.5):
    v = np.array([[x, y, z] for x in (-h, h) for y in (-h, h) for z in (-h, h)], np.float32)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    f = np.array([t for q in quads for t in ((q[0], q[1], q[2]), (q[0], q[2], q[3]))], np.int32)
    return v, f


def _uv_sphere(r=0.6, n_lat=24, n_lon=48, scale=(1.0, 1.3, 0.8), centre=(0.05, -0.2, 0.1)):
    """closed triangle mesh of an ellipsoid (poles as single vertices)"""
    vs = [(0, 0, r)]
    for i in range(1, n_lat):
        th = np.pi * i / n_lat
        for j in range(n_lon):
            ph = 2 * np.pi * j / n_lon
            vs.append((r * np.sin(th) * np.cos(ph), r * np.sin(th) * np.sin(ph), r * np.cos(th)))
    vs.append((0, 0, -r))
    v = (np.array(vs, np.float32) * np.array(scale, np.float32) + np.array(centre, np.float32)).astype(np.float32)
    f = []
    ring = lambda i, j: 1 + (i - 1) * n_lon + j % n_lon
    for j in range(n_lon):
        f.append((0, ring(1, j), ring(1, j + 1)))
        f.append((len(vs) - 1, ring(n_lat - 1, j + 1), ring(n_lat - 1, j)))
    for i in range(1, n_lat - 1):
        for j in range(n_lon):
            f.append((ring(i, j), ring(i + 1, j), ring(i + 1, j + 1)))
            f.append((ring(i, j), ring(i + 1, j + 1), ring(i, j + 1)))
    return v, np.array(f, np.int32)


def test_mesh_signed_distance_oracle_against_analytic_shapes(oracle):
    """oracle.mesh_signed_distance (the stand-in for kaolin's point_to_mesh_distance + check_sign, density_grid.py:62-70)
    against shapes whose signed distance is known in closed form: a box exactly, a sphere up to its faceting."""
    rng = np.random.RandomState(0)
    p = (rng.rand(20000, 3) * 2 - 1).astype(np.float32)
    v, f = _cube_mesh(0.5)
    sd = oracle.mesh_signed_distance(p, v, f)
    q = np.abs(p) - 0.5
    ana = np.linalg.norm(np.maximum(q, 0), axis=1) + np.minimum(q.max(1), 0)
    assert np.abs(sd - ana).max() < 1e-6 and (np.sign(sd) == np.sign(ana)).all()
    v, f = _uv_sphere(0.6, 48, 96, scale=(1, 1, 1), centre=(0, 0, 0))
    sd = oracle.mesh_signed_distance(p, v, f)
    ana = np.linalg.norm(p, axis=1) - 0.6
    assert np.abs(sd - ana).max() < 2e-3 and ((sd < 0) == (ana < 0))[np.abs(ana) > 2e-3].all()
    out = oracle.density_grid_smpl_init(v, f, np.zeros((16, 16, 16), np.float32), G=16)
    assert out["density_field"].any() and np.isinf(out["density_cached"][out["density_field"]]).all()
    assert (out["density_cached"][~out["density_field"]] == 0).all()


def _lpips_formula_weights(shape, salt):
    """tests/golden/make_lpips_golden.py: the deterministic stand-in trunk weights"""
    n = int(np.prod(shape))
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
    i = np.arange(n, dtype=np.float64)
    v = np.sin(i * 12.9898 + salt * 78.233) * np.sqrt(2.0 / fan_in) * 1.7
    if len(shape) == 1:
        v = 0.05 * np.sin(i * 0.7 + salt)
    import torch
    return torch.as_tensor(v.reshape(shape), dtype=torch.float32)


def test_lpips_module_matches_reference_golden():
    """utils.lpips.LPIPS against outputs of the REFERENCE's third_parties/lpips module (net="vgg", v0.1, its pretrained
    lin layers; trunk weights from a closed formula because torchvision's are not available offline): the state-dict
    layout is the reference's and the per-layer and total distances agree."""
    import torch
    from instantavatar_amd.utils.lpips import LPIPS
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "lpips_golden.npz"))
    m = LPIPS()
    m.load_lin_weights({k.replace("__", "."): torch.as_tensor(g[k]) for k in g.files if k.startswith("lin")})
    names = sorted(n for n, _ in m.net.named_parameters())
    assert names == [str(s) for s in g["param_order"]]                 # same parameter names as the reference's trunk wrapper
    with torch.no_grad():
        for salt, name in enumerate(names):
            p = dict(m.net.named_parameters())[name]
            p.copy_(_lpips_formula_weights(tuple(p.shape), salt))
        val, per = m(torch.as_tensor(g["x"]), torch.as_tensor(g["y"]), per_layer=True)
    assert np.allclose(val.numpy(), g["val"], rtol=2e-5, atol=1e-7), (val.reshape(-1), g["val"].reshape(-1))
    for k in range(5):
        assert np.allclose(per[k].numpy(), g["per"][k], rtol=2e-5, atol=1e-8), k
    # identical images -> 0; the loader of a torchvision `features` state dict maps onto the sliced layout
    assert float(m(torch.as_tensor(g["x"]), torch.as_tensor(g["x"])).abs().max()) == 0.0
    tv_sd = {n.split(".", 1)[1]: p.detach().clone() for n, p in m.net.named_parameters()}
    m2 = LPIPS().load_lin_weights({k.replace("__", "."): torch.as_tensor(g[k]) for k in g.files if k.startswith("lin")}).load_trunk_weights(tv_sd)
    with torch.no_grad():
        assert torch.equal(m2(torch.as_tensor(g["x"]), torch.as_tensor(g["y"])), val)
    assert m2.weights_loaded == {"trunk": True, "lin": True}


def test_lpips_alex_matches_reference_golden():
    """utils.lpips.LPIPS(net="alex") -- the network behind eval.py's LPIPS figure -- against the REFERENCE's third_parties/lpips
    module (net="alex", v0.1, its pretrained lin layers; formula trunk weights): parameter names, per-layer and total distances,
    both with the [0,1] -> [-1,1] rescaling and without it (what eval.py effectively computes: torchmetrics' default
    normalize=False on [0, 1] images)."""
    import torch
    from instantavatar_amd.utils.lpips import LPIPS, ALEX_CHANNELS
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "lpips_alex_golden.npz"))
    m = LPIPS(net="alex")
    m.load_lin_weights({k.replace("__", "."): torch.as_tensor(g[k]) for k in g.files if k.startswith("lin")})
    assert [tuple(l.model[0].weight.shape) for l in m.lins] == [(1, c, 1, 1) for c in ALEX_CHANNELS]
    names = sorted(n for n, _ in m.net.named_parameters())
    assert names == [str(s) for s in g["param_order"]]
    x, y = torch.as_tensor(g["x"]), torch.as_tensor(g["y"])
    with torch.no_grad():
        for salt, name in enumerate(names):
            p = dict(m.net.named_parameters())[name]
            p.copy_(_lpips_formula_weights(tuple(p.shape), salt))
        val, per = m(x, y, per_layer=True)
        val_raw, per_raw = m(x, y, normalize=False, per_layer=True)
    assert np.allclose(val.numpy(), g["val"], rtol=2e-5, atol=1e-7) and np.allclose(val_raw.numpy(), g["val_raw"], rtol=2e-5, atol=1e-7)
    for k in range(5):
        assert np.allclose(per[k].numpy(), g["per"][k], rtol=2e-5, atol=1e-8) and np.allclose(per_raw[k].numpy(), g["per_raw"][k], rtol=2e-5, atol=1e-8), k
    with pytest.raises(ValueError, match="alex"):
        LPIPS(net="squeeze")
    with pytest.raises(KeyError, match="alex.pth"):
        LPIPS(net="alex").load_lin_weights({})


def test_ngp_loss_lpips_term_and_missing_weights():
    """NGPLoss with w_lpips > 0: refuses to run without weight files, and with a loaded module adds
    w_lpips * sum(LPIPS(pred[BGR], target[BGR])) on patch batches (loss.py:28-32)."""
    import torch
    from instantavatar_amd.training import NGPLoss
    from instantavatar_amd.utils.lpips import LPIPS
    with pytest.raises(NotImplementedError, match="lpips_lin_weights"):
        NGPLoss(dict(w_rgb=1.0, w_alpha=0.1, w_reg=0.1, w_lpips=0.01))
    torch.manual_seed(0)
    lp = LPIPS()
    loss = NGPLoss(dict(w_rgb=1.0, w_alpha=0.1, w_reg=0.1, w_lpips=0.01), fused=False, lpips=lp)
    base = NGPLoss(dict(w_rgb=1.0, w_alpha=0.1, w_reg=0.1), fused=False)
    pred = {"rgb_coarse": torch.rand(1, 2, 32, 32, 3, requires_grad=True), "alpha_coarse": torch.rand(1, 2, 32, 32),
            "depth_coarse": torch.rand(1, 2, 32, 32), "weight_coarse": torch.rand(1, 2, 32, 32, 8)}
    tgt = {"rgb": torch.rand(1, 2, 32, 32, 3), "alpha": torch.rand(1, 2, 32, 32)}
    a, b = loss(pred, tgt), base(pred, tgt)
    want = lp(pred["rgb_coarse"][0][..., [2, 1, 0]].permute(0, 3, 1, 2).clip(max=1), tgt["rgb"][0][..., [2, 1, 0]].permute(0, 3, 1, 2)).sum()
    assert torch.allclose(a["loss_lpips"], want) and torch.allclose(a["loss"], b["loss"] + 0.01 * want)
    a["loss"].backward()
    assert pred["rgb_coarse"].grad is not None and torch.isfinite(pred["rgb_coarse"].grad).all()
    assert all(not p.requires_grad for p in lp.parameters())
    # flat (non-patch) batches skip the term, as the reference does
    flat = {k: v.reshape(1, -1, *v.shape[4:]) for k, v in pred.items()}
    assert "loss_lpips" not in loss(flat, {k: v.reshape(1, -1, *v.shape[4:]) for k, v in tgt.items()})


def test_implicit_differentiation_matches_reference_autograd_golden(oracle, small_world):
    """Row a7 pinned to the REFERENCE: oracle.implicit_diff_grad (the closed form the HIP kernel k_implicit_bwd is tested
    against) equals the gradient w.r.t. tfs that the reference's own autograd produces for the training branch of
    ForwardDeformer.forward (deformer_torch.py:50-67) -- golden generated by tests/golden/make_implicit_diff_golden.py, which
    imports the reference module on the CPU (CUDA extensions and KNN stubbed: they are not on the differentiated path) and
    feeds it the oracle's roots / J_inv / voxel weights."""
    body, init, fp, world = small_world
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "implicit_diff_golden.npz"))
    assert np.array_equal(g["tfs"], world["tfs"])                       # same world as the generator built
    got = oracle.implicit_diff_grad(init, g["xc"], g["J_inv"], g["valid"], g["r"])
    ref = g["grad_tfs"]
    assert ref.shape == (24, 4, 4) and np.abs(ref).max() > 1.0 and np.abs(ref[:, 3]).max() == 0
    assert np.abs(got - ref).max() < 2e-5 * np.abs(ref).max(), float(np.abs(got - ref).max())
    # and the roots the golden was built on are the oracle's Broyden roots of this world
    x, Jinv, valid = oracle.broyden(g["xd"], world["voxel_J"], world["tfs"], init, syn.INIT_BONES)
    assert np.array_equal(x, g["xc"]) and np.array_equal(oracle.filter_dup(x, valid).astype(bool), g["valid"])


def test_inverse_skinning_version2_matches_reference_autograd_golden(oracle, small_world):
    """ForwardDeformer `version: 2` (deformer_torch.py:68-75, confs/deformer/fast_snarf_debug.yaml): oracle.inverse_skinning --
    what the HIP kernels of that branch are tested against -- equals the VALUE the reference's forward returns and the gradient
    w.r.t. tfs its autograd produces (same generator, same roots as the version-1 golden)."""
    body, init, fp, world = small_world
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "implicit_diff_golden.npz"))
    val, grad = oracle.inverse_skinning(init, g["xc"], g["xd"], g["valid"], g["tfs"], g["r"])
    ref_v, ref_g = g["xc_value_v2"], g["grad_tfs_v2"]
    m = g["valid"].astype(bool)
    assert np.abs(ref_v[~m]).max() == 0 and np.abs(val[~m]).max() == 0                     # invalid slots are zero
    assert np.abs(val - ref_v).max() < 2e-6, float(np.abs(val - ref_v).max())
    assert np.abs(ref_v[m] - g["xc"][m]).max() > 1e-2                                       # NOT the roots: a blended R is not orthogonal
    assert np.abs(ref_g).max() > 1.0 and np.abs(ref_g[:, 3]).max() == 0
    assert np.abs(grad - ref_g).max() < 2e-5 * np.abs(ref_g).max(), float(np.abs(grad - ref_g).max())
    assert np.abs(ref_g - g["grad_tfs"]).max() > 0.1                                        # and not version 1's gradient either


@pytest.mark.parametrize("blend", [False, True])
def test_oracle_pipeline_matches_reference_python_golden(oracle, request, blend):
    """The oracle's restatement of the reference's Python glue against the REFERENCE'S PYTHON EXECUTING
    (tests/golden/make_pipeline_golden.py + ref_cpu_harness.py: instant_avatar.* imported on the CPU, its native
    extensions / tcnn replaced by adapters around the oracle's C functions, its random draws taken from seeded numpy
    streams re-created here):
      (A) DNeRFModel.render_image_fast: SMPL / LBS -> tfs, w2s; DensityGrid.initialize; Raymarcher.render_test
      (B) DNeRFModel.update_density_grid x 2 (steps 0 and 500: EMA, post-processing, valid switch, regulariser)
      (C) DNeRFModel.forward in training mode: Raymarcher.render_train with jitter and sigma noise
      (D) ForwardDeformer.switch_to_explicit's skinning-weight voxels (KNN + smoothing in torch)."""
    body, init, fp, _ = request.getfixturevalue("small_world_blend" if blend else "small_world")
    betas = syn.BLEND_BETAS if blend else np.zeros(10, np.float32)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pipeline_golden%s%s.npz" % ("_blend" if blend else "", _LAYOUT)))
    seed_init, seed_upd, seed_train = (int(v) for v in g["seeds"])
    res, frame = int(g["res"]), int(g["frame"])
    poses, tr = syn.procedural_pose_track(8)
    world = oracle.make_world(body, init, fp, betas, poses[frame, 3:], poses[frame, :3], tr[frame], syn.INIT_BONES)
    # (D) and the per-frame transforms (reference smplx + lbs.py + torch.inverse vs the oracle's numpy chain)
    assert float(g["D_lbs_max_abs_diff_to_oracle"]) < 1e-6
    assert np.abs(init["lbs_voxel"].reshape(24, -1)[:, ::97] - g["D_lbs_sample"]).max() < 1e-6
    assert np.abs(world["tfs"] - g["tfs"]).max() < 2e-6 and np.abs(world["w2s"] - g["w2s"]).max() < 2e-6
    assert np.abs(init["bbox"] - g["bbox"]).max() < 1e-6
    # (A)
    rs = np.random.RandomState(seed_init)
    jitter = np.stack([rs.rand(64, 64, 64, 3).astype(np.float32).reshape(-1, 3) for _ in range(5)])
    ro, rd = syn.make_camera_rays(res)
    out = oracle.render_image_fast(world, ro, rd, jitter)
    occ_ref = np.unpackbits(g["A_occ"])[:64 ** 3].reshape(64, 64, 64)
    assert np.abs(out["aabb"] - g["A_aabb"]).max() < 1e-5
    assert (out["occ"].astype(np.uint8) != occ_ref).mean() < 2e-4          # threshold / component decisions on ~1e-6 differences
    d_rgb = np.abs(out["rgb"].reshape(res, res, 3) - g["A_rgb"]).max(-1)
    d_alpha = np.abs(out["alpha"].reshape(res, res) - g["A_alpha"])
    # (the transforms of the two sides differ by 1e-6 -- torch vs numpy LBS -- which the steep field turns into ~1e-4 on a few rays)
    assert (d_rgb > 2e-4).mean() < 5e-3 and (d_alpha > 2e-4).mean() < 5e-3 and d_rgb.max() < 1e-3, ((d_rgb > 2e-4).mean(), d_rgb.max())
    assert (out["counter"].reshape(res, res) != g["A_counter"]).mean() < 5e-3
    assert np.median(d_rgb) < 1e-6 and (g["A_alpha"] > 0.5).mean() > 0.03
    hit = g["A_alpha"] > 0.5
    assert np.abs(out["depth"].reshape(res, res) - g["A_depth"])[hit].max() < 2e-3
    # (A2): same frame, MAX_BATCH_SIZE = 4096 -> many wave-front iterations with a changing N_step (raymarcher_acc.py:107)
    o_s, d_s, near_s, far_s = oracle.transform_rays_w2s(ro, rd, world["w2s"])
    a2 = oracle.render_test(o_s, d_s, near_s, far_s, occ_ref, g["A_aabb"], lambda p: oracle.deform_query(p, world, True), MAX_BATCH_SIZE=4096)
    assert (a2["counter"].reshape(-1) != g["A2_counter"].reshape(-1)).mean() < 5e-3 and g["A2_counter"].sum() < g["A_counter"].sum()
    assert (np.abs(a2["rgb"].reshape(res, res, 3) - g["A2_rgb"].reshape(res, res, 3)).max(-1) > 2e-4).mean() < 5e-3
    assert (np.abs(a2["alpha"].reshape(res, res) - g["A2_alpha"].reshape(res, res)) > 2e-4).mean() < 5e-3
    # (B)
    cached, field = np.zeros((64, 64, 64), np.float32), np.zeros((64, 64, 64), bool)
    for k, step in enumerate((0, 500)):
        jit = np.random.RandomState(seed_upd + k).rand(64, 64, 64, 3).astype(np.float32)
        u = oracle.density_grid_update(world, cached, field, jit, step)
        reg = oracle.update_density_grid_reg(u["density"], u["valid"], step)
        ref_field = np.unpackbits(g["B%d_field" % k])[:64 ** 3].reshape(64, 64, 64).astype(bool)
        assert (u["density_field"] != ref_field).mean() < 2e-4, k
        assert np.abs(u["density_cached"].reshape(-1)[::61] - g["B%d_cached_sample" % k]).max() < 1e-3 * max(1.0, float(g["B%d_cached_sample" % k].max()))
        assert abs(float(u["density_cached"].astype(np.float64).sum()) - float(g["B%d_cached_sum" % k])) < 1e-4 * float(g["B%d_cached_sum" % k])
        assert abs(float(reg) - float(g["B%d_reg" % k])) <= 1e-4 * abs(float(g["B%d_reg" % k])) + 1e-9, (k, float(reg), float(g["B%d_reg" % k]))
        cached, field = u["density_cached"], ref_field          # continue from the reference's state
    # (C)
    sel = g["C_sel"]
    o, d, near, far = oracle.transform_rays_w2s(ro[sel], rd[sel], world["w2s"])
    rs = np.random.RandomState(seed_train)
    n = len(sel)
    jit, noise = rs.rand(n, 256).astype(np.float32), rs.randn(n, 256).astype(np.float32)
    c = oracle.render_train(o, d, near, far, field, oracle.TRAIN_AABB, lambda p: oracle.deform_query(p, world, eval_mode=False), jit,
                            bg=g["C_bg"], noise=noise)
    for key, ref in (("rgb", g["C_rgb"]), ("alpha", g["C_alpha"]), ("depth", g["C_depth"]), ("weights", g["C_weights"])):
        dd = np.abs(c[key].reshape(ref.shape) - ref)
        assert (dd > 2e-4).mean() < 5e-3 and np.median(dd) < 1e-6, (key, (dd > 2e-4).mean(), dd.max())
    assert g["C_alpha"].max() > 0.5


@pytest.mark.parametrize("blend", [False, True])
def test_smpl_deformer_oracle_matches_reference_python_golden(oracle, request, blend):
    """f2: the oracle's SMPLDeformer restatement (smpl_deformer_prepare / smpl_nn_deform / smpl_deform_query) against
    the REFERENCE's smpl_deformer.py executing on the CPU (tests/golden/make_smpl_deformer_golden.py): per-vertex inverse
    transforms, posed vertices, boxes, nearest-vertex deformation, test- and train-mode field queries."""
    body, init, fp0, _ = request.getfixturevalue("small_world_blend" if blend else "small_world")
    betas = syn.BLEND_BETAS if blend else np.zeros(10, np.float32)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "smpl_deformer_golden%s%s.npz" % ("_blend" if blend else "", _LAYOUT)))
    frame = int(g["frame"])
    poses, tr = syn.procedural_pose_track(8)
    prep = oracle.smpl_deformer_prepare(body, betas, poses[frame, 3:], poses[frame, :3], tr[frame])
    assert np.abs(prep["T_inv"][::53] - g["T_inv_sample"]).max() < 5e-6
    assert np.abs(prep["vertices"][::53] - g["verts_sample"]).max() < 2e-6
    assert np.abs(prep["w2s"] - g["w2s"]).max() < 2e-6 and np.abs(prep["bbox"] - g["bbox"]).max() < 2e-6
    assert np.abs(oracle.get_bbox_from_smpl(prep["vertices"]) - g["bbox_deformed"]).max() < 2e-6
    cano, valid, _ = oracle.smpl_nn_deform(g["pts"], prep["vertices"], prep["T_inv"], 0.05)
    # a point within 1e-6 of the 5 cm threshold or of a tie between two vertices may flip with the 1e-6 vertex differences
    assert (valid != g["valid"]).mean() < 2e-3
    both = valid & g["valid"]
    d = np.abs(cano - g["cano"])[both].max(-1)
    assert np.median(d) < 2e-6 and (d > 1e-4).mean() < 2e-3, (np.median(d), (d > 1e-4).mean())
    field, keep = oracle.make_field(syn.make_field(g["cano_joints"], prep["bbox"], seed=42, n_levels=16))
    for mode, kr, ks in ((True, "rgb_test", "sigma_test"), (False, "rgb_train", "sigma_train")):
        rgb, sigma = oracle.smpl_deform_query(g["pts"], prep, field, eval_mode=mode)
        same = (valid == g["valid"])
        ds, dr = np.abs(sigma - g[ks])[same], np.abs(rgb - g[kr])[same].max(-1)
        # the field is steep (|sigma| up to 120): compare where the canonical points agree to 1e-6
        tight = same.copy(); tight[both] &= d < 1e-6
        assert (np.abs(sigma - g[ks])[tight] > 2e-2 * (1 + np.abs(g[ks][tight]))).mean() < 5e-3, mode
        assert (np.abs(rgb - g[kr])[tight].max(-1) > 2e-3).mean() < 5e-3, mode
        assert np.array_equal(sigma[~valid & ~g["valid"]], g[ks][~valid & ~g["valid"]])      # fills: 0 (test) / -1e5 (train)
    # the frame DNeRFModel.render_image_fast renders with this deformer plugged in: occupancy build from get_bbox_deformed,
    # wave-front loop through the generic deformer(pts, net) closure
    G = 64
    rs = np.random.RandomState(77)
    jit = [rs.rand(G, G, G, 3).astype(np.float32).reshape(-1, 3) for _ in range(5)]
    query = lambda p: oracle.smpl_deform_query(p, prep, field, eval_mode=True)
    aabb = oracle.get_bbox_from_smpl(prep["vertices"])
    idx = np.arange(G, dtype=np.float32)
    cx, cy, cz = np.meshgrid(idx, idx, idx, indexing="ij")
    coords0 = (np.stack([cx, cy, cz], -1).reshape(-1, 3) / np.float32(G)).astype(np.float32)
    density = np.zeros(G ** 3, np.float32)
    for j in jit:
        density = np.maximum(density, query(((coords0 + j / np.float32(G)) * (aabb[1] - aabb[0]) + aabb[0]).astype(np.float32))[1])
    occ = oracle.occupancy_from_density(density, G)
    occ_ref = np.unpackbits(g["F_occ"])[:G ** 3].reshape(G, G, G)
    assert np.abs(aabb - g["F_aabb"]).max() < 2e-6 and (occ.astype(np.uint8) != occ_ref).mean() < 3e-4
    ro, rd = syn.make_camera_rays(32)
    o, dd, near, far = oracle.transform_rays_w2s(ro, rd, prep["w2s"])
    ref = oracle.render_test(o, dd, near, far, occ_ref, g["F_aabb"], query)
    e_rgb = np.abs(ref["rgb"].reshape(32, 32, 3) - g["F_rgb"]).max(-1)
    assert (e_rgb > 1e-3).mean() < 1e-2 and np.median(e_rgb) < 1e-5, ((e_rgb > 1e-3).mean(), e_rgb.max())
    assert (np.abs(ref["alpha"].reshape(32, 32) - g["F_alpha"]) > 1e-3).mean() < 1e-2 and (g["F_alpha"] > 0.5).mean() > 0.03


def test_data_oracle_matches_reference_python_golden():
    """f4: oracle/data_oracle.py against the REFERENCE's data side executing on the CPU (tests/golden/make_data_golden.py:
    peoplesnapshot.make_rays, PeopleSnapshotDataset.__getitem__ (train), EdgeSampler, PatchSampler with / without dilate and
    its uniform branch; cv2 morphology stood in by scipy filters, numpy's random functions scripted from explicit draws).
    Includes the reference's flattened-mask edge band: EdgeSampler reshapes the mask to 1-D before cv2.erode / dilate."""
    from oracle import data_oracle as do
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_data_golden as mk
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "data_golden.npz"))
    mask, img, K, c2w, smpl = mk.scene()
    H, W = mask.shape
    ro, rd = do.make_rays(K, c2w, H, W)
    assert np.array_equal(ro, g["rays_o"]) and np.abs(rd - g["rays_d"]).max() <= 6e-8
    imgf = img.astype(np.float32)
    out = do.edge_sampler_sample(mask, [imgf, g["rays_o"], g["rays_d"]], g["edge_draws"], num_sample=512, ratio_mask=0.6, ratio_edge=0.3, kernel_size=8)
    for i, a in enumerate(out):
        assert np.array_equal(np.asarray(a).reshape(g["edge_out%d" % i].shape), g["edge_out%d" % i]), ("edge", i)
    # the flattened band differs from a 2-D band: make sure the golden really exercises that
    band1 = (do.dilate(mask.reshape(-1, 1), 8) - do.erode(mask.reshape(-1, 1), 8)).reshape(-1)
    band2 = (do.dilate(mask, 8) - do.erode(mask, 8)).reshape(-1)
    assert (band1 != band2).sum() > 50
    for tag, dil in (("pm", 0), ("pd", 6), ("pu", 0)):
        x, y = do.patch_sampler_corners(mask, g[tag + "_draws"], num_patch=4, patch_size=16, ratio_mask=0.9, dilate_k=dil)
        ref_mask_patches = g[tag + "_out0"]
        got = np.stack([mask[a:a + 16, b:b + 16] for a, b in zip(x, y)])
        assert np.array_equal(got, ref_mask_patches), tag
        got_rays = np.stack([g["rays_d"][a:a + 16, b:b + 16] for a, b in zip(x, y)])
        assert np.array_equal(got_rays, g[tag + "_out3"]), tag
    fn = lambda msk, *args: do.patch_sampler_sample(msk, args, g["gi_draws"], 4, 16, 0.9)
    d = do.getitem_train(img, mask, g["rays_o"], g["rays_d"], smpl, 0, fn, g["gi_bg"])
    for k in ("rgb", "rays_o", "rays_d", "betas", "global_orient", "body_pose", "transl", "alpha", "bg_color", "near", "far"):
        assert np.array_equal(np.asarray(d[k], np.float32), np.asarray(g["gi_" + k], np.float32)), k
    assert int(d["idx"]) == int(g["gi_idx"])
    # the "val" split: whole frame, white background
    e = do.getitem_eval(img, mask, g["rays_o"], g["rays_d"], smpl, 0)
    for k in ("rgb", "rays_o", "rays_d", "betas", "global_orient", "body_pose", "transl", "alpha", "bg_color", "near", "far"):
        a, r = np.asarray(e[k], np.float32), np.asarray(g["ge_" + k], np.float32)
        assert a.shape == r.shape and np.array_equal(a, r), k


def test_animate_sequence_matches_reference_animate_dataset_golden():
    """drivers.animate.AnimateSequence (camera, pose-track handling, per-frame batch) against the REFERENCE's
    AnimateDataset executing on the CPU on the pose track the reference ships (tests/golden/make_animate_golden.py; the
    three frames compared travel with the golden)."""
    from instantavatar_amd.drivers.animate import AnimateSequence
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "animate_golden.npz"))
    seq = AnimateSequence(g["track_poses"], g["track_trans"], g["betas"], "cpu", downscale=16)
    assert (seq.H, seq.W) == (int(g["H"]), int(g["W"])) and int(g["n"]) == 320
    for row, f in enumerate(g["frames"]):
        b = seq.batch(row)
        for k in ("rays_o", "rays_d", "betas", "global_orient", "body_pose", "transl", "near", "far"):
            a, r = b[k][0].numpy(), g["f%d_%s" % (f, k)]
            assert a.shape == r.shape, (k, a.shape, r.shape)
            assert np.array_equal(a, r) or np.abs(a - r).max() <= 1.2e-7 * max(1.0, np.abs(r).max()), (int(f), k, np.abs(a - r).max())


def test_rotation_sequence_matches_reference_novel_view_dataset_golden():
    """drivers.novel_view.RotationSequence (camera, the fixed pose, the turn about y composed with the body orientation and
    converted back to a rotation vector) against the REFERENCE's novel_view.py AnimateDataset executing on the CPU
    (tests/golden/make_novel_view_golden.py; its cv2.Rodrigues is a scipy stand-in).  Orientations are compared as rotation
    matrices: every frame is a rotation by exactly pi, where +-pi k are the same rotation."""
    from instantavatar_amd.drivers.novel_view import RotationSequence, rotvec_to_matrix
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "novel_view_golden.npz"))
    n = int(g["n"])
    seq = RotationSequence(n, g["betas"], "cpu", downscale=16)
    assert (seq.H, seq.W) == (int(g["H"]), int(g["W"])) and len(seq) == n
    b = seq.batch(0)
    for k in ("rays_o", "rays_d", "betas", "body_pose", "transl", "near", "far"):
        a, r = b[k][0].numpy(), g["f0_" + k]
        assert a.shape == r.shape, (k, a.shape, r.shape)
        assert np.array_equal(a, r) or np.abs(a - r).max() <= 1.2e-7 * max(1.0, np.abs(r).max()), (k, np.abs(a - r).max())
    for f in range(n):
        a, r = seq.batch(f)["global_orient"][0].numpy(), g["f%d_global_orient" % f]
        assert a.shape == r.shape == (3,) and a.dtype == np.float32
        assert abs(np.linalg.norm(a) - np.pi) < 1e-6                                        # R_y(angle) R_x(pi) always turns by pi
        assert np.abs(rotvec_to_matrix(a) - rotvec_to_matrix(r)).max() < 2e-6, (f, a, r)


def test_rodrigues_conversions_against_scipy():
    """drivers.novel_view.rotvec_to_matrix / matrix_to_rotvec (the package's stand-in for cv2.Rodrigues) against
    scipy.spatial.transform.Rotation over random rotations, small angles and the neighbourhood of pi."""
    from scipy.spatial.transform import Rotation
    from instantavatar_amd.drivers.novel_view import matrix_to_rotvec, rotvec_to_matrix
    rs = np.random.RandomState(3)
    for th in list(rs.uniform(0, np.pi, 200)) + [0.0, 1e-9, 1e-4, np.pi - 1e-4, np.pi - 1e-7, np.pi]:
        k = rs.randn(3)
        v = k / np.linalg.norm(k) * th
        R = rotvec_to_matrix(v)
        assert np.abs(R - Rotation.from_rotvec(v).as_matrix()).max() < 1e-12
        w = matrix_to_rotvec(R)
        assert np.abs(rotvec_to_matrix(w) - R).max() < 1e-7 and np.linalg.norm(w) <= np.pi + 1e-9
        if th < np.pi - 1e-3:
            assert np.abs(w - v).max() < 1e-7, (th, v, w)


def test_eval_metrics_psnr_ssim_and_evaluator():
    """utils.metrics (the figures of eval.py's Evaluator): PSNR against its definition, SSIM against an independent
    scipy implementation of Wang et al. 2004 with the window torchmetrics uses (11 x 11 Gaussian, sigma 1.5, reflect
    padding, border cropped), the Evaluator's clamp / layout handling, and LPIPS going in WITHOUT the [-1, 1] rescaling."""
    from scipy.ndimage import correlate1d
    from instantavatar_amd.utils.lpips import LPIPS
    from instantavatar_amd.utils.metrics import Evaluator, psnr, ssim
    rs = np.random.RandomState(0)
    x = rs.rand(2, 3, 40, 52)
    y = np.clip(x + 0.1 * rs.randn(2, 3, 40, 52), 0, 1)

    def ssim_np(x, y):
        d = np.arange(-5, 6)
        g = np.exp(-(d / 1.5) ** 2 / 2)
        g /= g.sum()
        f = lambda a: correlate1d(correlate1d(a, g, axis=-1, mode="mirror"), g, axis=-2, mode="mirror")
        mx, my = f(x), f(y)
        sxx, syy, sxy = f(x * x) - mx * mx, f(y * y) - my * my, f(x * y) - mx * my
        c1, c2 = 0.01 ** 2, 0.03 ** 2
        m = ((2 * mx * my + c1) * (2 * sxy + c2)) / ((mx * mx + my * my + c1) * (sxx + syy + c2))
        return m[..., 5:-5, 5:-5].reshape(len(x), -1).mean(-1).mean()
    assert abs(float(ssim(torch.tensor(x), torch.tensor(y))) - ssim_np(x, y)) < 1e-12
    assert abs(float(ssim(torch.tensor(x, dtype=torch.float32), torch.tensor(y, dtype=torch.float32))) - ssim_np(x, y)) < 2e-6
    assert float(ssim(torch.tensor(x), torch.tensor(x))) == 1.0
    assert abs(float(psnr(torch.tensor(x), torch.tensor(y))) - 10 * np.log10(1.0 / ((x - y) ** 2).mean())) < 1e-9
    assert abs(float(psnr(torch.tensor(x) * 255, torch.tensor(y) * 255, data_range=255.0)) - float(psnr(torch.tensor(x), torch.tensor(y)))) < 1e-9
    with pytest.raises(ValueError, match="NCHW"):
        ssim(torch.zeros(3, 8, 8), torch.zeros(3, 8, 8))
    # Evaluator: NHWC in, prediction clamped to <= 1, no LPIPS entry without a loaded network, a network without weights refused
    pred = torch.tensor(np.transpose(y, (0, 2, 3, 1)), dtype=torch.float32) * 1.2
    gt = torch.tensor(np.transpose(x, (0, 2, 3, 1)), dtype=torch.float32)
    out = Evaluator()(pred, gt)
    assert set(out) == {"psnr", "ssim"}
    pc = pred.clamp(max=1.0).permute(0, 3, 1, 2)
    assert abs(float(out["psnr"]) - float(psnr(pc, gt.permute(0, 3, 1, 2)))) < 1e-5 and abs(float(out["ssim"]) - float(ssim(pc, gt.permute(0, 3, 1, 2)))) < 1e-6
    with pytest.raises(ValueError, match="pretrained"):
        Evaluator(lpips=LPIPS(net="alex"))
    lp = LPIPS(net="alex")
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "lpips_alex_golden.npz"))
    lp.load_lin_weights({k.replace("__", "."): torch.as_tensor(g[k]) for k in g.files if k.startswith("lin")})
    with torch.no_grad():
        for salt, name in enumerate(sorted(n for n, _ in lp.net.named_parameters())):
            p = dict(lp.net.named_parameters())[name]
            p.copy_(_lpips_formula_weights(tuple(p.shape), salt))
    lp.weights_loaded["trunk"] = True
    gx, gy = torch.as_tensor(g["x"]).permute(0, 2, 3, 1), torch.as_tensor(g["y"]).permute(0, 2, 3, 1)
    out = Evaluator(lpips=lp)(gx, gy)
    assert abs(float(out["lpips"]) - float(g["val_raw"].mean())) < 2e-6            # normalize=False, mean over the batch


def test_test_image_round_trip_is_cv2_imwrite_then_imread_bgr2rgb(tmp_path):
    """evaluation.write_png_bgr / read_png_as_rgb / evaluate_folder: the file holds the panel's channels reversed and rounded
    half-to-even (cv2.imwrite of a float image), reading it back returns the reversed 8-bit array / 255 (cv2.imread +
    BGR2RGB), and the folder's metrics are prediction (middle third) against ground truth (left third); the JET table has
    OpenCV's end points and ramps."""
    from PIL import Image
    from instantavatar_amd import evaluation as ev
    from instantavatar_amd.utils.metrics import Evaluator, psnr
    rs = np.random.RandomState(5)
    H, W = 24, 20
    gt, pred = rs.rand(H, W, 3).astype(np.float32), rs.rand(H, W, 3).astype(np.float32)
    err = rs.rand(H, W, 3).astype(np.float32)
    panel = torch.tensor(np.concatenate([gt, pred, err], 1))
    ev.write_png_bgr(str(tmp_path / "0.png"), panel)
    raw = np.asarray(Image.open(str(tmp_path / "0.png")))
    want = np.clip(np.rint(panel.numpy() * np.float32(255)), 0, 255).astype(np.uint8)          # fp32 product, as the reference forms it
    assert raw.shape == (H, 3 * W, 3) and np.array_equal(raw, want[..., ::-1])
    assert ev.to_u8(torch.tensor([0.5, 1.5, 2.5, 254.5, 255.5, -0.5, 300.0])).tolist() == [0, 2, 2, 254, 255, 0, 255]   # ties to even, saturation
    back = ev.read_png_as_rgb(str(tmp_path / "0.png"))
    assert np.array_equal(back.numpy(), want[..., ::-1].astype(np.float32) / 255)
    ev.write_png_bgr(str(tmp_path / "1.png"), panel.flip(0))
    res, n = ev.evaluate_folder(str(tmp_path), Evaluator(), "cpu")
    q = lambda a: torch.tensor(np.rint(a * np.float32(255)) / 255)
    assert n == 2 and abs(res["psnr"] - float(psnr(q(pred), q(gt)))) < 1e-4
    ev.write_results(str(tmp_path / "results.txt"), res)
    assert open(str(tmp_path / "results.txt")).read().splitlines() == ["PSNR: %.2f" % res["psnr"], "SSIM: %.4f" % res["ssim"]]
    lut = ev.jet_bgr(torch.arange(256, dtype=torch.uint8)).numpy()
    assert list(lut[0]) == [128, 0, 0] and list(lut[255]) == [0, 0, 128]                  # dark blue ... dark red, B G R order
    assert list(lut[51]) == [255, 77, 0] and list(lut[204]) == [0, 76, 255] and lut[:, 1].max() == 255 and int(lut[:, 1].argmax()) in range(96, 160)


def test_losses_match_reference_loss_py_golden():
    """training.NeRFLoss / NGPLoss (torch formulation, fused=False: what the fused HIP kernel is tested against on the GPU)
    against the REFERENCE's utils/loss.py executing on the CPU (tests/golden/make_loss_golden.py): every reported value and
    the gradients w.r.t. rgb / alpha / depth / weights."""
    from instantavatar_amd.training import NGPLoss, NeRFLoss
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_golden.npz"))
    for tag, loss in (("nerf", NeRFLoss(dict(w_rgb=1.0, w_alpha=0.1, w_reg=0.1), fused=False)),
                      ("ngp", NGPLoss(dict(w_rgb=1.0, w_alpha=0.1, w_reg=0.1, w_lpips=0.0, w_depth_reg=0.01), fused=False))):
        pred = {k: torch.tensor(g["%s_in_%s" % (tag, k)], requires_grad=True) for k in ("rgb_coarse", "alpha_coarse", "depth_coarse", "weight_coarse")}
        tgt = {k: torch.tensor(g["%s_tgt_%s" % (tag, k)]) for k in ("rgb", "alpha")}
        out = loss(pred, tgt)
        out["loss"].backward()
        keys = [k[len(tag) + 1:] for k in g.files if k.startswith(tag + "_") and not k.startswith(tag + "_in_") and not k.startswith(tag + "_grad_") and not k.startswith(tag + "_tgt_")]
        assert "loss" in keys and "reg_density" in keys and (tag == "nerf" or "loss_depth_reg" in keys)
        for k in keys:
            assert abs(float(out[k].detach()) - float(g["%s_%s" % (tag, k)])) <= 2e-6 * max(1.0, abs(float(g["%s_%s" % (tag, k)]))), (tag, k)
        for k, v in pred.items():
            ref = g["%s_grad_%s" % (tag, k)]
            got = v.grad.numpy() if v.grad is not None else np.zeros_like(ref)
            assert np.abs(got - ref).max() <= 2e-6 * max(1e-3, np.abs(ref).max()), (tag, k)


def test_checkpoint_surface_equals_reference_modules_state_dict():
    """The keys, shapes and dtypes a checkpoint of the path carries: the product's modules against what the REFERENCE's
    modules register (state_dict() of the reference model built by tests/golden/ref_cpu_harness.py; the two tcnn vectors are
    stand-ins there, their sizes are covered by test_tcnn_param_vector_sizes_and_loader_messages)."""
    from instantavatar_amd.models.networks.ngp import NeRFNGPNet
    from instantavatar_amd.pipeline import AvatarModel
    from instantavatar_amd.renderers.raymarcher_acc import Raymarcher
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pipeline_golden.npz"))
    ref = {}
    for row in g["E_state_dict"]:
        k, shape, dtype = str(row).split("|")
        ref[k] = (shape, dtype)
    renderer = Raymarcher(256, 291600)
    renderer.initialize(1)                                   # as DNeRFModel.__init__ does (DNeRF.py:28): the training grid exists ...
    model = AvatarModel(None, NeRFNGPNet(dict(center=[0, 0, 0], scale=[1, 1, 1])), renderer)
    assert len(renderer.density_grid_train_all) == 1 and renderer.density_grid_train is renderer.density_grid_train_all[0]
    got = {k: ("x".join(str(d) for d in v.shape), str(v.dtype)) for k, v in model.state_dict().items()}   # ... but is not checkpointed
    assert set(got) == set(ref), (sorted(set(got) ^ set(ref)))
    for k in ref:
        if k.endswith(".params"):
            assert got[k][1] == ref[k][1]                 # flat fp32 vectors, sizes differ from the 1-element stand-ins
        else:
            assert got[k] == ref[k], (k, got[k], ref[k])


def test_optimizer_groups_and_lr_schedule_equal_reference_configure_optimizers():
    """training.configure_optimizer / configure_scheduler against DNeRFModel.configure_optimizers of the reference executing
    (pipeline golden, part F): three Adam groups -- hash encoding, the rest, the (here empty) SMPL tables with their own
    learning rate -- same hyper-parameters, and the per-epoch learning rate of the LambdaLR (1 - epoch / max_epochs) ** 1.5."""
    from instantavatar_amd.models.networks.ngp import NeRFNGPNet
    from instantavatar_amd.pipeline import AvatarModel
    from instantavatar_amd.renderers.raymarcher_acc import Raymarcher
    from instantavatar_amd.training import configure_optimizer, configure_scheduler
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pipeline_golden.npz"))
    model = AvatarModel(None, NeRFNGPNet(dict(center=[0, 0, 0], scale=[1, 1, 1])), Raymarcher(256, 291600))
    opt = configure_optimizer(model, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, smpl_lr=5e-4)
    names = {id(p): n for n, p in model.named_parameters()}
    got = ["%s|%g|%s|%g" % (",".join(names[id(p)] for p in grp["params"]), grp["lr"], tuple(grp["betas"]), grp["eps"]) for grp in opt.param_groups]
    assert got == [str(r) for r in g["F_groups"]], (got, list(g["F_groups"]))
    sched = configure_scheduler(opt, max_epochs=30)
    lrs = []
    for epoch in range(31):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
    assert np.allclose(lrs, g["F_lr_per_epoch"], rtol=1e-12, atol=0)


def test_new_entry_points_validate_arguments_before_any_launch():
    """ia_precompute_ws / ia_patch_corners / ia_near_far reject bad arguments with a message and never reach a launch
    (callable without a GPU for exactly that reason)."""
    from instantavatar_amd import _lib
    L = _lib.lib()
    g = _lib.SnarfGrid()
    g.D, g.H, g.W = 8, 32, 30          # W not a multiple of 4
    assert L.ia_precompute_workspace_bytes(C.byref(g)) > 0
    assert L.ia_precompute_ws(None, None, None, None, None, C.byref(g), None, 0, None) != 0 and b"null pointer" in L.ia_last_error()
    one = C.c_void_p(16)
    assert L.ia_precompute_ws(one, one, one, None, None, C.byref(g), None, 0, None) != 0 and b"multiple of 4" in L.ia_last_error()
    assert L.ia_patch_corners(None, None, None, 4, 64, 64, 16, C.c_float(1.0), None, None, None) != 0 and b"ia_patch_corners" in L.ia_last_error()
    assert L.ia_patch_corners(one, one, one, 4, 16, 64, 16, C.c_float(1.0), one, one, None) != 0          # H <= patch
    assert L.ia_patch_corners(one, one, one, 0, 64, 64, 16, C.c_float(1.0), one, one, None) == 0          # n == 0: nothing to do
    assert L.ia_near_far(None, 5, None, None, None) != 0 and b"ia_near_far" in L.ia_last_error()
    assert L.ia_near_far(None, 0, None, None, None) == 0


def test_round3_entry_points_validate_arguments_before_any_launch():
    """ia_snarf_search_compact_jinv / ia_snarf_implicit_bwd_compact / ia_smpl_tfs_bwd / ia_frame_stats / ia_profile_get_units
    reject bad arguments with a message and never reach a launch (so this runs without a GPU)."""
    from instantavatar_amd import _lib
    L = _lib.lib()
    g = _lib.SnarfGrid()
    g.D, g.H, g.W = 8, 32, 32
    one = C.c_void_p(16)
    bones = _lib.bone_array([0, 1, 2])
    tail = (C.c_float(1e-5), C.c_float(1e-1))
    # n_cand is required; P < 0 is rejected; P == 0 returns before touching anything; the J_inv output and the workspace are required
    big = 1 << 20
    assert L.ia_snarf_search_jinv_workspace_bytes(4, 3) == 4 * 3 * 9 * 4
    assert L.ia_snarf_search_compact_jinv(one, 4, None, one, one, bones, 3, C.byref(g), *tail, one, one, 16, one, one, None, 0, one, big, None) != 0
    assert b"n_cand" in L.ia_last_error()
    assert L.ia_snarf_search_compact_jinv(one, -1, None, one, one, bones, 3, C.byref(g), *tail, one, one, 16, one, one, one, 0, one, big, None) != 0
    assert L.ia_snarf_search_compact_jinv(None, 0, None, None, None, bones, 3, C.byref(g), *tail, None, None, 0, None, None, one, 0, None, 0, None) == 0
    assert L.ia_snarf_search_compact_jinv(one, 4, None, one, one, bones, 3, C.byref(g), *tail, one, None, 16, one, one, one, 0, one, big, None) != 0
    assert b"null pointer" in L.ia_last_error()
    assert L.ia_snarf_search_compact_jinv(one, 4, None, one, one, bones, 3, C.byref(g), *tail, one, one, 16, one, one, one, 0, one, 8, None) != 0
    assert b"workspace" in L.ia_last_error()
    assert L.ia_snarf_search_compact_jinv(one, 4, None, one, one, bones, 99, C.byref(g), *tail, one, one, 16, one, one, one, 0, one, big, None) != 0
    assert b"n_init" in L.ia_last_error()
    # compact implicit backward: the device-side count is mandatory, the workspace is checked
    assert L.ia_snarf_implicit_bwd_compact(one, one, one, 100, None, one, 1, C.byref(g), one, one, 1 << 20, None) != 0
    assert b"n_cand" in L.ia_last_error()
    assert L.ia_snarf_implicit_bwd_compact(one, one, one, 100, one, one, 1, C.byref(g), one, one, 8, None) != 0
    assert b"workspace" in L.ia_last_error()
    assert L.ia_snarf_implicit_bwd_compact(one, one, one, 0, one, one, 1, C.byref(g), one, one, 0, None) == 0     # nothing to do
    assert L.ia_smpl_tfs_bwd(None, None, None, None, None, None, None, None, None) != 0 and b"ia_smpl_tfs_bwd" in L.ia_last_error()
    assert L.ia_frame_stats(None, None, 0, None, None) != 0 and b"ia_frame_stats" in L.ia_last_error()
    u = (C.c_uint64 * 3)()
    assert L.ia_profile_get_units(7, u, 3) != 0 and L.ia_profile_get_units(0, u, 9) != 0
    assert L.ia_profile_get_units(0, u, 3) == 0 and list(u) == [0, 0, 0]                                           # profiling never enabled
    # device self-tests of the shared-reciprocal division: null pointers / a ragged wave are refused, n = 0 is a no-op
    assert L.ia_selftest_shared_rcp(None, None, 4, None, None, None) != 0 and b"ia_selftest_shared_rcp" in L.ia_last_error()
    assert L.ia_selftest_shared_rcp(None, None, 0, None, None, None) == 0
    assert L.ia_selftest_jinv_update(one, one, one, 65, one, one, one, None) != 0 and b"n % 64" in L.ia_last_error()
    assert L.ia_selftest_jinv_update(None, None, None, 0, None, None, None, None) == 0


def test_affine_inverse_equals_torch_inverse_value_and_gradient():
    """snarf_deformer.affine_inverse (cofactor inverse of [M t; 0 0 0 1] in tensor ops: no LU status read-back, capturable)
    against torch.inverse: value to rounding, gradient w.r.t. the three variable rows."""
    from instantavatar_amd.deformers.snarf_deformer import affine_inverse
    g = torch.Generator().manual_seed(0)
    A = torch.eye(4, dtype=torch.float64).repeat(6, 1, 1)
    A[:, :3, :3] = torch.linalg.qr(torch.randn(6, 3, 3, generator=g, dtype=torch.float64))[0] + 0.05 * torch.randn(6, 3, 3, generator=g, dtype=torch.float64)
    A[:, :3, 3] = 5 * torch.randn(6, 3, generator=g, dtype=torch.float64)
    a, b = A.clone().requires_grad_(True), A.clone().requires_grad_(True)
    x, y = affine_inverse(a), torch.inverse(b)
    assert (x - y).abs().max() < 1e-12
    w = torch.randn(6, 4, 4, generator=g, dtype=torch.float64)
    (x * w).sum().backward()
    (y * w).sum().backward()
    assert (a.grad[:, :3] - b.grad[:, :3]).abs().max() < 1e-10
    x32 = affine_inverse(A.float())
    assert (x32.double() - y.detach()).abs().max() < 5e-6


def test_smpl_chain_backward_formulas_match_autograd():
    """The chain rule `k_smpl_tfs_bwd` writes out (E_j = D_j B_j^T, dA_j = R_W^T E_j, dW = sum E_j A_j^T, dA_0 -= W^T dW W^T,
    children-before-parents, Rodrigues), restated in float64 numpy and checked against autograd through the product's
    lbs.py-style torch ops: the derivation the kernel transcribes (the kernel itself is compared on the GPU:
    test_smpl_chain_backward_kernel_equals_autograd_through_lbs)."""
    from instantavatar_amd import synthetic as syn
    from instantavatar_amd.deformers.smplx import SMPL
    from instantavatar_amd.deformers.snarf_deformer import affine_inverse
    smpl = SMPL.from_dict(syn.make_body(42)).double()
    betas = torch.zeros(1, 10, dtype=torch.float64)
    rest = smpl(betas=betas, body_pose=torch.as_tensor(syn.cano_pose("A_pose"))[None].double())
    Binv = torch.inverse(rest.A)[0]
    poses, tr = syn.procedural_pose_track(8)
    pose = torch.tensor(poses[3:4]).double().requires_grad_(True)
    tau = torch.tensor(tr[3:4]).double().requires_grad_(True)
    out = smpl(betas=betas, body_pose=pose[:, 3:], global_orient=pose[:, :3], transl=tau, return_verts=False)
    tfs = (affine_inverse(out.A[:, 0])[:, None] @ out.A @ Binv)[0]
    D = torch.randn(24, 4, 4, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    D[:, 3] = 0
    (tfs * D).sum().backward()
    J, par = smpl.rest_joints(betas).numpy(), smpl.parents_list
    th, t = poses[3].astype(np.float64).reshape(24, 3), tr[3].astype(np.float64)
    R, K, KK = np.zeros((24, 3, 3)), np.zeros((24, 3, 3)), np.zeros((24, 3, 3))
    ang, sn, cs = np.zeros(24), np.zeros(24), np.zeros(24)
    rel = J.copy()
    for j in range(1, 24):
        rel[j] = J[j] - J[par[j]]
    for j in range(24):
        a = np.sqrt(((th[j] + 1e-8) ** 2).sum())
        d = th[j] / a
        K[j] = np.array([[0, -d[2], d[1]], [d[2], 0, -d[0]], [-d[1], d[0], 0]])
        KK[j] = K[j] @ K[j]
        ang[j], sn[j], cs[j] = a, np.sin(a), np.cos(a)
        R[j] = np.eye(3) + sn[j] * K[j] + (1 - cs[j]) * KK[j]
    G = np.zeros((24, 4, 4))
    for j in range(24):
        L = np.eye(4)
        L[:3, :3], L[:3, 3] = R[j], rel[j]
        G[j] = L if j == 0 else G[par[j]] @ L
    A = G.copy()
    for j in range(24):
        A[j, :3, 3] = G[j, :3, 3] - G[j, :3, :3] @ J[j] + t
    W, B, Dn = np.linalg.inv(A[0]), Binv.numpy(), D.numpy()
    E = np.einsum("jaq,jbq->jab", Dn[:, :3, :], B)
    dA = np.einsum("qa,jqb->jab", W[:3, :3], E)
    dW = np.zeros((4, 4))
    dW[:3] = np.einsum("jaq,jbq->ab", E, A)
    dA[0] -= (W.T @ dW @ W.T)[:3]
    dRG, dg = dA[:, :, :3] - dA[:, :, 3:4] * J[:, None, :], dA[:, :, 3].copy()
    d_tau = dg.sum(0)
    dRl = np.zeros((24, 3, 3))
    for i in range(23, 0, -1):
        p = par[i]
        dRl[i] = G[p, :3, :3].T @ dRG[i]
        dRG[p] += dRG[i] @ R[i].T + np.outer(dg[i], rel[i])
        dg[p] += dg[i]
    dRl[0] = dRG[0]
    d_pose = np.zeros((24, 3))
    for j in range(24):
        dR = dRl[j]
        dK = sn[j] * dR + (1 - cs[j]) * (dR @ K[j].T + K[j].T @ dR)
        d_ang = cs[j] * (dR * K[j]).sum() + sn[j] * (dR * KK[j]).sum()
        d_dir = np.array([dK[2, 1] - dK[1, 2], dK[0, 2] - dK[2, 0], dK[1, 0] - dK[0, 1]])
        d_pose[j] = d_dir / ang[j] + (d_ang - (d_dir @ th[j]) / ang[j] ** 2) * (th[j] + 1e-8) / ang[j]
    ref = pose.grad[0].numpy()
    assert np.abs(d_pose.reshape(-1) - ref).max() < 1e-10 * max(1.0, np.abs(ref).max())
    assert np.abs(d_tau).max() < 1e-12 and np.abs(tau.grad.numpy()).max() < 1e-12      # root-frame transforms: no gradient to the translation


# --------------------------------------------------------------------------------------------------------------------
# rows a10 / a11: the tcnn pin (VERDICT r04 task 2)
def test_tcnn_golden():
    """The oracle's restatement of tiny-cuda-nn v1.6 (HashGrid layout, hashing, interpolation, fp16 rounding; the two fully
    fused MLPs) against outputs of tiny-cuda-nn itself: tests/golden/tcnn_golden.npz, written by tools/make_tcnn_golden.py on
    a box where `import tinycudann` works.  Absent -> XFAIL "parity unpinned" (tcnn is not installable here)."""
    import tcnn_golden as TG
    g = TG.load()
    if g is None or not g["is_pin"]:
        pytest.xfail(TG.UNPINNED)
    from oracle import oracle as orc
    orc.build()
    res = TG.check_oracle(g, orc)
    print("tcnn golden (%s): %s" % (g["meta"], res))
    TG.assert_oracle(res)


@pytest.mark.parametrize("r3", [54, 55])
def test_tcnn_golden_consumer_runs_on_a_self_made_file(tmp_path, r3):
    """The consumer itself, exercised on a record the CPU oracle wrote (tools/make_tcnn_golden.py --self-made) in either level-3
    layout: the regenerated parameters, the layout decision from `n_enc`, the feature and MLP comparisons all run and agree
    exactly -- and the record is NOT accepted as the pin."""
    import tcnn_golden as TG
    from oracle import oracle as orc
    orc.build()
    t = TG.tool()
    path = str(tmp_path / "self.npz")
    np.savez_compressed(path, **t.run_oracle(t.golden_points(), level3_res=r3))
    g = TG.load(path)
    assert g is not None and not g["is_pin"] and g["points"].shape == (4096, 3) and g["feat"].shape == (4096, 32)
    pts = g["points"]
    assert (pts == 0).any() and (pts == 1).any() and len(np.unique(pts, axis=0)) > 4000          # corners / faces are in, no degenerate set
    res = TG.check_oracle(g, orc)
    assert res["level3_res"] == r3 and res["feat_mismatch"] == 0 and res["enc_out_max_rel_fp32"] == 0.0 and res["col_out_max_abs_fp32"] < 1e-3
    TG.assert_oracle(res)
    with pytest.raises(SystemExit):
        t.main(["--self-made", TG.PIN_PATH])                                                       # a self-made file cannot take the pin's place


def test_pack_rgba8_oracle_is_the_reference_expression():
    """oracle.pack_rgba8 == `(img.cpu().numpy() * 255).astype(np.uint8)` of animate.py:107-113 on images inside [0, 1] (where
    the reference's expression is defined), clamps outside, and truncates (254.999 -> 254, k / 255 -> k)."""
    from oracle import oracle as orc
    rs = np.random.RandomState(3)
    rgb, alpha = rs.rand(17, 9, 3).astype(np.float32), rs.rand(17, 9).astype(np.float32)
    img = np.concatenate([rgb, alpha[..., None]], -1)
    assert np.array_equal(orc.pack_rgba8(rgb, alpha), (img * 255).astype(np.uint8))
    lv = (np.arange(256, dtype=np.float32) / np.float32(255))
    got = orc.pack_rgba8(np.stack([lv, np.nextafter(lv, np.float32(-1)), np.nextafter(lv, np.float32(2))], -1), lv)
    assert np.array_equal(got[:, 0], (lv * np.float32(255)).astype(np.uint8)) and np.array_equal(got[:, 3], got[:, 0])
    assert (np.abs(got[:, 0].astype(int) - np.arange(256)) <= 1).all() and got[0, 1] == 0 and got[255, 2] == 255      # clamped below 0 / above 1
    assert orc.pack_rgba8(np.float32([[1.5, -0.5, 0.999999]]), np.float32([2.0]))[0].tolist() == [255, 0, 254, 255]


def test_round5_entry_points_validate_arguments_before_any_launch():
    """ia_pack_rgba8 / ia_snarf_inverse_skinning[_bwd] / ia_expand_candidate_points: empty inputs are no-ops, bad arguments
    are refused with a message, nothing reaches a launch (so this runs without a GPU)."""
    from instantavatar_amd import _lib
    L = _lib.lib()
    g = _lib.SnarfGrid()
    g.D, g.H, g.W = 8, 32, 32
    one, odd = C.c_void_p(16), C.c_void_p(18)
    assert L.ia_pack_rgba8(None, None, 0, None, None) == 0                                               # an empty frame
    assert L.ia_pack_rgba8(one, one, -1, one, None) != 0 and b"R < 0" in L.ia_last_error()
    assert L.ia_pack_rgba8(one, None, 4, one, None) != 0 and b"null pointer" in L.ia_last_error()
    assert L.ia_pack_rgba8(one, one, 4, odd, None) != 0 and b"aligned" in L.ia_last_error()             # one 32-bit store per ray
    assert L.ia_snarf_inverse_skinning(None, None, None, 13, None, 0, None, None, 1, C.byref(g), None, None, None) == 0
    assert L.ia_snarf_inverse_skinning(one, one, None, 13, None, -1, None, one, 1, C.byref(g), one, one, None) != 0 and b"n < 0" in L.ia_last_error()
    assert L.ia_snarf_inverse_skinning(one, one, None, 0, None, 8, None, one, 1, C.byref(g), one, one, None) != 0       # neither cand_pt nor n_init
    assert L.ia_snarf_inverse_skinning(one, None, None, 13, None, 8, None, one, 1, C.byref(g), one, one, None) != 0 and b"null pointer" in L.ia_last_error()
    big = 1 << 20
    assert L.ia_snarf_inverse_skinning_bwd(None, None, None, 13, None, None, 0, None, None, 1, C.byref(g), None, None, None, None, 0, None) == 0
    assert L.ia_snarf_inverse_skinning_bwd(one, one, None, 13, None, one, 8, None, one, 1, C.byref(g), one, one, None, one, big, None) != 0   # no mask, no count
    assert L.ia_snarf_inverse_skinning_bwd(one, one, None, 13, one, one, 8, None, one, 1, C.byref(g), one, one, None, one, 8, None) != 0
    assert b"workspace" in L.ia_last_error()
    assert L.ia_snarf_inverse_skinning_bwd(one, None, None, 13, one, one, 8, None, one, 1, C.byref(g), one, one, None, one, big, None) != 0
    assert b"version-2" in L.ia_last_error()
    assert L.ia_expand_candidate_points(None, None, 0, None, None, 0, None) == 0
    assert L.ia_expand_candidate_points(one, None, 4, None, one, 8, None) != 0 and b"null pointer" in L.ia_last_error()
    assert L.ia_expand_candidate_points(one, one, -1, None, one, 8, None) != 0


def test_search_kernel_isa_has_no_dpp_read_after_valu_write_hazard():
    """k_search issues `v_add_u32_dpp` from inline asm, which the compiler's hazard recogniser cannot see into; the
    guard (`s_nop 1` and the eight DPP adds as ONE asm statement) is structural, and this checks the result: in the gfx950
    assembly of ia_search.hip no DPP instruction reads a VGPR that a VALU instruction wrote within the two preceding wait
    states (tools/analyse_search_isa.py: dpp_hazards, which is first shown to catch a planted hazard)."""
    import importlib.util
    from instantavatar_amd import build
    if not build.have_compiler():
        pytest.skip("hipcc not available")
    spec = importlib.util.spec_from_file_location("analyse_search_isa", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "analyse_search_isa.py"))
    A = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(A)
    dpp = "  v_add_u32_dpp v7, v5, v3 quad_perm:[0,0,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1"
    assert len(A.dpp_hazards(["  v_add_u32_e32 v5, v1, v2", dpp])) == 1
    assert len(A.dpp_hazards(["  v_add_u32_e32 v5, v1, v2", "  v_mov_b32_e32 v9, v1", dpp])) == 1          # one wait state is not enough
    assert A.dpp_hazards(["  v_add_u32_e32 v5, v1, v2", "  s_nop 1", dpp]) == []
    assert A.dpp_hazards(["  v_add_u32_e32 v6, v1, v2", dpp]) == []                                          # another register
    isa = A.device_isa()
    n_dpp = sum(1 for l in isa if "_dpp" in l or "quad_perm:" in l)
    assert n_dpp >= 72, n_dpp                                                                                 # 3 kernels x 24 DPP adds at least
    assert A.dpp_hazards(isa) == []


def test_adam_oracle_matches_torch_adam_over_twenty_steps_with_a_skipped_one(oracle):
    """oracle.adam_step (what `ia_adam_step` is checked against on the GPU) vs `torch.optim.Adam` on the CPU -- the optimiser of
    DNeRF.py:46-50 with the reference's three parameter groups and hyper-parameters -- over 20 steps of which one carries an inf
    gradient (GradScaler's skip, DNeRF.py:151-154: parameters, moments and step counters untouched).  Moments are bit-equal;
    parameters agree to the rounding of torch's CPU sqrt (Sleef's, not correctly rounded: ~0.5 % of the elements differ in the
    last bit of one step's update)."""
    rs = np.random.RandomState(0)
    shapes, lrs = [(40003,), (64, 16), (7, 72)], [1e-2, 1e-2, 1e-5]
    P = [torch.nn.Parameter(torch.as_tensor(rs.randn(*s).astype(np.float32) * 0.1)) for s in shapes]
    opt = torch.optim.Adam([{"params": [P[0]]}, {"params": [P[1]]}, {"params": [P[2]], "lr": lrs[2]}], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    p = [q.detach().numpy().copy() for q in P]
    st = [dict(step=0.0, exp_avg=np.zeros(s, np.float32), exp_avg_sq=np.zeros(s, np.float32)) for s in shapes]
    n_steps = 20
    for k in range(n_steps):
        g = [(rs.randn(*s) * 10.0 ** rs.uniform(-7, 0)).astype(np.float32) for s in shapes]
        for a in g:
            a.reshape(-1)[::5] = 0
        bad = k == 7
        if bad:
            g[1].reshape(-1)[3] = np.inf
        before = [a.copy() for a in p]
        assert oracle.adam_step(p, [a.copy() for a in g], st, lrs) == bad
        if bad:
            assert all(np.array_equal(a, b) for a, b in zip(p, before)) and st[0]["step"] == 7.0
            continue
        for q, a in zip(P, g):
            q.grad = torch.as_tensor(a.copy())
        opt.step()
    for q, a, s, lr in zip(P, p, st, lrs):
        assert float(opt.state[q]["step"]) == s["step"] == n_steps - 1
        assert np.array_equal(opt.state[q]["exp_avg"].numpy(), s["exp_avg"]) and np.array_equal(opt.state[q]["exp_avg_sq"].numpy(), s["exp_avg_sq"])
        d = np.abs(q.detach().numpy() - a)
        assert (d <= n_steps * lr * 2.4e-7 + 1.2e-7 * np.abs(a)).all() and (d > 0).mean() < 0.02, (float(d.max()), float((d > 0).mean()))
