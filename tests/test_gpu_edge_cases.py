"""Edge cases of the C ABI on the GPU (-m gpu): empty / ragged inputs, argument errors,
non-finite inputs, rays that miss everything, state-dict surface."""
import ctypes as C

import os

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
def gw():
    model, body, fp, init = W.build(DEV, 64, 16)
    poses, tr = W.poses()
    model.deformer.prepare_deformer(make_batch(DEV, 32, poses[1], tr[1]))
    return model, poses, tr


def test_argument_errors_raise_and_report(gw):
    model = gw[0]
    L = _lib.lib()
    with pytest.raises(_lib.IAError) as e:
        _lib.check(L.ia_field_fwd(None, 10, None, C.byref(model.net_coarse.field_desc()), None, None, None), "ia_field_fwd")
    assert "null pointer" in str(e.value)
    g = _lib.SnarfGrid(); g.D, g.H, g.W = 8, 32, 30  # W not a multiple of 4
    t = torch.zeros(16, device=DEV)
    with pytest.raises(_lib.IAError):
        _lib.check(L.ia_precompute(_lib.ptr(t), _lib.ptr(t), _lib.ptr(t), None, None, C.byref(g), None), "ia_precompute")
    bones = _lib.bone_array([0, 1, 99])  # joint id out of range
    with pytest.raises(_lib.IAError):
        _lib.check(L.ia_snarf_search(_lib.ptr(t), 1, _lib.ptr(t), _lib.ptr(t), bones, 3, C.byref(model.deformer.deformer.grid_desc()),
                                     1e-5, 1e-1, _lib.ptr(t), _lib.ptr(t), None, None, None), "ia_snarf_search")
    ws = torch.empty(16, dtype=torch.uint8, device=DEV)  # workspace too small
    rc = L.ia_occupancy_from_density(_lib.ptr(torch.zeros(64 ** 3, device=DEV)), 64, _lib.ptr(torch.zeros(8193, dtype=torch.int32, device=DEV)),
                                     None, _lib.ptr(ws), 16, None)
    assert rc == -3 and b"workspace" in L.ia_last_error()
    with pytest.raises(_lib.IAError):
        model.net_coarse(torch.zeros(4, 3), None)  # CPU tensor: no fallback


def test_empty_and_ragged_sizes(gw):
    model = gw[0]
    net, dfm = model.net_coarse, model.deformer
    with torch.no_grad():
        r, s = net(torch.zeros((0, 3), device=DEV), None)
    assert r.shape == (0, 3) and s.shape == (0,)
    bb = dfm.bbox
    g = torch.Generator(device=DEV).manual_seed(0)
    for n in (1, 2, 63, 64, 65, 127, 129, 1000, 4097):
        x = torch.rand((n, 3), device=DEV, generator=g) * (bb[1] - bb[0]) + bb[0]
        with torch.no_grad():
            r1, s1 = net(x, None)
            r2, s2 = net(torch.cat([x, x.flip(0)]), None)  # other tile composition, same samples
        assert torch.equal(r1, r2[:n]) and torch.equal(s1, s2[:n])
        assert torch.equal(r1, r2[n:].flip(0)) and torch.equal(s1, s2[n:].flip(0))
        rq, sq = dfm(x, net, True)                     # fused query
        rc, sc = dfm(x, lambda p, d: net(p, d), True)  # generic route
        assert torch.equal(sq, sc) and torch.equal(rq, rc)
    rq, sq = dfm(torch.zeros((0, 3), device=DEV), lambda p, d: net(p, d), True)
    assert sq.shape == (0,)


def test_non_finite_inputs_do_not_fault(gw):
    model = gw[0]
    x = torch.tensor([[float("nan"), 0, 0], [float("inf"), 1, 1], [-float("inf"), 0, 0], [1e30, -1e30, 0], [0.0, 0.0, 0.0]], device=DEV)
    with torch.no_grad():
        r, s = model.net_coarse(x, None)
        rq, sq = model.deformer(x, model.net_coarse, True)
    torch.cuda.synchronize()
    assert torch.isfinite(s[1:]).all() and torch.isfinite(sq).all()   # nan_to_num at test time (snarf_deformer.py:136-137)
    assert (sq >= 0).all()


def test_density_grid_init_small_workspace_route_equals_batched(gw):
    """`ia_density_grid_init` with the workspace of `ia_density_init_workspace_bytes` (one probe set per launch) against the
    batched workspace (all sets in one launch, what the Python caller hands over by default): the header promises the same
    result.  ADVICE r05: the probe-cell permutation (Morton order) had only been applied to the batched route's maximum, the
    small route scattered cell perm(p)'s density to cell p."""
    model, poses, tr = gw
    grid = model.renderer.density_grid_test
    jit = torch.as_tensor(np.random.RandomState(9).rand(3, 64 ** 3, 3).astype(np.float32), device=DEV)
    model.deformer.prepare_deformer(make_batch(DEV, 64, poses[3], tr[3]))
    out = {}
    for batched in (True, False):
        grid.batched_probes = batched
        try:
            grid.initialize(model.deformer, model.net_coarse, jitter=jit)
        finally:
            grid.batched_probes = True
        out[batched] = (grid.density_probe.clone(), grid.density_field.clone())
    assert int(out[True][1].sum()) > 500
    assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][1], out[False][1])


def test_rays_that_miss_everything(gw):
    model, poses, tr = gw
    res = 32
    batch = make_batch(DEV, res, poses[1], tr[1])
    batch["rays_d"] = -batch["rays_d"]  # look away from the body
    rgb, depth, alpha, counter = model.render_image_fast(batch, (res, res))
    assert torch.all(alpha == 0) and torch.all(counter == 0) and torch.allclose(rgb, torch.ones_like(rgb))
    assert model.renderer.last_iters >= 2


def test_rays_with_an_empty_or_inverted_depth_range_take_no_sample(gw):
    """far <= near gives a march step <= 0: the reference's loop `while (t < far && cnt < N)` would never advance on such a ray
    (raymarcher.cu:44-69).  Here it takes no sample and dies in the first compaction; the other rays of the frame are untouched."""
    from instantavatar_amd.models.structures.utils import Rays
    model, poses, tr = gw
    res = 48
    batch = make_batch(DEV, res, poses[1], tr[1])
    ref = [t.clone() for t in model.render_image_fast(batch, (res, res))]      # prepares the deformer and the occupancy grid
    rays = Rays(o=batch["rays_o"], d=batch["rays_d"], near=batch["near"], far=batch["far"])
    model.deformer.transform_rays_w2s(rays)
    near, far = rays.near.clone(), rays.far.clone()
    empty = torch.zeros_like(near, dtype=torch.bool); empty.view(-1)[::7] = True
    inverted = torch.zeros_like(near, dtype=torch.bool); inverted.view(-1)[3::11] = True
    far[empty] = near[empty]
    far[inverted] = near[inverted] - 0.5
    rays.near, rays.far = near, far
    d = model.renderer.render_test_fused(rays, model.deformer, model.net_coarse)      # (`ia_render_test`: the fused wave-front loop)
    torch.cuda.synchronize()
    bad = (empty | inverted).reshape(-1)
    alpha, counter = d["alpha_coarse"].reshape(-1), d["counter_coarse"].reshape(-1)
    assert torch.all(alpha[bad] == 0) and torch.all(counter[bad] == 0)
    # (the other rays: the same samples, the same colours -- only the wave-front batching differs with fewer rays alive)
    good = ~bad
    assert torch.allclose(d["rgb_coarse"].reshape(-1, 3)[good], ref[0].reshape(-1, 3)[good], atol=1e-5)
    assert torch.equal(alpha[good] > 0.5, ref[2].reshape(-1)[good] > 0.5) and float((alpha[good] > 0.5).float().mean()) > 0.02


def test_large_render_1024(gw):
    """BASELINE configs[4] image size: 1024x1024 = 1 048 576 rays (> MAX_BATCH_SIZE: N_step = 1 until rays retire)."""
    model, poses, tr = gw
    res = 1024
    rgb, depth, alpha, counter = model.render_image_fast(make_batch(DEV, res, poses[2], tr[2]), (res, res))
    cov = (alpha > 0.5).float().mean().item()
    assert 0.03 < cov < 0.5 and torch.isfinite(rgb).all()
    # downsampled silhouette agrees with a 256^2 render of the same frame
    rgb2, _, alpha2, _ = model.render_image_fast(make_batch(DEV, 256, poses[2], tr[2]), (256, 256))
    a_ds = torch.nn.functional.avg_pool2d(alpha[None], 4)[0]
    assert ((a_ds > 0.5) != (alpha2 > 0.5)).float().mean() < 0.02


def test_state_dict_surface_matches_reference_keys(gw):
    """Checkpoint keys on the path: names and shapes of the reference's tcnn modules."""
    model = gw[0]
    sd = model.state_dict()
    grid_params = 13026992 if os.environ.get("IA_TCNN_LEVEL3_RES", "54") == "54" else 13044816   # tcnn level-3 resolution 54 / 55
    assert sd["net_coarse.encoder.params"].shape == (3072 + grid_params,)
    assert sd["net_coarse.color_net.params"].shape == (6144,)
    for k in ("net_coarse.center", "net_coarse.scale", "renderer.density_grid_test.density_cached",
              "renderer.density_grid_test.density_field"):
        assert k in sd, k
    assert sd["renderer.density_grid_test.density_field"].dtype == torch.bool
    # round trip + occupancy bits refresh
    model.renderer.density_grid_test.load_state_dict(model.renderer.density_grid_test.state_dict())
    model.renderer.density_grid_test.pack_bits()
    names = [n for n, _ in model.named_parameters()]
    assert any("encoder" in n for n in names)  # optimiser grouping of DNeRF.py:42-45


def test_animate_driver_writes_frames(tmp_path):
    """PL-free animate driver (animate.py equivalent): synthetic avatar, 3 frames at 135x135 -> RGBA PNGs
    + GIF; the frames must equal render_image_fast on the same batches."""
    from PIL import Image
    from instantavatar_amd.drivers import animate
    out = str(tmp_path / "anim")
    assert animate.main(["--synthetic", "--max-frames", "3", "--downscale", "8", "--out", out]) == 0
    files = sorted(os.listdir(out))
    assert files == ["0.png", "1.png", "2.png", "animation.gif"]
    im = np.asarray(Image.open(os.path.join(out, "1.png")))
    assert im.shape == (135, 135, 4) and im.dtype == np.uint8
    assert (im[..., 3] > 128).mean() > 0.02          # the body covers part of the frame
    assert Image.open(os.path.join(out, "animation.gif")).n_frames == 3


def test_novel_view_driver_turns_the_body(tmp_path):
    """PL-free novel_view driver (novel_view.py equivalent): synthetic avatar, 4 steps of the turn about y at 135 x 135 ->
    RGBA PNGs + GIF; the frames equal render_image_fast on the same batches and differ from each other (the body turns)."""
    from PIL import Image
    from instantavatar_amd.drivers import novel_view
    from instantavatar_amd.pipeline import build_synthetic_model
    out = str(tmp_path / "rot")
    assert novel_view.main(["--synthetic", "--frames", "4", "--downscale", "8", "--out", out]) == 0
    assert sorted(os.listdir(out)) == ["0.png", "1.png", "2.png", "3.png", "rotation.gif"]
    ims = [np.asarray(Image.open(os.path.join(out, "%d.png" % i))) for i in range(4)]
    assert all(im.shape == (135, 135, 4) and im.dtype == np.uint8 for im in ims)
    assert all((im[..., 3] > 128).mean() > 0.02 for im in ims)
    assert (ims[0] != ims[1]).mean() > 0.01 and (ims[0] != ims[2]).mean() > 0.01
    assert Image.open(os.path.join(out, "rotation.gif")).n_frames == 4
    model, _, _ = build_synthetic_model(DEV)
    model.eval()
    seq = novel_view.RotationSequence(4, np.zeros(10, np.float32), torch.device(DEV), downscale=8)
    with torch.no_grad():
        rgb, _, alpha, _ = model.render_image_fast(seq.batch(1), (seq.H, seq.W))
    want = (torch.cat([rgb, alpha[..., None]], -1)[0].clamp(0, 1) * 255).to(torch.uint8).cpu().numpy()
    want = want[..., [2, 1, 0, 3]]      # the model's channels are (B, G, R): the file holds R first (cv2.imwrite semantics)
    assert (np.abs(want.astype(int) - ims[1].astype(int)) > 1).mean() < 1e-3


def test_eval_driver_refines_renders_and_measures(tmp_path, capsys):
    """PL-free eval driver (eval.py equivalent) on the synthetic subject: the SMPL tables of the test frames are refined with
    everything else frozen (the field's parameters must not move, the tables must), the test frames are rendered WITH the
    refined tables (render_image_fast's is_refine branch, DNeRF.py:73-86) into [gt | rendering | error map] PNGs, and
    results.txt holds PSNR / SSIM of the middle panel against the left one, recomputed here from the files."""
    from PIL import Image
    from instantavatar_amd import evaluation as ev
    from instantavatar_amd.drivers import eval as eval_driver
    from instantavatar_amd.utils.metrics import psnr, ssim
    out = str(tmp_path / "eval")
    assert eval_driver.main(["--synthetic", "--frames", "3", "--res", "96", "--epochs", "2", "--check-val-every-n-epoch", "2", "--out", out]) == 0
    text = capsys.readouterr().out
    assert text.count("val/rgb_loss") == 1 and "epoch 1  val/rgb_loss" in text          # one validation run (and scheduler step), after the 2nd epoch
    assert "refined 3 frames in 6 steps (field parameters untouched)" in text and "LPIPS: --" in text     # (the driver raises if the tables stay put)
    files = sorted(os.listdir(os.path.join(out, "test")))
    assert files == ["0.png", "1.png", "2.png"]
    ps, ss = [], []
    for f in files:
        im = np.asarray(Image.open(os.path.join(out, "test", f)))
        assert im.shape == (96, 288, 3) and im.dtype == np.uint8
        gt, pred = torch.tensor(im[:, :96].copy()).float() / 255, torch.tensor(im[:, 96:192].copy()).float() / 255
        assert (gt < 0.99).any(-1).float().mean() > 0.02                      # a body in front of the white background
        ps.append(float(psnr(pred, gt)))
        ss.append(float(ssim(pred.permute(2, 0, 1)[None], gt.permute(2, 0, 1)[None])))
    lines = open(os.path.join(out, "results.txt")).read().splitlines()
    assert lines == ["PSNR: %.2f" % np.mean(ps), "SSIM: %.4f" % np.mean(ss)]
    print("eval driver on the synthetic subject: PSNR %s -> %.2f, SSIM %s -> %.4f" % (np.round(ps, 2), np.mean(ps), np.round(ss, 4), np.mean(ss)))
    # measured on the MI355X (round 4, three runs, identical): PSNR 25.09 / 25.45 / 24.24 -> 24.93, SSIM 0.9238 / 0.9265 / 0.9124 -> 0.9209
    # (0.03 rad off a pose the field knows, two refinement epochs: close, not equal); round 3 had loosened these bounds to 15 / 0.7
    # without a measurement
    assert np.mean(ps) > 22 and np.mean(ss) > 0.88


def test_render_image_fast_takes_the_refined_smpl_tables_when_is_refine():
    """DNeRF.py:73-86: with `SMPL_param` tables and is_refine the frame is rendered with the TABLE's row `idx` (global_orient,
    body_pose, transl) and near / far from the refined translation, whatever the batch carries; without is_refine the batch wins."""
    from instantavatar_amd.models.structures.body_model_param import SMPLParamEmbedding
    from instantavatar_amd.pipeline import build_synthetic_model
    model, _, _ = build_synthetic_model(DEV)
    model.eval()
    poses, tr = syn.procedural_pose_track(8)
    res = 64
    tables = dict(betas=np.zeros((1, 10), np.float32), global_orient=poses[:4, :3].copy(), body_pose=poses[:4, 3:].copy(), transl=tr[:4].copy())
    model.SMPL_param = SMPLParamEmbedding(**{k: torch.as_tensor(v.copy()) for k, v in tables.items()}).to(DEV)
    J = torch.rand((5, 64 ** 3, 3), device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))   # the occupancy probes' jitter, fixed
    with torch.no_grad():
        ref2, _, a2, _ = model.render_image_fast(make_batch(DEV, res, poses[2], tr[2]), (res, res), jitter=J)
        ref0, _, a0, _ = model.render_image_fast(make_batch(DEV, res, poses[0], tr[0]), (res, res), jitter=J)
        b = make_batch(DEV, res, poses[0], tr[0])                # the batch says frame 0 ...
        b["idx"] = torch.tensor([2])                             # ... the index says row 2
        got_plain, *_ = model.render_image_fast(dict(b), (res, res), jitter=J)
        model.is_refine = True
        b["near"], b["far"] = b["near"].clone(), b["far"].clone()
        got_refine, *_ = model.render_image_fast(b, (res, res), jitter=J)
    assert float((got_plain - ref0).abs().max()) < 1e-6 and float((ref0 - ref2).abs().max()) > 0.1
    assert float((got_refine - ref2).abs().max()) < 1e-6
    d2 = float(np.sqrt((tr[2] ** 2).sum()))
    assert abs(float(b["near"][0, 0]) - (d2 - 1)) < 1e-5 and abs(float(b["far"][0, 0]) - (d2 + 1)) < 1e-5


def test_train_driver_checkpoint_feeds_animate_driver(tmp_path):
    """drivers.train (synthetic targets) -> Lightning-layout checkpoint -> drivers.animate renders with
    the trained weights: the loss must fall and the round-tripped field must render a body."""
    from PIL import Image
    from instantavatar_amd.drivers import animate, checkpoint as ck, train
    from instantavatar_amd.pipeline import build_synthetic_model
    ckpt = str(tmp_path / "ck" / "last.ckpt")
    assert train.main(["--synthetic", "--steps", "60", "--res", "128", "--ckpt", ckpt]) == 0
    sd = torch.load(ckpt, weights_only=False)
    assert sd["global_step"] == 60 and "optimizer_states" in sd
    model, _, _ = build_synthetic_model(DEV)
    before = model.net_coarse.encoder.params.detach().clone()
    missing, unexpected = ck.load_checkpoint(model, ckpt, map_location=DEV)
    assert not unexpected and model.global_step == 60
    assert not torch.equal(before, model.net_coarse.encoder.params)
    # --resume continues the interrupted run: Adam moments, step counters and the LR-scheduler epoch come from the checkpoint
    # (a resumed run used to restart Adam from zero moments)
    st = sd["optimizer_states"][0]["state"]
    assert all(float(v["step"]) == 60 for v in st.values()) and "lr_schedulers" in sd
    m0 = {k: v["exp_avg"].clone() for k, v in st.items()}
    assert train.main(["--synthetic", "--steps", "8", "--res", "128", "--ckpt", ckpt, "--resume", "--steps-per-epoch", "4",
                       "--check-val-every-n-epoch", "1"]) == 0      # a validation run (and with it ONE scheduler step) after each of the 2 epochs
    sd2 = torch.load(ckpt, weights_only=False)
    assert sd2["global_step"] == 68
    st2 = sd2["optimizer_states"][0]["state"]
    assert all(float(v["step"]) == 68 for v in st2.values())       # 60 + 8: counters continued, not restarted at 8
    assert all(not torch.equal(m0[k].to(st2[k]["exp_avg"].device), st2[k]["exp_avg"]) for k in st2)
    assert sd2["lr_schedulers"][0]["last_epoch"] == sd["lr_schedulers"][0]["last_epoch"] + 2
    out = str(tmp_path / "anim")
    assert animate.main(["--synthetic", "--ckpt", ckpt, "--max-frames", "2", "--downscale", "8", "--out", out, "--no-gif"]) == 0
    im = np.asarray(Image.open(os.path.join(out, "0.png")))
    assert im.shape == (135, 135, 4) and (im[..., 3] > 128).mean() > 0.01


def test_checkpoint_of_the_other_tcnn_layout_is_adopted_and_self_checked(tmp_path):
    """a checkpoint whose `encoder.params` has the OTHER of tcnn's two possible sizes (level-3
    resolution 54 / 55) is followed by the loader (decided from numel), and the loaded field passes `self_check` (finite,
    no constant level); a vector with a level of zeros fails it."""
    from instantavatar_amd.drivers.checkpoint import load_checkpoint, save_checkpoint
    from instantavatar_amd.pipeline import build_synthetic_model
    import os as _os
    own = int(_os.environ.get("IA_TCNN_LEVEL3_RES", "54"))
    other = 55 if own == 54 else 54
    model, _, _ = build_synthetic_model(DEV, resolution=32)
    src, _, _ = build_synthetic_model(DEV, resolution=32)
    assert src.net_coarse.adopt_tcnn_layout(src.net_coarse.tcnn_encoder_sizes()[other]) == other
    with torch.no_grad():
        src.net_coarse.encoder.params.uniform_(-0.5, 0.5)
    path = str(tmp_path / "other_layout.ckpt")
    save_checkpoint(src, path)
    assert int(model.net_coarse.hash_desc.res[3]) == own
    load_checkpoint(model, path)
    assert int(model.net_coarse.hash_desc.res[3]) == other and model.tcnn_self_check["level3_res"] == other
    assert torch.equal(model.net_coarse.encoder.params, src.net_coarse.encoder.params)
    assert min(model.tcnn_self_check["feature_std_per_level"]) > 0 and model.tcnn_self_check["finite"]
    # a degenerate table (one level all zero) is caught
    net = model.net_coarse
    off = [int(o) for o in net.hash_desc.offset[:net.n_levels + 1]]
    w_end = net.sig_w1_size + 1024
    with torch.no_grad():
        net.encoder.params[w_end + 2 * off[5]:w_end + 2 * off[6]] = 0
    net.mark_updated()
    with pytest.raises(ValueError):
        net.self_check()


def test_field_follows_centre_and_scale_changed_in_place():
    """a model that has already rendered keeps host copies of centre / scale inside its kernel
    descriptor; loading other values IN PLACE (checkpoint, broadcast) must reach the kernels."""
    from instantavatar_amd.pipeline import build_synthetic_model
    model, _, _ = build_synthetic_model(DEV, resolution=32)
    net = model.net_coarse
    bb = model.deformer.bbox
    x = torch.rand((4096, 3), device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)) * (bb[1] - bb[0]) * 0.5 + bb[0] + (bb[1] - bb[0]) * 0.25
    with torch.no_grad():
        r0, s0 = net(x)
        new_c, new_s = net.center + 0.05, net.scale * 1.1
        net.center.copy_(new_c)          # what load_state_dict / broadcast_module_state do
        net.scale.copy_(new_s)
        net.mark_updated()
        r1, s1 = net(x)
    fresh, _, _ = build_synthetic_model(DEV, resolution=32)
    with torch.no_grad():
        fresh.net_coarse.center, fresh.net_coarse.scale = new_c.clone(), new_s.clone()
        fresh.net_coarse._desc = None
        r2, s2 = fresh.net_coarse(x)
    assert not torch.equal(s0, s1)
    assert torch.equal(s1, s2) and torch.equal(r1, r2)


def test_shared_reciprocal_division_is_the_ieee_division_in_its_range():
    """k_search divides the nine numerators of the Broyden update (fuse_cuda_kernel_fast.cu:23-55) by their common scalar with ONE
    reciprocal chain (`rcp_refined` / `div_shared` in ia_snarf.hip) where the reference compiles nine IEEE divisions.  Inside the
    exponent range the kernel's guard admits -- numerators zero or in [2^-98, 2^16), denominators in [2^-67, 2^22), plus zero /
    inf / NaN operands -- the two must agree BIT FOR BIT: swept here on the device over every exponent pair and random mantissas."""
    L = _lib.lib()
    rs = np.random.RandomState(0)

    def mant(n):
        m = 1.0 + rs.rand(n)
        m[rs.rand(n) < 0.05] = 1.0                                   # exact powers of two
        m[rs.rand(n) < 0.05] = 2.0 - 2.0 ** -23                      # all-ones mantissas
        return m * rs.choice([-1.0, 1.0], n)
    en, ed = np.arange(-98, 16), np.arange(-67, 22)
    EN, ED = np.meshgrid(en, ed, indexing="ij")
    reps = 96
    EN, ED = np.repeat(EN.reshape(-1), reps), np.repeat(ED.reshape(-1), reps)
    num = (mant(len(EN)) * np.exp2(EN.astype(np.float64))).astype(np.float32)
    den = (mant(len(ED)) * np.exp2(ED.astype(np.float64))).astype(np.float32)
    # the cases v_div_fixup decides: zero numerators of both signs, zero / inf / NaN denominators and numerators
    special_n = np.array([0.0, -0.0, 0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1.0, 0.0, 3.5, np.inf], np.float32)
    special_d = np.array([1.5, 2.5, -3.0, 0.0, -0.0, 2.0, 2.0, 2.0, np.inf, 0.0, np.nan, np.inf], np.float32)
    num, den = np.concatenate([num, special_n]), np.concatenate([den, special_d])
    n = len(num)
    tn, td = torch.from_numpy(num).to(DEV), torch.from_numpy(den).to(DEV)
    qs, qi = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    _lib.check(L.ia_selftest_shared_rcp(_lib.ptr(tn), _lib.ptr(td), n, _lib.ptr(qs), _lib.ptr(qi), _lib.stream()), "ia_selftest_shared_rcp")
    torch.cuda.synchronize()
    a, b = qs.cpu().numpy(), qi.cpu().numpy()
    nan = np.isnan(b)
    assert np.array_equal(np.isnan(a), nan)
    bad = (a.view(np.int32) != b.view(np.int32)) & ~nan
    assert not bad.any(), (int(bad.sum()), num[bad][:5], den[bad][:5], a[bad][:5], b[bad][:5])
    assert np.array_equal(b[:-12][::977], (num[:-12][::977].astype(np.float64) / den[:-12][::977]).astype(np.float32))   # and that IS num / den
    assert n > 900000


def test_broyden_update_with_shared_reciprocal_equals_the_update_with_nine_divisions():
    """`jinv_update` as k_search runs it (per wave: the shared reciprocal when all 64 lanes are inside its range, the compiler's
    divisions otherwise) against the same update with the compiler's divisions only, bit for bit: realistic magnitudes (every
    wave takes the shared path), magnitudes swept across the edges of the range wave by wave, degenerate rows (zero step, zero
    residual difference: s = 0 -> inf / NaN exactly as the reference produces them), and waves that one lane pushes out of range."""
    L = _lib.lib()
    rs = np.random.RandomState(1)
    n_waves = 512
    n = n_waves * 64
    Ji = (np.linalg.qr(rs.randn(n, 3, 3))[0] * (1 + 0.2 * rs.randn(n, 1, 1)) + 0.05 * rs.randn(n, 3, 3)).reshape(n, 9).astype(np.float32)
    x = (rs.randn(n, 3) * np.exp(rs.uniform(np.log(1e-6), np.log(1e-1), (n, 1)))).astype(np.float32)
    g = (rs.randn(n, 3) * np.exp(rs.uniform(np.log(1e-6), np.log(1e-1), (n, 1)))).astype(np.float32)
    wave = np.arange(n) // 64
    # waves 128..383: everything scaled by 2^k, k from -70 to 57 twice: in range, at the edge, beyond it
    k = np.where((wave >= 128) & (wave < 384), (wave - 128) % 128 - 70, 0)
    x = (x * np.exp2(k)[:, None].astype(np.float64)).astype(np.float32)
    g = (g * np.exp2(k // 2)[:, None].astype(np.float64)).astype(np.float32)
    # waves 384..447: degenerate rows sprinkled in
    deg = (wave >= 384) & (wave < 448)
    x[deg & (rs.rand(n) < 0.2)] = 0.0
    g[deg & (rs.rand(n) < 0.2)] = 0.0
    # waves 448..: ONE lane far out of range
    x[(wave >= 448) & (np.arange(n) % 64 == 17)] *= np.float32(2.0 ** 40)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    tJ, tx, tg = t(Ji), t(x), t(g)
    oa, ob = torch.empty(n, 9, device=DEV), torch.empty(n, 9, device=DEV)
    took = torch.zeros(n, dtype=torch.uint8, device=DEV)
    _lib.check(L.ia_selftest_jinv_update(_lib.ptr(tJ), _lib.ptr(tx), _lib.ptr(tg), n, _lib.ptr(oa), _lib.ptr(ob), _lib.ptr(took), _lib.stream()),
               "ia_selftest_jinv_update")
    torch.cuda.synchronize()
    a, b, took = oa.cpu().numpy(), ob.cpu().numpy(), took.cpu().numpy().astype(bool)
    nan = np.isnan(b)
    assert np.array_equal(np.isnan(a), nan)
    bad = (a.view(np.int32) != b.view(np.int32)) & ~nan
    assert not bad.any(), (int(bad.sum()), np.argwhere(bad)[:5])
    tw = took.reshape(n_waves, 64)
    assert (tw.all(1) | ~tw.any(1)).all()                               # a wave takes ONE path
    assert tw[:128].all()                                               # realistic magnitudes: always the shared reciprocal
    assert not tw[448:].any()                                           # one lane out of range: the whole wave divides
    swept = tw[128:384, 0]
    assert swept.any() and not swept.all()                              # the sweep crosses the edge of the range
    assert np.isinf(b).any() or nan.any()                               # the degenerate rows really produce inf / NaN
