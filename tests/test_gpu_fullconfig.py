"""Parity on the BENCH configuration (-m gpu): BASELINE.json configs[1] / north_star --
32x128x128 skinning voxels, 16-level hash grid, 512x512 rays, MAX_SAMPLES 256,
MAX_BATCH_SIZE 291600, five jittered 64^3 probe sets -- i.e. exactly the model `bench.py`
times, against the CPU oracle on the same injected jitter.

What only this size exercises: the device-side N_step schedule with 262 144 alive rays in the
first wave-front iteration, the `sample_cap` clamp of the compact sample queue, the XCD-sharded
encoding (calls >= 8192 samples) and the batched occupancy probes (5 x 64^3 x 13 solves in one
launch).  Reference: models/DNeRF.py:72-97, renderers/raymarcher_acc.py:83-138,
models/structures/density_grid.py:95-110.

Tolerances (tests/world.py: rays_within / cells_within, sized to what is measured): rgb / alpha within 1e-3
absolute on all but 1e-4 of the rays (26 of 262 144; measured 0-3 -- the reference's own discontinuities:
occupancy cell flips, alpha < 0.01 skips, T <= 1e-4 stops); occupancy cells differing <= 2e-5 (5; measured
<= 1); sample counters equal on >= 99.95 % of the rays.
"""
import numpy as np
import pytest
import torch

from instantavatar_amd import synthetic as syn
from instantavatar_amd.pipeline import GraphedRenderer, make_batch

import world as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = 64


@pytest.fixture(scope="module")
def bench_world(oracle):
    model, body, fp, init = W.build(DEV, 128, 16)   # == bench.py: build_synthetic_model(resolution=128, n_levels=16)
    assert model.renderer.MAX_SAMPLES == 256 and model.renderer.MAX_BATCH_SIZE == 291600
    assert tuple(model.deformer.deformer.lbs_voxel_final.shape[-3:]) == (32, 128, 128)
    poses, tr = W.poses()
    return model, body, fp, init, poses, tr


def _check(rgb, alpha, counter, occ_g, ref, what):
    rgb, alpha = rgb.reshape(-1, 3).cpu().numpy(), alpha.reshape(-1).cpu().numpy()
    counter = counter.reshape(-1).cpu().numpy()
    err_rgb = np.abs(rgb - ref["rgb"]).max(1)
    err_a = np.abs(alpha - ref["alpha"])
    info = dict(cov=float((ref["alpha"] > 0.5).mean()), occ_flips=float((occ_g.cpu().numpy() != ref["occ"].astype(bool)).mean()),
                frac_rgb=float((err_rgb > 1e-3).mean()), frac_alpha=float((err_a > 1e-3).mean()), max_rgb=float(err_rgb.max()),
                median_rgb_on_body=float(np.median(err_rgb[ref["alpha"] > 0.5])), counter_mismatch=float((counter != ref["counter"]).mean()),
                counter_mean=(float(counter.mean()), float(ref["counter"].mean())))
    print(what, info)
    W.cells_within(occ_g.cpu().numpy(), ref["occ"].astype(bool), what + " occupancy")          # <= 5 of 262 144 cells (measured <= 1)
    assert info["cov"] > 0.02, (what, info)
    W.rays_within(err_rgb, what + " rgb")                                                        # <= 1e-4 of the rays (measured 0-3 of 262 144)
    W.rays_within(err_a, what + " alpha")
    assert info["median_rgb_on_body"] < 1e-4, (what, info)
    assert info["counter_mismatch"] < 5e-4, (what, info)
    assert abs(info["counter_mean"][0] - info["counter_mean"][1]) < 0.005 * max(1.0, info["counter_mean"][1]), (what, info)
    return info


def _eager_graph_pipelined(oracle, bench_world, frames, seed0, what):
    """`frames`: list of (pose72, transl, betas or None).  Renders every frame at 512x512 through `render_image_fast`
    (eager), the captured HIP graph (`GraphedRenderer`) and two frames in flight (`PipelinedRenderer`, what bench.py
    times by default), each against `oracle.render_image_fast` on the same injected occupancy jitter."""
    model, body, fp, init = bench_world[:4]
    res = 512
    ro, rd = syn.make_camera_rays(res)
    grid = model.renderer.density_grid_test
    n = len(frames)
    jits = [np.random.RandomState(seed0 + i).rand(5, G ** 3, 3).astype(np.float32) for i in range(n)]
    refs, infos = [], []
    for i, (pose, transl, betas) in enumerate(frames):
        ow = W.oracle_world(oracle, body, fp, init, pose, transl, betas)
        refs.append(oracle.render_image_fast(ow, ro, rd, jits[i]))
    batch = lambda i: make_batch(DEV, res, frames[i][0], frames[i][1], betas=frames[i][2])
    # eager
    for i in range(n):
        rgb, depth, alpha, counter = model.render_image_fast(batch(i), (res, res), jitter=torch.as_tensor(jits[i], device=DEV))
        infos.append(_check(rgb, alpha, counter, grid.density_field, refs[i], "%s 512^2 eager frame %d" % (what, i)))
    # graph replay: the occupancy jitter is a static device buffer read by the captured launches
    jit_dev = torch.as_tensor(jits[0], device=DEV).clone()
    orig = grid.initialize
    grid.initialize = lambda deformer, net, iters=5, jitter=None: orig(deformer, net, iters=iters, jitter=jit_dev)
    try:
        g = GraphedRenderer(model, batch(0), (res, res), sync_check=True)
        for i in range(n):
            jit_dev.copy_(torch.as_tensor(jits[i], device=DEV))
            out = [t.clone() for t in g(batch(i))]
            _check(out[0], out[2], out[3], grid.density_field, refs[i], "%s 512^2 graph frame %d" % (what, i))
        assert g.finish() == 0
    finally:
        grid.initialize = orig
    # frames in flight, as the product runs them (PipelinedRenderer's defaults: three replicas, the first one's stream at high
    # priority) and with two replicas: replica k renders the calls k, k + n_rep, ...; each replica reads its own static jitter
    # buffer, refreshed before the call that uses it
    import instantavatar_amd.pipeline as P
    from instantavatar_amd.pipeline import PipelinedRenderer
    patched, real_clone = [], P.clone_for_stream

    def patch(m, jd):
        gr = m.renderer.density_grid_test
        o = gr.initialize
        gr.initialize = lambda deformer, net, iters=5, jitter=None, _o=o: _o(deformer, net, iters=iters, jitter=jd)
        patched.append((gr, o))

    try:
        for n_rep in (3, 2):
            jit_rep = [torch.as_tensor(jits[min(k, n - 1)], device=DEV).clone() for k in range(n_rep)]
            made = [0]

            def patched_clone(m):
                made[0] += 1
                c = real_clone(m)
                patch(c, jit_rep[made[0]])
                return c
            patch(model, jit_rep[0])
            P.clone_for_stream = patched_clone
            # (round robin: this test refreshes the jitter buffer of the replica that call i WILL run on; the product's default with
            #  three replicas hands a frame to the least loaded one -- covered by test_pipelined_renderer_frames_in_flight_equal_eager)
            pr = (PipelinedRenderer(model, batch(0), (res, res), n_in_flight=n_rep) if n_rep != 3
                  else PipelinedRenderer(model, batch(0), (res, res), schedule="round_robin"))
            assert len(pr.replicas) == n_rep and pr.priorities == ([-1, 0, 0] if n_rep == 3 else [0, 0]), pr.priorities
            assert pr.schedule == "round_robin" and PipelinedRenderer.__init__.__defaults__[0] == 3
            order = list(range(n)) * n_rep               # every frame passes through several replicas
            got = []
            jits_dev = [torch.as_tensor(j, device=DEV) for j in jits]
            torch.cuda.synchronize()
            for call, i in enumerate(order):
                k = call % n_rep
                with torch.cuda.stream(pr.streams[k]):    # behind the replica's previous frame, ahead of its next replay; the
                    jit_rep[k].copy_(jits_dev[i])          # other replicas' streams are not made to wait: the frames stay in flight
                pr(batch(i), consume=lambda out, kk: got.append(([t.clone() for t in out], pr.replicas[kk].renderer.density_grid_test.density_field.clone())))
            pr.synchronize()
            assert pr.finish() == 0
            for call, i in enumerate(order):
                out, occ = got[call]
                _check(out[0], out[2], out[3], occ, refs[i], "%s 512^2 %d in flight, call %d frame %d" % (what, n_rep, call, i))
            P.clone_for_stream = real_clone
            for gr, o in patched:
                gr.initialize = o
            del patched[:]
    finally:
        P.clone_for_stream = real_clone
        for gr, o in patched:
            gr.initialize = o
    return infos


@pytest.fixture(scope="module")
def blend_bench_world(oracle):
    """the bench configuration on the blend-shape body (non-zero shapedirs / posedirs, dense J_regressor) initialised with
    synthetic.BLEND_BETAS: the configuration class a real SMPL pickle + a data set's betas are (VERDICT r05 missing 3)"""
    model, body, fp, init = W.build(DEV, 128, 16, blend=True)
    assert tuple(model.deformer.deformer.lbs_voxel_final.shape[-3:]) == (32, 128, 128)
    assert float(model.deformer.body_model.shapedirs.abs().max()) > 0.01 and float(model.deformer.body_model.posedirs.abs().max()) > 0.005
    poses, tr = W.poses()
    return model, body, fp, init, poses, tr


def test_blend_shape_body_parity_512_eager_graph_and_pipelined(oracle, blend_bench_world):
    """512 x 512 frames with NON-ZERO betas on the blend-shape body through all three launch modes against the oracle: the rest
    joints (-> `ia_smpl_tfs`), the rest-pose vertices (-> the voxelised weights) and the field's canonical body all depend on
    the shape coefficients here.  One procedural pose and one frame of a shipped pose track."""
    poses, tr = blend_bench_world[4:]
    mp, mt, _ = _pose_track("male3")     # (its own betas belong to another subject: this body was initialised with BLEND_BETAS)
    frames = [(poses[2], tr[2], syn.BLEND_BETAS), (mp[57], mt[57], syn.BLEND_BETAS)]
    infos = _eager_graph_pipelined(oracle, blend_bench_world, frames, 700, "blend-shape body")
    assert all(i["cov"] > 0.02 for i in infos)


def test_bench_configuration_parity_512_eager_and_graph(oracle, bench_world):
    """>= 2 procedural poses at 512x512 through `render_image_fast` (eager), the captured HIP graph
    (`GraphedRenderer`) and two frames in flight, all against `oracle.render_image_fast`."""
    poses, tr = bench_world[4], bench_world[5]
    _eager_graph_pipelined(oracle, bench_world, [(poses[1], tr[1], None), (poses[5], tr[5], None)], 500, "procedural")


def test_bench_workload_parity_aist_demo_track(oracle, bench_world):
    """The workload bench.py TIMES (BASELINE config 3, animate.py:48-50): frames 0, 100 and 199 of
    `tests/golden/aist_demo_200.npz` = data/animation/aist_demo.npz[:200] with `trans - trans[0] + (0, 0.15, 5)`,
    through the three launch modes of the bench (eager, graph, two frames in flight) against the oracle.  The track has
    global-orient flips and a moving root that the procedural track does not."""
    import os
    poses, tr = syn.load_animation_track(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "aist_demo_200.npz"))
    assert poses.shape == (200, 72)
    infos = _eager_graph_pipelined(oracle, bench_world, [(poses[i], tr[i], None) for i in (0, 100, 199)], 900, "aist_demo")
    print("aist_demo frames 0/100/199:", [(round(x["cov"], 4), x["frac_rgb"], x["max_rgb"], x["occ_flips"]) for x in infos])


def _pose_track(name):
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pose_tracks.npz"))
    return (np.concatenate([z[name + "_global_orient"], z[name + "_body_pose"]], 1).astype(np.float32), z[name + "_transl"].astype(np.float32),
            z[name + "_betas"][0].astype(np.float32))


@pytest.mark.parametrize("name,frame", [("male3", 0), ("male3", 57), ("seattle", 20)])
def test_shipped_pose_tracks_frame_and_training_render(oracle, bench_world, name, frame):
    """BASELINE configs 2 and 4 are quoted on the pose tracks the reference SHIPS
    (data/PeopleSnapshot/male-3-casual/poses/anim_nerf_train.npz, data/custom/seattle/poses/train.npz; served by
    peoplesnapshot.py:127-131 as betas / global_orient / body_pose / transl, near / far = |transl| -+ 1, :143-150):
    one 512x512 frame and one training render (4 096 rays, jitter + sigma noise injected) of the bench model under those
    SMPL parameters against the oracle.  Fixture: tests/golden/pose_tracks.npz (make_pose_tracks.py)."""
    from instantavatar_amd.models.structures.utils import Rays
    model, body, fp, init = bench_world[:4]
    poses, tr, betas = _pose_track(name)
    res = 512
    ro, rd = syn.make_camera_rays(res)
    jit = np.random.RandomState(1300 + frame).rand(5, G ** 3, 3).astype(np.float32)
    ow = W.oracle_world(oracle, body, fp, init, poses[frame], tr[frame], betas)
    ref = oracle.render_image_fast(ow, ro, rd, jit)
    batch = make_batch(DEV, res, poses[frame], tr[frame], betas=betas)
    rgb, depth, alpha, counter = model.render_image_fast(batch, (res, res), jitter=torch.as_tensor(jit, device=DEV))
    info = _check(rgb, alpha, counter, model.renderer.density_grid_test.density_field, ref, "%s[%d] 512^2" % (name, frame))
    # ---- training render of the same frame (row a15 / a8 under the shipped parameters)
    n = 4096
    grid = model.renderer.density_grid_train
    model.deformer.prepare_deformer(batch)
    coords = (grid.coords + 0.5 / G) * (grid.aabb[1] - grid.aabb[0]) + grid.aabb[0]
    with torch.no_grad():
        _, dens = model.deformer(coords.reshape(-1, 3), model.net_coarse, True)
    grid._postprocess(dens.reshape(G, G, G))
    hit = torch.nonzero(alpha.reshape(-1) > 0.5).reshape(-1)
    gsel = torch.Generator(device=DEV).manual_seed(3)
    # half of the rays on the body, half anywhere (the patch sampler's rays nearly all cross the body)
    sel = torch.cat([hit[torch.randint(0, hit.numel(), (n // 2,), device=DEV, generator=gsel)],
                     torch.randint(0, res * res, (n // 2,), device=DEV, generator=gsel)])
    rays = Rays(o=batch["rays_o"][:, sel].clone(), d=batch["rays_d"][:, sel].clone(), near=batch["near"][:, sel].clone(),
                far=batch["far"][:, sel].clone())
    model.deformer.transform_rays_w2s(rays)
    bg = torch.rand((1, n, 3), device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
    c = lambda t: t.detach().reshape(-1, t.shape[-1]).cpu().numpy() if t.dim() > 2 else t.detach().reshape(-1).cpu().numpy()
    torch.manual_seed(11)
    jitter = torch.rand((n, 256), device=DEV)
    noise = torch.randn((n, 256), device=DEV)
    torch.manual_seed(11)
    out = model.renderer.render_train_fused(rays, model.deformer, model.net_coarse, 1, bg)
    tref = oracle.render_train(c(rays.o), c(rays.d), c(rays.near), c(rays.far), grid.density_field.cpu().numpy(),
                               grid.aabb.cpu().numpy(), lambda p: oracle.deform_query(p, ow, eval_mode=False),
                               jitter.cpu().numpy(), bg=c(bg), noise=noise.cpu().numpy())
    trgb = out["rgb_coarse"].detach().reshape(-1, 3).cpu().numpy()
    talpha = out["alpha_coarse"].detach().reshape(-1).cpu().numpy()
    tw = out["weight_coarse"].detach().reshape(n, 256).cpu().numpy()
    e_rgb, e_a, e_w = np.abs(trgb - tref["rgb"]).max(1), np.abs(talpha - tref["alpha"]), np.abs(tw - tref["weights"]).max(1)
    tinfo = dict(hit=float((tref["alpha"] > 0.5).mean()), n_field=int(tref["n_field"]), frac_rgb=float((e_rgb > 1e-3).mean()),
                 frac_alpha=float((e_a > 1e-3).mean()), frac_w=float((e_w > 1e-3).mean()), max_rgb=float(e_rgb.max()), median_rgb=float(np.median(e_rgb)))
    print("%s[%d] training render" % (name, frame), tinfo)
    assert tinfo["hit"] > 0.2 and tinfo["n_field"] > 10000, tinfo
    for e, nm in ((e_rgb, "rgb"), (e_a, "alpha"), (e_w, "weights")):
        W.rays_within(e, "%s[%d] training render %s" % (name, frame, nm), frac=5e-4)               # <= 2 of 4 096 rays (measured 0)
    assert tinfo["median_rgb"] < 1e-4, tinfo


def test_bench_configuration_parity_1024(oracle, bench_world):
    """BASELINE.json configs[4] renders 1024x1024: one pose at that size (R = 1 048 576 > MAX_BATCH_SIZE,
    so the first wave-front iteration runs with N_step = 1 on a queue sized by R, not by MAX_BATCH)."""
    model, body, fp, init, poses, tr = bench_world
    res = 1024
    ro, rd = syn.make_camera_rays(res)
    jit = np.random.RandomState(77).rand(5, G ** 3, 3).astype(np.float32)
    ow = W.oracle_world(oracle, body, fp, init, poses[3], tr[3])
    ref = oracle.render_image_fast(ow, ro, rd, jit)
    rgb, depth, alpha, counter = model.render_image_fast(make_batch(DEV, res, poses[3], tr[3]), (res, res),
                                                         jitter=torch.as_tensor(jit, device=DEV))
    _check(rgb, alpha, counter, model.renderer.density_grid_test.density_field, ref, "1024^2")


def test_intra_frame_row_sharding_equals_the_whole_frame(bench_world):
    """SURVEY 8e (optional intra-frame sharding for latency; BASELINE config 5 renders 1024^2): the frame rendered as the row
    blocks 8 ranks would take (`parallel.shard_rows`), one after the other on this GPU, against the whole frame.  A ray's march
    and compositing do not depend on which other rays are alive: rgb / depth / alpha must be BIT-EQUAL; the per-ray sample
    counter follows the N_step schedule (fewer alive rays -> more samples marched per iteration, raymarcher_acc.py:104): a ray
    that terminates mid-iteration has marched the rest of that iteration's samples too, so the two counts of a ray differ by
    the schedules' overshoot in either direction.  What does not depend on the schedule: WHICH rays march any sample at all."""
    from instantavatar_amd.parallel import render_frame_tiled, shard_rows
    model, body, fp, init, poses, tr = bench_world
    res, world = 512, 8
    jit = torch.rand((5, G ** 3, 3), device=DEV, generator=torch.Generator(device=DEV).manual_seed(77))
    batch = make_batch(DEV, res, poses[2], tr[2])
    full = [t.clone() for t in model.render_image_fast(batch, (res, res), jitter=jit)]
    parts = []
    for r in range(world):
        out = render_frame_tiled(model, make_batch(DEV, res, poses[2], tr[2]), (res, res), world, r, jitter=jit, gather=False)
        r0, r1 = shard_rows(res, r, world)
        assert out[0].shape == (1, r1 - r0, res, 3)
        parts.append([t.clone() for t in out])
    for i, name in enumerate(("rgb", "depth", "alpha")):
        got = torch.cat([p[i] for p in parts], dim=1)
        assert torch.equal(got, full[i]), (name, float((got - full[i]).abs().max()))
    cnt = torch.cat([p[3] for p in parts], dim=1)
    assert (full[2] > 0.5).float().mean() > 0.02
    hit = full[2] > 0.01
    assert torch.equal(cnt > 0, full[3] > 0), "the set of rays that meet an occupied cell depends on the row sharding"
    assert bool((cnt[hit] >= 1).all()) and float((cnt.float() - full[3].float()).abs().mean()) < 64
