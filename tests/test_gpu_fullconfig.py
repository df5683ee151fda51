"""Parity on the BENCH configuration (-m gpu): BASELINE.json configs[1] / north_star --
32x128x128 skinning voxels, 16-level hash grid, 512x512 rays, MAX_SAMPLES 256,
MAX_BATCH_SIZE 291600, five jittered 64^3 probe sets -- i.e. exactly the model `bench.py`
times, against the CPU oracle on the same injected jitter.

What only this size exercises: the device-side N_step schedule with 262 144 alive rays in the
first wave-front iteration, the `sample_cap` clamp of the compact sample queue, the XCD-sharded
encoding (calls >= 8192 samples) and the batched occupancy probes (5 x 64^3 x 13 solves in one
launch).  Reference: models/DNeRF.py:72-97, renderers/raymarcher_acc.py:83-138,
models/structures/density_grid.py:95-110.

Tolerances are those of tests/test_gpu_parity.py: rgb / alpha within 1e-3 absolute with at most
0.2 % of the rays moved by the reference's own discontinuities (occupancy cell flips,
alpha < 0.01 skips, T <= 1e-4 stops); occupancy cell flips < 2e-4; sample counters equal on
>= 99.8 % of the rays.
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
    assert info["occ_flips"] < 2e-4, (what, info)
    assert info["cov"] > 0.02, (what, info)
    assert info["frac_rgb"] < 2e-3 and info["frac_alpha"] < 2e-3, (what, info)
    assert info["median_rgb_on_body"] < 1e-4, (what, info)
    assert info["counter_mismatch"] < 2e-3, (what, info)
    assert abs(info["counter_mean"][0] - info["counter_mean"][1]) < 0.005 * max(1.0, info["counter_mean"][1]), (what, info)
    return info


def test_bench_configuration_parity_512_eager_and_graph(oracle, bench_world):
    """>= 2 procedural poses at 512x512 through `render_image_fast` (eager) AND the captured HIP graph
    (`GraphedRenderer`, what bench.py times), both against `oracle.render_image_fast`."""
    model, body, fp, init, poses, tr = bench_world
    res = 512
    ro, rd = syn.make_camera_rays(res)
    grid = model.renderer.density_grid_test
    jits = {i: np.random.RandomState(500 + i).rand(5, G ** 3, 3).astype(np.float32) for i in (1, 5)}
    refs = {}
    for i in (1, 5):
        ow = W.oracle_world(oracle, body, fp, init, poses[i], tr[i])
        refs[i] = oracle.render_image_fast(ow, ro, rd, jits[i])
    # eager
    for i in (1, 5):
        rgb, depth, alpha, counter = model.render_image_fast(make_batch(DEV, res, poses[i], tr[i]), (res, res),
                                                             jitter=torch.as_tensor(jits[i], device=DEV))
        _check(rgb, alpha, counter, grid.density_field, refs[i], "512^2 eager pose %d" % i)
    # graph replay: the occupancy jitter is a static device buffer read by the captured launches
    jit_dev = torch.as_tensor(jits[1], device=DEV).clone()
    orig = grid.initialize
    grid.initialize = lambda deformer, net, iters=5, jitter=None: orig(deformer, net, iters=iters, jitter=jit_dev)
    try:
        g = GraphedRenderer(model, make_batch(DEV, res, poses[1], tr[1]), (res, res), sync_check=True)
        for i in (1, 5):
            jit_dev.copy_(torch.as_tensor(jits[i], device=DEV))
            out = [t.clone() for t in g(make_batch(DEV, res, poses[i], tr[i]))]
            _check(out[0], out[2], out[3], grid.density_field, refs[i], "512^2 graph pose %d" % i)
        assert g.finish() == 0
    finally:
        grid.initialize = orig
    # two frames in flight (what bench.py times by default): every replica's frames against the oracle as well
    import instantavatar_amd.pipeline as P
    from instantavatar_amd.pipeline import PipelinedRenderer
    jit_a, jit_b = (torch.as_tensor(jits[i], device=DEV).clone() for i in (1, 5))
    patched, real_clone = [], P.clone_for_stream

    def patch(m, jit_dev):
        gr = m.renderer.density_grid_test
        o = gr.initialize
        gr.initialize = lambda deformer, net, iters=5, jitter=None, _o=o: _o(deformer, net, iters=iters, jitter=jit_dev)
        patched.append((gr, o))

    def patched_clone(m):
        c = real_clone(m)
        patch(c, jit_b)          # replica 1 always renders pose 5 below, replica 0 pose 1
        return c
    patch(model, jit_a)
    P.clone_for_stream = patched_clone
    try:
        pr = PipelinedRenderer(model, make_batch(DEV, res, poses[1], tr[1]), (res, res), n_in_flight=2)
        got = []
        keep = [make_batch(DEV, res, poses[i], tr[i]) for i in (1, 5, 1, 5)]
        for b in keep:
            pr(b, consume=lambda out, k: got.append(([t.clone() for t in out], pr.replicas[k].renderer.density_grid_test.density_field.clone())))
        pr.synchronize()
        assert pr.finish() == 0
        for n, i in enumerate((1, 5, 1, 5)):
            out, occ = got[n]
            _check(out[0], out[2], out[3], occ, refs[i], "512^2 two in flight, call %d pose %d" % (n, i))
    finally:
        P.clone_for_stream = real_clone
        for gr, o in patched:
            gr.initialize = o


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
