"""Sizes the ONE documented deviation from tiny-cuda-nn v1.6: the HIP kernels (and the oracle's default
mode) accumulate the MLP dot products in fp32, tcnn's FullyFusedMLP declares __half wmma accumulators.  Renders frames of
the bench workload (512^2, aist_demo track, the bench model) through the ORACLE in both modes and reports how many rays move
by more than the parity budget (1e-3 in rgb / alpha).  Runs on the GPU box (the model's skinning voxels come from the product;
the two renders are CPU, ~5 s each on its host cores):   python tools/size_mlp_accumulation.py [frame ...]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instantavatar_amd import synthetic as syn  # noqa: E402
from instantavatar_amd.pipeline import build_synthetic_model  # noqa: E402
from oracle import oracle as orc  # noqa: E402

dev = "cuda:0"
res = 512
model, body, fp = build_synthetic_model(dev, resolution=128, n_levels=16)
fd = model.deformer.deformer
init = dict(tfs_inv_t=model.deformer.tfs_inv_t[0].cpu().numpy(), lbs_voxel=np.ascontiguousarray(fd.lbs_voxel_final[0].cpu().numpy()),
            offset_kernel=fd.offset_kernel.reshape(3).cpu().numpy().astype(np.float32), scale_kernel=fd.scale_kernel.reshape(3).cpu().numpy().astype(np.float32),
            bbox=model.deformer.bbox.cpu().numpy(), D=32, H=128, W=128)
poses, tr = syn.load_animation_track(os.path.join(ROOT, "tests", "golden", "aist_demo_200.npz"))
ro, rd = syn.make_camera_rays(res)
frames = [int(a) for a in sys.argv[1:]] or [0, 100, 199]
out = []
for f in frames:
    world = orc.make_world(body, init, fp, np.zeros(10, np.float32), poses[f, 3:], poses[f, :3], tr[f], syn.INIT_BONES)
    jit = np.random.RandomState(900 + f).rand(5, 64 ** 3, 3).astype(np.float32)
    orc.set_mlp_half_accumulate(False)
    a = orc.render_image_fast(world, ro, rd, jit)
    orc.set_mlp_half_accumulate(True)
    b = orc.render_image_fast(world, ro, rd, jit)
    orc.set_mlp_half_accumulate(False)
    hit = (a["alpha"] > 0.01) | (b["alpha"] > 0.01)
    e_rgb = np.abs(a["rgb"] - b["rgb"]).max(1)
    e_a = np.abs(a["alpha"] - b["alpha"])
    r = dict(frame=f, rays=int(e_rgb.size), hit_rays=int(hit.sum()), occ_cells_differ=int((a["occ"] != b["occ"]).sum()),
             frac_rays_rgb_gt_1e3=float((e_rgb > 1e-3).mean()), frac_rays_alpha_gt_1e3=float((e_a > 1e-3).mean()),
             frac_hit_rays_rgb_gt_1e3=float((e_rgb[hit] > 1e-3).mean()), frac_hit_rays_alpha_gt_1e3=float((e_a[hit] > 1e-3).mean()),
             max_rgb=float(e_rgb.max()), max_alpha=float(e_a.max()), median_rgb_on_hit=float(np.median(e_rgb[hit])),
             p99_rgb_on_hit=float(np.percentile(e_rgb[hit], 99)), counter_differs=float((a["counter"] != b["counter"]).mean()))
    print(json.dumps(r), flush=True)
    out.append(r)
print(json.dumps({"summary": {k: float(np.mean([r[k] for r in out])) for k in out[0] if k != "frame"}}))
