"""Dev tool (round 6): a 512^2 frame through the SMPLDeformer plugin (render_image_fast: occupancy build with 5 x 64^3 probes +
wave-front loop through the fused `ia_smpl_deform_query`) with the per-frame vertex grid against brute-force nearest-vertex search."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import world as W
from instantavatar_amd.pipeline import make_batch
dev = "cuda:0"
model, body, fp = W.build_smpl_deformer_world(dev)
poses, tr = W.poses()
model.eval()
out = {}
for use in (True, False):
    model.deformer.use_nn_grid = use
    for i in range(3):
        rgb, depth, alpha, counter = model.render_image_fast(make_batch(dev, 512, poses[i % 8], tr[i % 8]), (512, 512))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for i in range(n):
        rgb, depth, alpha, counter = model.render_image_fast(make_batch(dev, 512, poses[i % 8], tr[i % 8]), (512, 512))
    torch.cuda.synchronize()
    out[use] = (time.perf_counter() - t0) / n * 1e3
    print("SMPLDeformer frame 512^2, %s: %.2f ms per frame (%.1f frames/s), alpha coverage %.3f" % (
        "vertex grid" if use else "brute force", out[use], 1e3 / out[use], float((alpha > 0.5).float().mean())))
