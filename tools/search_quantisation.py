"""Dev tool (round 6): what every k_search launch of the inference frame is made of -- per wave-front
iteration the alive rays, N_step, the sample points handed to the search (= P of the launch), live workgroups (64 points
each) against the 1 024 resident slots (256 CUs x 4 workgroups at 108 VGPRs) and the candidates found -- read from the
renderer's device-side RenderState records after eager frames of the bench workload (aist_demo, 512^2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from instantavatar_amd import synthetic as syn
from instantavatar_amd.pipeline import build_synthetic_model, make_batch
dev = torch.device("cuda:0")
model, body, fp = build_synthetic_model(dev)
poses, tr = syn.load_animation_track(os.path.join(bench.ROOT, "tests", "golden", "aist_demo_200.npz"))
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rows = []
for f in range(n_frames):
    model.render_image_fast(make_batch(dev, 512, poses[f * 10 % len(poses)], tr[f * 10 % len(poses)]), (512, 512))
    torch.cuda.synchronize()
    st = model.renderer._ws[:32 * 16].view(torch.int32).reshape(16, 8).cpu().numpy()
    n_it = int(model.renderer.iters_executed())
    rows.append(st[:max(n_it, 1) + 1].copy())
n_it = max(len(r) for r in rows)
print("frames %d; columns: mean over frames" % n_frames)
print("%-6s %10s %7s %10s %10s %9s %10s" % ("iter", "alive rays", "N_step", "points P", "live wg", "wg/1024", "candidates"))
for i in range(n_it):
    v = np.array([r[i] for r in rows if len(r) > i], np.float64)
    if not len(v) or v[:, 6].mean() == 0:
        continue
    P = v[:, 1]
    print("%-6d %10.0f %7.1f %10.0f %10.0f %9.2f %10.0f" % (i, v[:, 6].mean(), v[:, 5].mean(), P.mean(), np.ceil(P / 64).mean(), np.ceil(P / 64).mean() / 1024, v[:, 2].mean()))
