"""Dev tool (round 5): does the ORDER of a real frame's samples change the isolated hash-grid encode time?
pipeline order (ray-major) vs shuffled vs Morton-sorted vs step-major within groups of 64 adjacent rays."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from instantavatar_amd import synthetic as syn
from instantavatar_amd.models.structures.utils import Rays
from instantavatar_amd.pipeline import build_synthetic_model, make_batch
dev = torch.device("cuda:0")
model, body, fp = build_synthetic_model(dev)
poses, tr = syn.load_animation_track(os.path.join(bench.ROOT, "tests", "golden", "aist_demo_200.npz"))
res = 512
batch = make_batch(dev, res, poses[0], tr[0])
rgb, depth, alpha, counter = model.render_image_fast(batch, (res, res))
rays = Rays(o=batch["rays_o"], d=batch["rays_d"], near=batch["near"], far=batch["far"])
model.deformer.transform_rays_w2s(rays)
sel = (alpha.reshape(-1) > 0.5).nonzero().reshape(-1)
o, d = rays.o.reshape(-1, 3)[sel], rays.d.reshape(-1, 3)[sel]
S = 64
ks = (torch.arange(S, device=dev, dtype=torch.float32) - 8) * (2.0 / 256)
t = depth.reshape(-1)[sel][:, None] + ks[None]
pts = (o[:, None] + d[:, None] * t[..., None]).reshape(-1, 3).contiguous()
sc = model.deformer.search_compact(pts)
n = int(sc["n_cand"].item())
x = sc["cand_xc"][:n].contiguous()
cnt = sc["pt_cnt"].long()
pt_of_cand = torch.repeat_interleave(torch.arange(pts.shape[0], device=dev), cnt)
# candidates of one point are contiguous starting at pt_off; the global order of points follows atomic arrival per workgroup: recover by sorting on pt_off
order_pts = torch.argsort(sc["pt_off"].long() * 16 + 0)   # not needed for keys below, only for reference
off = sc["pt_off"].long()
cand_pt = torch.empty(n, dtype=torch.long, device=dev)
has = cnt > 0
idx_pts = has.nonzero().reshape(-1)
for k in range(int(cnt.max())):
    m = idx_pts[cnt[idx_pts] > k]
    cand_pt[off[m] + k] = m
ray, slot = cand_pt // S, cand_pt % S
net = model.net_coarse
bb = model.deformer.bbox
def T(xx, what):
    us = bench._time_encode(net, xx.contiguous(), 30)
    print("%-46s %8.1f us  %.3f Gsamples/s  frac %.3f" % (what, us, xx.shape[0] / us * 1e-3, xx.shape[0] * 512 / (us * 1e-6) / 8e12))
print("samples", n, "rays", len(sel))
T(x, "pipeline order (compaction order)")
T(x[torch.argsort(ray * S + slot)], "strict ray-major, step-minor")
T(x[torch.randperm(n, device=dev)], "shuffled")
om, _ = bench._morton_order(x, bb)
T(x[om], "Morton-sorted (canonical position, 30 bit)")
for Gr in (16, 64, 256):
    key = (ray // Gr) * (S * Gr) + slot * Gr + ray % Gr
    T(x[torch.argsort(key)], "step-major within groups of %d adjacent rays" % Gr)
key = slot * (len(sel) + 1) + ray
T(x[torch.argsort(key)], "step-major over the whole frame")
