"""Dev tool: ia_snarf_search_compact in isolation on the sample points of a real 512^2 frame (64 march steps around the
surface of every hit ray, ray-major) and on the 64^3 x 5 probe points; events on the launch stream."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from instantavatar_amd import _lib, synthetic as syn
from instantavatar_amd.models.structures.utils import Rays
from instantavatar_amd.pipeline import build_synthetic_model, make_batch

dev = torch.device("cuda", 0)
model, body, fp = build_synthetic_model(dev, resolution=128, n_levels=16)
poses, tr = syn.procedural_pose_track(8)
res = 512
import os
b = make_batch(dev, res, poses[1], tr[1])
S = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cache = "/tmp/bench_search_pts_%d.pt" % S   # the point list comes from a CORRECT build (first variant of an A/B run)
if os.path.exists(cache):
    model.deformer.prepare_deformer(b)
    pts = torch.load(cache).to(dev)
else:
    rgb, depth, alpha, counter = model.render_image_fast(b, (res, res))
    rays = Rays(o=b["rays_o"], d=b["rays_d"], near=b["near"], far=b["far"])
    model.deformer.transform_rays_w2s(rays)
    sel = (alpha.reshape(-1) > 0.5).nonzero().reshape(-1)
    o, d = rays.o.reshape(-1, 3)[sel], rays.d.reshape(-1, 3)[sel]
    ks = (torch.arange(S, device=dev, dtype=torch.float32) - S // 2) * (2.0 / 256)
    t = depth.reshape(-1)[sel][:, None] + ks[None]
    pts = (o[:, None] + d[:, None] * t[..., None]).reshape(-1, 3).contiguous()
    torch.save(pts.cpu(), cache)
dd = model.deformer
fd = dd.deformer
k = len(fd.init_bones)
P = pts.shape[0]
L = _lib.lib()
cand = torch.empty((P * k, 3), device=dev)
pt_off = torch.empty(P, dtype=torch.int32, device=dev)
pt_cnt = torch.empty(P, dtype=torch.uint8, device=dev)
n_cand = torch.zeros(64 * 32, dtype=torch.int32, device=dev)
tfs = dd.tfs.detach().float().contiguous()


def run(n=30):
    def once():
        n_cand.zero_()
        _lib.check(L.ia_snarf_search_compact(_lib.ptr(pts), P, None, _lib.ptr(fd.voxel_J_cl), _lib.ptr(tfs), fd._bones_c, k,
                                             C.byref(fd.grid_desc()), 1e-5, 1e-1, _lib.ptr(cand), P * k, _lib.ptr(pt_off), _lib.ptr(pt_cnt),
                                             _lib.ptr(n_cand), 0, _lib.stream()))
    for _ in range(5):
        once()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        once()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


us = run()
print("tag=%s  P=%d  n_cand=%d  %.1f us  (%.2f G points/s)" % (sys.argv[1] if len(sys.argv) > 1 else "", P, int(n_cand.sum()), us, P / us * 1e-3))
