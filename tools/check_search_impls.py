"""Dev tool (GPU): the persistent-wave search (k_solve + k_roots) against the workgroup-queue kernel of rounds 2-3 on the
sample points of a real 512^2 frame and on the occupancy probe points -- results must be BIT-IDENTICAL (per point: the
number of surviving roots, and the roots in init order; dense layout: xc / valid / valid_raw / J_inv; compact J_inv) --
and their launch times side by side.   python tools/check_search_impls.py [S]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instantavatar_amd import _lib, synthetic as syn  # noqa: E402
from instantavatar_amd.models.structures.utils import Rays  # noqa: E402
from instantavatar_amd.pipeline import build_synthetic_model, make_batch  # noqa: E402

dev = torch.device("cuda", 0)
model, body, fp = build_synthetic_model(dev, resolution=128, n_levels=16)
poses, tr = syn.load_animation_track(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "aist_demo_200.npz"))
res = 512
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
b = make_batch(dev, res, poses[0], tr[0])
rgb, depth, alpha, counter = model.render_image_fast(b, (res, res))
rays = Rays(o=b["rays_o"], d=b["rays_d"], near=b["near"], far=b["far"])
model.deformer.transform_rays_w2s(rays)
sel = (alpha.reshape(-1) > 0.5).nonzero().reshape(-1)
o, d = rays.o.reshape(-1, 3)[sel], rays.d.reshape(-1, 3)[sel]
ks = (torch.arange(S, device=dev, dtype=torch.float32) - S // 2) * (2.0 / 256)
t = depth.reshape(-1)[sel][:, None] + ks[None]
pts_frame = (o[:, None] + d[:, None] * t[..., None]).reshape(-1, 3).contiguous()
grid = model.renderer.density_grid_test
G = 64
g = torch.Generator(device=dev).manual_seed(1)
cells = (grid.coords.reshape(-1, 3)[:, None] + torch.rand((G ** 3, 5, 3), device=dev, generator=g) / G)   # cell-major: 5 jittered points per cell
pts_probe = (cells.reshape(-1, 3) * (grid.aabb[1] - grid.aabb[0]) + grid.aabb[0]).contiguous()

dd = model.deformer
fd = dd.deformer
k = len(fd.init_bones)
L = _lib.lib()
tfs = dd.tfs.detach().float().contiguous()


def compact(pts, impl, with_jinv=False, n=20):
    P = pts.shape[0]
    _lib.check(L.ia_search_set_impl(impl), "ia_search_set_impl")
    cand = torch.zeros((P * k, 3), device=dev)
    cj = torch.zeros((P * k, 9), device=dev) if with_jinv else None
    pt_off = torch.zeros(P, dtype=torch.int32, device=dev)
    pt_cnt = torch.zeros(P, dtype=torch.uint8, device=dev)
    n_cand = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = torch.empty(int(L.ia_snarf_search_workspace_bytes(P, k, 2 if with_jinv else 1)), dtype=torch.uint8, device=dev)

    def once():
        head = (_lib.ptr(pts), P, None, _lib.ptr(fd.voxel_J_cl), _lib.ptr(tfs), fd._bones_c, k, C.byref(fd.grid_desc()), 1e-5, 1e-1, _lib.ptr(cand))
        tail = (P * k, _lib.ptr(pt_off), _lib.ptr(pt_cnt), _lib.ptr(n_cand), 1, _lib.ptr(ws), ws.numel(), _lib.stream())
        if with_jinv:
            _lib.check(L.ia_snarf_search_compact_jinv(*head, _lib.ptr(cj), *tail), "search_compact_jinv")
        else:
            _lib.check(L.ia_snarf_search_compact(*head, *tail), "search_compact")
    for _ in range(3):
        once()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        once()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    return dict(cand=cand, cj=cj, pt_off=pt_off, pt_cnt=pt_cnt, n=int(n_cand.item()), us=us)


def per_point(r, width):
    """[P, 13, width] with the roots of every point in init order (zero padded), from the compact lists"""
    P = r["pt_off"].shape[0]
    idx = r["pt_off"].long()[:, None] + torch.arange(13, device=dev)[None]
    m = torch.arange(13, device=dev)[None] < r["pt_cnt"].long()[:, None]
    src = r["cand"] if width == 3 else r["cj"]
    out = src[idx.clamp(max=src.shape[0] - 1)]
    return torch.where(m[..., None], out, torch.zeros_like(out))


def dense(pts, impl):
    P = pts.shape[0]
    _lib.check(L.ia_search_set_impl(impl), "ia_search_set_impl")
    xc = torch.full((P, k, 3), 7.0, device=dev); valid = torch.full((P, k), 9, device=dev, dtype=torch.uint8)
    raw = torch.full((P, k), 9, device=dev, dtype=torch.uint8); Ji = torch.full((P, k, 9), 7.0, device=dev)
    ws = torch.empty(int(L.ia_snarf_search_workspace_bytes(P, k, 0)), dtype=torch.uint8, device=dev)
    _lib.check(L.ia_snarf_search(_lib.ptr(pts), P, _lib.ptr(fd.voxel_J_cl), _lib.ptr(tfs), fd._bones_c, k, C.byref(fd.grid_desc()), 1e-5, 1e-1,
                                 _lib.ptr(xc), _lib.ptr(valid), _lib.ptr(raw), _lib.ptr(Ji), _lib.ptr(ws), ws.numel(), _lib.stream()), "ia_snarf_search")
    torch.cuda.synchronize()
    return xc, valid, raw, Ji


bad = 0
for name, pts in (("frame", pts_frame), ("probe", pts_probe)):
    a, bq = compact(pts, 0), compact(pts, 1)
    same_cnt = bool(torch.equal(a["pt_cnt"], bq["pt_cnt"]))
    same_x = bool(torch.equal(per_point(a, 3).view(torch.int32), per_point(bq, 3).view(torch.int32)))
    print("%s: P=%d  n_cand %d / %d  counts equal %s  roots bit-equal %s   workgroup queues %.1f us   persistent waves %.1f us" % (
        name, pts.shape[0], a["n"], bq["n"], same_cnt, same_x, a["us"], bq["us"]))
    bad += (not same_cnt) + (not same_x) + (a["n"] != bq["n"])
sub = pts_frame[: 40000].contiguous()
a, bq = compact(sub, 0, True, n=3), compact(sub, 1, True, n=3)
ok = torch.equal(per_point(a, 9).view(torch.int32), per_point(bq, 9).view(torch.int32)) and torch.equal(per_point(a, 3).view(torch.int32), per_point(bq, 3).view(torch.int32))
print("compact + J_inv (40 000 points): bit-equal", bool(ok), " %.1f / %.1f us" % (a["us"], bq["us"]))
bad += not ok
da, db = dense(sub, 0), dense(sub, 1)
for nm, u, v in zip(("xc", "valid", "valid_raw", "J_inv"), da, db):
    e = bool(torch.equal(u.view(torch.int32) if u.dtype == torch.float32 else u, v.view(torch.int32) if v.dtype == torch.float32 else v))
    print("dense %s bit-equal %s" % (nm, e))
    bad += not e
# a point count that is not a multiple of 64, and a tiny one
for P in (1, 63, 65, 1000):
    a, bq = compact(pts_frame[:P].contiguous(), 0, n=1), compact(pts_frame[:P].contiguous(), 1, n=1)
    e = torch.equal(a["pt_cnt"], bq["pt_cnt"]) and torch.equal(per_point(a, 3).view(torch.int32), per_point(bq, 3).view(torch.int32))
    print("P=%d equal %s" % (P, bool(e)))
    bad += not e
print("MISMATCHES:", bad)
sys.exit(1 if bad else 0)
