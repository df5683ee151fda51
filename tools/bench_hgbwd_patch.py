"""Dev tool: k_hashgrid_bwd per level on the candidates of a PATCH batch (4 patches of 32 x 32 rays on the body, as the
reference's default sampler draws them): neighbouring rays hit the same coarse cells from different waves."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from instantavatar_amd import _lib, synthetic as syn
from instantavatar_amd.models.structures.utils import Rays
from instantavatar_amd.pipeline import build_synthetic_model, make_batch

dev = "cuda:0"
res = 512
model, body, fp = build_synthetic_model(dev, resolution=128)
poses, tr = syn.procedural_pose_track(8)
b = make_batch(dev, res, poses[1], tr[1])
rgb, depth, alpha, counter = model.render_image_fast(b, (res, res))
rays = Rays(o=b["rays_o"], d=b["rays_d"], near=b["near"], far=b["far"])
model.deformer.transform_rays_w2s(rays)
m = (alpha.reshape(res, res) > 0.5)
ys, xs = m.nonzero(as_tuple=True)
g = torch.Generator(device=dev).manual_seed(3)
pick = torch.randint(0, ys.numel(), (4,), device=dev, generator=g)
sel = []
for k in pick.tolist():
    y0 = int(min(max(int(ys[k]) - 16, 0), res - 32)); x0 = int(min(max(int(xs[k]) - 16, 0), res - 32))
    yy, xx = torch.meshgrid(torch.arange(y0, y0 + 32, device=dev), torch.arange(x0, x0 + 32, device=dev), indexing="ij")
    sel.append((yy * res + xx).reshape(-1))
sel = torch.cat(sel)
o, d = rays.o.reshape(-1, 3)[sel], rays.d.reshape(-1, 3)[sel]
S = 96
dep = depth.reshape(-1)[sel]
dep = torch.where(dep > 0, dep, dep[dep > 0].mean())
ks = (torch.arange(S, device=dev, dtype=torch.float32) - S // 2) * (2.0 / 256)
t = dep[:, None] + ks[None]
pts = (o[:, None] + d[:, None] * t[..., None]).reshape(-1, 3).contiguous()
sc = model.deformer.search_compact(pts)
n = int(sc["n_cand"].item())
x = sc["cand_xc"][:n].contiguous()
net = model.net_coarse
V = x.shape[0]
L = _lib.lib()
dfeat = torch.randn((V, 32), device=dev) * 1e-3
dtable = torch.zeros(2 * net.n_entries, device=dev)


def t_(l0, l1, xx, reps=10):
    for _ in range(2):
        _lib.check(L.ia_hashgrid_bwd_levels(_lib.ptr(xx), V, None, C.byref(net.field_desc()), _lib.ptr(dfeat), dtable.data_ptr(), l0, l1, _lib.stream()))
    a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        _lib.check(L.ia_hashgrid_bwd_levels(_lib.ptr(xx), V, None, C.byref(net.field_desc()), _lib.ptr(dfeat), dtable.data_ptr(), l0, l1, _lib.stream()))
    b_.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b_) / reps * 1e3


print("tag=%s V = %d" % (sys.argv[1] if len(sys.argv) > 1 else "", V))
print("all 16 levels: %.1f us" % t_(0, 16, x))
print("levels: " + " ".join("%d:%.0f" % (l, t_(l, l + 1, x)) for l in range(16)))


# ---- round 3: would ordering the candidates by cell make the in-wave run reduction catch more? -------------------------
# Measured on MI355X (294 k candidates of a patch batch): all 16 levels 982 -> 815 us with the candidates in 30-bit Morton
# order (levels 0-7: 209/139/91/72/72/79/80/87 -> 108/90/75/53/58/61/58/64 us, levels 8-15 unchanged at ~71 us: they are at
# the atomic-request ceiling whatever the order).  A radix sort + two gathers per step would cost about half of the 167 us
# it saves: not built.
import time


def morton(xx, bb, bits=10):
    q = ((xx - bb[0]) / (bb[1] - bb[0]) * (2 ** bits - 1)).clamp_(0, 2 ** bits - 1).to(torch.int64)

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


if "--morton" in sys.argv:
    bb = model.deformer.bbox
    order = torch.argsort(morton(x, bb))
    xs = x[order].contiguous()
    dfeat = dfeat[order].contiguous()
    print("morton-ordered: all 16 levels: %.1f us" % t_(0, 16, xs))
    print("levels: " + " ".join("%d:%.0f" % (l, t_(l, l + 1, xs)) for l in range(16)))
