"""Dev tool: where does k_hashgrid_bwd spend its time?  Times the scatter per level group on the canonical samples of a
real frame (ray-major order, as the training step produces them) and on the same samples shuffled."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from instantavatar_amd import _lib, synthetic as syn
from instantavatar_amd.pipeline import build_synthetic_model, make_batch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # frame_coherent_samples

dev = "cuda:0"
model, body, fp = build_synthetic_model(dev, resolution=128)
poses, tr = syn.procedural_pose_track(8)
x = bench.frame_coherent_samples(model, make_batch(dev, 512, poses[1], tr[1]), 512)[:180000].contiguous()
net = model.net_coarse
V = x.shape[0]
L = _lib.lib()
dfeat = torch.randn((V, 32), device=dev) * 1e-3
dtable = torch.zeros(2 * net.n_entries, device=dev)


def t(l0, l1, xx, reps=20):
    for _ in range(3):
        _lib.check(L.ia_hashgrid_bwd_levels(_lib.ptr(xx), V, None, C.byref(net.field_desc()), _lib.ptr(dfeat), dtable.data_ptr(), l0, l1, _lib.stream()))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        _lib.check(L.ia_hashgrid_bwd_levels(_lib.ptr(xx), V, None, C.byref(net.field_desc()), _lib.ptr(dfeat), dtable.data_ptr(), l0, l1, _lib.stream()))
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


xs = x[torch.randperm(V, device=dev)].contiguous()
print("V =", V)
for name, xx in (("ray-major", x), ("shuffled", xs)):
    print(name, "all 16 levels: %.1f us" % t(0, 16, xx))
    for l in range(16):
        print("   level %2d: %6.1f us" % (l, t(l, l + 1, xx)))
