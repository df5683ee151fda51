"""Dev tool: ia_precompute in isolation (events on the launch stream), with and without voxel_d / bbox."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from instantavatar_amd import _lib
from instantavatar_amd.pipeline import build_synthetic_model, make_batch
from instantavatar_amd import synthetic as syn

dev = torch.device("cuda", 0)
model, body, fp = build_synthetic_model(dev, resolution=128, n_levels=16)
poses, tr = syn.procedural_pose_track(8)
b = make_batch(dev, 64, poses[1], tr[1])
model.deformer.prepare_deformer(b)
fd = model.deformer.deformer
tfs = model.deformer.tfs.detach().float().contiguous()
fr = fd._frame
L = _lib.lib()


def run(want_d, want_bbox, n=200):
    args = (_lib.ptr(fd.lbs_voxel_final), _lib.ptr(tfs), _lib.ptr(fr["J"]), _lib.ptr(fr["d"]) if want_d else None,
            _lib.ptr(fr["bbox"]) if want_bbox else None, C.byref(fd.grid_desc()), _lib.stream())
    for _ in range(10):
        _lib.check(L.ia_precompute(*args))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        _lib.check(L.ia_precompute(*args))
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def run_ws(n=200):
    """the product's call: voxel_d + bounding box through per-workgroup extrema (ia_precompute_ws)"""
    ws = torch.empty(int(L.ia_precompute_workspace_bytes(C.byref(fd.grid_desc()))), dtype=torch.uint8, device=dev)
    args = (_lib.ptr(fd.lbs_voxel_final), _lib.ptr(tfs), _lib.ptr(fr["J"]), _lib.ptr(fr["d"]), _lib.ptr(fr["bbox"]), C.byref(fd.grid_desc()),
            _lib.ptr(ws), ws.numel(), _lib.stream())
    for _ in range(10):
        _lib.check(L.ia_precompute_ws(*args))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        _lib.check(L.ia_precompute_ws(*args))
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


us = run_ws()
print("tag=%s  product call (d + bbox via workspace, incl. the 1-workgroup reduce launch) %.1f us = %.2f TB/s of the 81.8 MB" % (sys.argv[1] if len(sys.argv) > 1 else "", us, 81.8e6 / us / 1e6))
print("tag=%s  d+bbox %.1f us   no d %.1f us   no d no bbox %.1f us" % (sys.argv[1] if len(sys.argv) > 1 else "", run(True, True), run(False, True), run(False, False)))
