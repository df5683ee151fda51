"""Dev tool: k_search accounting of the occupancy probes vs the render loop of one frame."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from instantavatar_amd import _lib, synthetic
from instantavatar_amd.pipeline import build_synthetic_model, make_batch
dev = "cuda:0"
model, body, fp = build_synthetic_model(dev)
poses, transl = synthetic.procedural_pose_track(4)
batch = make_batch(dev, 512, poses[1], transl[1])
L = _lib.lib()
def get(i):
    ms = C.c_double(); n = C.c_int64(); u = (C.c_uint64 * 2)()
    L.ia_profile_get(i, C.byref(ms), C.byref(n), u)
    return ms.value, n.value, u[0], u[1]
for rep in range(2):
    model.render_image_fast(batch, (512, 512))
L.ia_profile_enable(1)
L.ia_profile_reset()
model.deformer.prepare_deformer(batch)
model.renderer.density_grid_test.initialize(model.deformer, model.net_coarse)
torch.cuda.synchronize()
print("probes : search ms %.3f launches %d solves %d fetches %d | field ms %.3f launches %d samples %d" % (get(0) + get(1)[:3]))
L.ia_profile_reset()
model.forward(batch, eval_mode=True)
torch.cuda.synchronize()
print("render : search ms %.3f launches %d solves %d fetches %d | field ms %.3f launches %d samples %d" % (get(0) + get(1)[:3]))
