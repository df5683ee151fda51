"""Dev tool (round 6): where a fit-stage step goes -- the SMPL body model under autograd (SMPLDeformer.prepare_deformer forward +
backward) against the whole training_step, wall clock with a device synchronisation around each part (the step is host-bound)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantavatar_amd.drivers import fit as fit_driver
from instantavatar_amd.training import NGPLoss, configure_optimizer, training_step
dev = torch.device("cuda:0")
frames, body_model, true = fit_driver.synthetic_frames(dev, res=256, n_frames=4, noise=0.02, patch=32)
model = fit_driver.build_fit_model(frames, body_model, dev)
opt = configure_optimizer(model, lr=1e-3, smpl_lr=1e-4)
loss_fn = NGPLoss(dict(w_rgb=1.0, w_alpha=0.1, w_reg=0.1, w_lpips=0.0, w_depth_reg=0.01))
model.train()
for it in range(25):
    training_step(model, frames.batch(it % 4), opt, loss_fn)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 50
for it in range(N):
    training_step(model, frames.batch(it % 4), opt, loss_fn)
torch.cuda.synchronize()
step_ms = (time.perf_counter() - t0) / N * 1e3
d = model.deformer
t0 = time.perf_counter()
for it in range(N):
    p = model.SMPL_param(torch.tensor([it % 4], device=dev))
    d.prepare_deformer(dict(p))
    g = torch.ones_like(d.T_inv)
    (d.T_inv * g).sum().backward()
    d.release_graph()
torch.cuda.synchronize()
prep_ms = (time.perf_counter() - t0) / N * 1e3
print("fit step %.2f ms (%.1f it/s); SMPL body model forward + backward alone %.2f ms" % (step_ms, 1e3 / step_ms, prep_ms))
