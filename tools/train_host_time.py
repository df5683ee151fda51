"""Dev tool: how much of a training step is host time?  Runs the bench's training loop and prints the time the
host needs to ENQUEUE a step (no synchronisation) next to the synchronised step time, then a cProfile of 200 steps."""
import cProfile
import pstats
import sys
import time

import torch

sys.path.insert(0, ".")
from instantavatar_amd import synthetic as syn  # noqa: E402
from instantavatar_amd.pipeline import build_synthetic_model, make_batch  # noqa: E402
from instantavatar_amd.training import NeRFLoss, configure_optimizer, training_step  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(42)
model, body, fp = build_synthetic_model(dev, resolution=128, n_levels=16)
poses, tr = syn.procedural_pose_track(200)
res, n_rays = 512, 4096
targets = []
with torch.no_grad():
    for f in range(4):
        b = make_batch(dev, res, poses[f], tr[f])
        rgb, _, alpha, _ = model.render_image_fast(b, (res, res))
        targets.append((b, rgb.reshape(1, -1, 3), alpha.reshape(1, -1)))
trainee, _, _ = build_synthetic_model(dev, resolution=128, n_levels=16)
trainee.net_coarse.reset_parameters()
trainee.train()
opt = configure_optimizer(trainee)
loss_fn = NeRFLoss(dict(w_rgb=1.0, w_alpha=0.1, w_reg=0.1))
g = torch.Generator(device=dev).manual_seed(1234)


def step(i):
    b, rgb, alpha = targets[i % 4]
    sel = torch.randint(0, res * res, (n_rays,), device=dev, generator=g)
    batch = dict(b)
    for k in ("rays_o", "rays_d", "near", "far"):
        batch[k] = b[k][:, sel]
    batch["rgb"], batch["alpha"] = rgb[:, sel], alpha[:, sel]
    batch["bg_color"] = torch.ones_like(batch["rgb"])
    return training_step(trainee, batch, opt, loss_fn)


for i in range(30):
    step(i)
torch.cuda.synchronize()
N = 200
t0 = time.perf_counter()
for i in range(N):
    step(30 + i)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("host enqueue %.3f ms/step, synchronised %.3f ms/step (%.0f it/s)" % (t_host / N * 1e3, t_all / N * 1e3, N / t_all))
# host-only cost: same loop while the GPU is kept far behind is not possible; profile instead
pr = cProfile.Profile()
pr.enable()
for i in range(N):
    step(300 + i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
