import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import test_gpu_refine as T
from instantavatar_amd.training import training_step
from instantavatar_amd.deformers import snarf_deformer as sd
G = T.G
for fused in (True, True, False, False):
    sd.FUSED_SMPL_BACKWARD = fused
    model, opt, loss_fn = T._setup()
    losses = training_step(model, T._batch(0), opt, loss_fn, is_refine=True, draws=T._draws(0, model))
    g = T._grads(model)
    print("fused_smpl", fused, "loss %.9e" % float(losses["loss"]), {n: "cos %.6f rel %.4f" % (T._cos(g[n], G[k]), T._rel(g[n], G[k])) for n, k in
          (("d_tfs", "d_tfs_0"), ("body_pose", "g_body_pose_0"), ("mlp_sigma", "g_mlp_sigma_0"), ("mlp_color", "g_mlp_color_0"))},
          "skipped", float(losses["skipped_non_finite"]), "amax-ish", float(np.abs(g["mlp_sigma"]).max()))
