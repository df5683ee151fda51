"""The `fit.py` stage (reference fit.py:15-75, bash/run-neuman-demo.sh:6) on the GPU: SMPL parameters
as trainable embeddings, optimised together with a fresh field through the SMPLDeformer plugin on patch batches from
the device-side data path, exported as poses/train.npz."""
import os

import numpy as np
import pytest
import torch

from instantavatar_amd.drivers import fit as fit_driver
from instantavatar_amd.training import NGPLoss, configure_optimizer, training_step

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_fit_stage_optimises_smpl_parameters_and_exports(tmp_path):
    torch.manual_seed(0)
    frames, body_model, true = fit_driver.synthetic_frames(torch.device(DEV), res=96, n_frames=3, noise=0.03, patch=16)
    model = fit_driver.build_fit_model(frames, body_model, torch.device(DEV))
    init = {k: v.copy() for k, v in model.SMPL_param.export().items()}
    # parameter groups of DNeRF.py:32-50: encoder / rest / SMPL tables with their own learning rate
    opt = configure_optimizer(model, lr=1e-3, smpl_lr=1e-4)
    assert len(opt.param_groups) == 3 and opt.param_groups[2]["lr"] == 1e-4 and len(opt.param_groups[2]["params"]) == 4
    loss_fn = NGPLoss(dict(w_rgb=1.0, w_alpha=0.1, w_reg=0.1, w_lpips=0.0, w_depth_reg=0.01))
    with pytest.raises(NotImplementedError):
        NGPLoss(dict(w_lpips=0.01))
    model.train()
    hist = []
    for it in range(45):
        losses = training_step(model, frames.batch(it % 3), opt, loss_fn)
        hist.append(float(losses["mse_loss"].detach()))
        if it == 0:
            assert "loss_depth_reg" in losses          # patch batches [1, n_patch, P, P, 3] -> loss.py:33-39 is active
            for k in ("body_pose", "global_orient", "transl"):
                g = getattr(model.SMPL_param, k).weight.grad
                assert g is not None and torch.isfinite(g).all() and g[0].abs().sum() > 0, k   # frame 0 was used: its row has a gradient
                assert g[1:].abs().sum() == 0                                                  # the other frames' rows have none
    assert np.isfinite(hist).all() and np.mean(hist[-9:]) < np.mean(hist[:9]), (hist[:9], hist[-9:])
    out = model.SMPL_param.export()
    for k in ("body_pose", "global_orient", "transl"):
        assert out[k].shape == init[k].shape and np.abs(out[k] - init[k]).max() > 0, k
    path = fit_driver.export_params(model, str(tmp_path))
    z = np.load(path)
    assert os.path.basename(os.path.dirname(path)) == "poses" and set(z.files) == {"betas", "global_orient", "transl", "body_pose"}
    assert z["body_pose"].shape == (3, 69) and z["betas"].shape == (1, 10)


def test_ngp_loss_with_lpips_term_trains_on_the_device():
    """The refine configuration's loss (confs/SNARF_NGP_refine.yaml: NGPLoss with w_lpips) on the device: the LPIPS term is
    present for patch batches, differentiable through the renderer, and its module equals its CPU evaluation.  (Random trunk
    weights: the pretrained ones are not available offline; the module is pinned to the reference on CPU,
    test_lpips_module_matches_reference_golden.)"""
    from instantavatar_amd.utils.lpips import LPIPS
    torch.manual_seed(0)
    frames, body_model, true = fit_driver.synthetic_frames(torch.device(DEV), res=96, n_frames=2, noise=0.0, patch=32)
    model = fit_driver.build_fit_model(frames, body_model, torch.device(DEV))
    lp = LPIPS()
    x, y = torch.rand(4, 3, 32, 32), torch.rand(4, 3, 32, 32)
    cpu = lp(x, y)
    lp = lp.to(DEV)
    assert torch.allclose(lp(x.to(DEV), y.to(DEV)).cpu(), cpu, rtol=1e-3, atol=1e-6)
    opt = configure_optimizer(model, lr=1e-3, smpl_lr=1e-4)
    loss_fn = NGPLoss(dict(w_rgb=1.0, w_alpha=0.1, w_reg=0.1, w_lpips=0.01, w_depth_reg=0.01), lpips=lp)
    model.train()
    for it in range(3):
        losses = training_step(model, frames.batch(it % 2), opt, loss_fn)
        # (randomly initialised lin layers have weights of both signs: only the pretrained ones make it a distance >= 0)
        assert "loss_lpips" in losses and torch.isfinite(losses["loss_lpips"]) and float(losses["loss_lpips"].detach()) != 0
        assert torch.isfinite(losses["loss"]) and float(losses["skipped_non_finite"]) == 0.0
    assert model.net_coarse.encoder.params.grad.abs().sum() > 0
