"""`fit.py` without Lightning / Hydra (reference: fit.py:15-75, confs/SNARF_NGP_fitting.yaml): the stage of the
Neuman pipeline (bash/run-neuman-demo.sh:6) that optimises the per-frame SMPL parameters TOGETHER with a field through
the `SMPLDeformer` plugin (deformer=smpl), on patch batches (sampler=patch) with NGPLoss, and writes the optimised
parameters to `<out>/poses/train.npz` for the train stage.

    python -m instantavatar_amd.drivers.fit --synthetic --steps 200 --out /tmp/seq

`fit_sequence(...)` takes any `DeviceFrames` (instantavatar_amd/datasets/device_frames.py); `--synthetic` builds one
from the synthetic avatar with perturbed initial poses (no dataset ships with this package).  The LPIPS term of the
reference's loss needs pretrained VGG weights and is not available offline (NGPLoss raises when asked for it).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

from .. import synthetic
from ..datasets.device_frames import DeviceFrames
from ..deformers.smpl_deformer import SMPLDeformer
from ..deformers.smplx import SMPL
from ..models.networks.ngp import NeRFNGPNet
from ..models.structures.body_model_param import SMPLParamEmbedding
from ..pipeline import AvatarModel, build_synthetic_model, make_batch
from ..renderers.raymarcher_acc import Raymarcher
from ..training import GraphedTrainStep, NGPLoss, configure_optimizer, configure_scheduler, training_step
from ..utils.sampler import PatchSampler


def build_fit_model(frames, body_model, device, threshold=0.05, n_levels=16):
    """DNeRFModel.__init__ for the fitting configuration (DNeRF.py:18-30): NeRFNGPNet + SMPLDeformer + Raymarcher and
    the SMPL parameter embedding initialised from the dataset (datamodule.trainset.get_SMPL_params())."""
    deformer = SMPLDeformer(None, "neutral", threshold=threshold, k=1, body_model=body_model)
    net = NeRFNGPNet(dict(center=[0, -0.3, 0], scale=[2.5, 2.5, 2.5]), n_levels=n_levels).to(device)
    renderer = Raymarcher(256, 291600).to(device)
    renderer.initialize(len(frames))
    model = AvatarModel(deformer, net, renderer).to(device)
    model.SMPL_param = SMPLParamEmbedding(**{k: v.detach().cpu() for k, v in frames.smpl_params.items()}).to(device)
    return model


def fit_sequence(model, frames, steps, lr=1e-3, smpl_lr=1e-4, max_epochs=300, loss_opt=None, log_every=50, out=sys.stdout,
                 generator=None, check_val_every_n_epoch=10, rank=0, world_size=1, graphed=True):
    """The optimisation loop of fit.py (trainer.fit with SNARF_NGP_fitting.yaml: Adam lr 1e-3, SMPL tables lr 1e-4, one frame
    per step; the LambdaLR steps once per validation run = every `check_val_every_n_epoch` epochs, DNeRF.py:163-166 -- see
    training.configure_scheduler).  Returns the last losses.
    world_size > 1: every rank calls this; replicas start from rank 0's state, rank r takes positions r, r + W, ... of the
    epoch's (shared) shuffle -- a frame's SMPL rows receive a gradient from the rank that drew the frame only, the average
    over ranks scales it by 1 / W, which Adam's normalisation absorbs -- and an epoch is ceil(frames / W) steps.
    NOT a parity mode (the reference's fit.py is single-GPU): when the frame count is not a multiple of W the last step of an
    epoch wraps around, so the first frames of the shuffle are drawn twice in that epoch, and the moments of SMPL rows that got
    no gradient in a step still decay -- the optimisation trajectory differs from the 1-rank run's (same optimum, tested to
    train: tests/test_gpu_drivers.py)."""
    from ..parallel import broadcast_module_state
    broadcast_module_state(model, world_size)
    opt = configure_optimizer(model, lr=lr, smpl_lr=smpl_lr)
    sched = configure_scheduler(opt, max_epochs)
    loss_fn = NGPLoss(loss_opt or dict(w_rgb=1.0, w_alpha=0.1, w_reg=0.1, w_lpips=0.0, w_depth_reg=0.01))
    model.train()
    # graphed (one rank): the step is captured once (once per frame with `smpl_init`, where every frame has its own occupancy grid)
    # and replayed afterwards; occupancy-update steps and a capture that fails run eagerly (training.GraphedTrainStep)
    stepper = GraphedTrainStep(model, opt, loss_fn, world_size=world_size, enabled=graphed)
    n = len(frames)
    if generator is None and world_size > 1:
        generator = torch.Generator().manual_seed(42)      # the ranks must shuffle alike (pl.seed_everything(42) in the reference)
    per_epoch = -(-n // world_size)
    order = torch.randperm(n, generator=generator).tolist()
    t0 = time.perf_counter()
    losses = None
    for it in range(steps):
        if it % per_epoch == 0 and it > 0:
            if (it // per_epoch) % check_val_every_n_epoch == 0:
                sched.step()
            order = torch.randperm(n, generator=generator).tolist()   # DataLoader(shuffle=True)
        losses = stepper(frames.batch(order[((it % per_epoch) * world_size + rank) % n], out=stepper.inputs))
        if log_every and (it + 1) % log_every == 0:
            torch.cuda.synchronize()
            print("fit step %d  loss %.5f  mse %.5f  %.1f it/s" % (it + 1, float(losses["loss"].detach()), float(losses["mse_loss"].detach()),
                                                                   (it + 1) / (time.perf_counter() - t0)), file=out)
    return losses


def export_params(model, out_dir):
    """fit.py:48-64: optimised SMPL tables -> <out_dir>/poses/train.npz"""
    root = os.path.join(out_dir, "poses")
    os.makedirs(root, exist_ok=True)
    path = os.path.join(root, "train.npz")
    np.savez(path, **model.SMPL_param.export())
    return path


def synthetic_frames(device, res=128, n_frames=4, noise=0.02, seed=0, patch=32, blendshapes=False):
    """Frames rendered from the synthetic avatar; the SMPL parameters handed to the fit stage are perturbed.
    blendshapes: a subject with non-zero shape / pose directions and shape coefficients synthetic.BLEND_BETAS (the betas row
    handed to the fit stage is perturbed too: the stage optimises it, DNeRF.py:121-123)."""
    betas = synthetic.BLEND_BETAS if blendshapes else np.zeros(10, np.float32)
    teacher, body, _ = build_synthetic_model(device, resolution=64, blendshapes=blendshapes, betas=betas)
    poses, tr = synthetic.procedural_pose_track(max(n_frames, 8))
    imgs, masks = [], []
    with torch.no_grad():
        for f in range(n_frames):
            rgb, _, alpha, _ = teacher.render_image_fast(make_batch(device, res, poses[f], tr[f], betas=betas), (res, res))
            imgs.append((rgb[0].clamp(0, 1) * 255).round().to(torch.uint8))
            masks.append((alpha[0] > 0.5).float())
    rng = np.random.RandomState(seed)
    K = np.array([[2000.0 * res / 1080, 0, res / 2], [0, 2000.0 * res / 1080, res / 2], [0, 0, 1]])
    true = dict(betas=betas.reshape(1, 10).astype(np.float32).copy(), body_pose=poses[:n_frames, 3:].copy(), global_orient=poses[:n_frames, :3].copy(),
                transl=tr[:n_frames].copy())
    init = {k: v.copy() for k, v in true.items()}
    init["body_pose"] += rng.randn(*init["body_pose"].shape).astype(np.float32) * noise
    init["transl"] += rng.randn(*init["transl"].shape).astype(np.float32) * noise * 0.5
    if blendshapes:
        init["betas"] += rng.randn(1, 10).astype(np.float32) * 0.1
    frames = DeviceFrames(torch.stack(imgs), torch.stack(masks), K, np.eye(4), init, PatchSampler(num_patch=4, patch_size=patch, ratio_mask=1))
    return frames, SMPL.from_dict(body).to(device), true


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--synthetic", action="store_true", help="frames rendered from the synthetic avatar, perturbed initial SMPL parameters")
    ap.add_argument("--frames", help="npz of a pre-decoded sequence (drivers.train --frames: images uint8 [N,H,W,3], masks, K, (c2w), betas, "
                                     "global_orient, body_pose, transl = the initial SMPL parameters, e.g. the output of a pose estimator)")
    ap.add_argument("--smpl-dir", default="./data/SMPLX/smpl")
    ap.add_argument("--gender", default="neutral")
    ap.add_argument("--synthetic-body", action="store_true", help="--frames: the synthetic SMPL-like body instead of a SMPL pickle from --smpl-dir")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--res", type=int, default=128)
    ap.add_argument("--out", default="outputs/fit")
    args = ap.parse_args(argv)
    if bool(args.synthetic) == bool(args.frames):
        ap.error("exactly one of --synthetic / --frames <npz> is required")
    from .launch import Launch
    launch = Launch.from_env(who="fit")
    try:
        device = launch.device
        if args.frames:
            # confs/SNARF_NGP_fitting.yaml: deformer=smpl, sampler=patch -- the sequence from pre-decoded arrays (see drivers/train.py)
            from . import config as cfg
            from .train import load_frames
            from ..deformers.smplx import _abs_smpl
            confs = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "confs")
            frames = load_frames(args.frames, cfg.instantiate(cfg.load_group(confs, "sampler", "patch", {})), device)
            body_model = (SMPL.from_dict(synthetic.make_body()) if args.synthetic_body else SMPL(_abs_smpl(args.smpl_dir), gender=args.gender)).to(device)
        else:
            frames, body_model, _ = synthetic_frames(device, res=args.res)
        model = build_fit_model(frames, body_model, device)
        losses = fit_sequence(model, frames, args.steps, rank=launch.rank, world_size=launch.world_size, log_every=50 if launch.is_main else 0)
        if launch.is_main:     # replicas are identical: rank 0 exports
            path = export_params(model, args.out)
            print("saved %s (mse %.5f, %d rank(s))" % (path, float(losses["mse_loss"]), launch.world_size))
        launch.barrier()
    finally:
        launch.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
