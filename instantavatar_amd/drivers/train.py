"""`train.py` without Lightning / Hydra, for the part of it that lies on the hot path: the optimisation
loop of DNeRFModel.training_step (DNeRF.py:112-161) over per-frame ray batches, with Lightning-layout
checkpoints, followed by one validation_step (DNeRF.py:171-188).  Image files are not read by this package (the samplers
and `__getitem__` run on device-resident frames: datasets.DeviceFrames); the loop takes any iterable of batches with the
reference's keys (`rays_o`, `rays_d`, `near`, `far`, `rgb`, `alpha`, `bg_color`, SMPL parameters).  `--synthetic` supplies one: targets rendered
from the synthetic field, 4 096 random rays of a random frame per step (what bench.py times).  `--frames <npz>` trains on a
PRE-DECODED sequence -- the arrays peoplesnapshot.py:99-151 reads per item from image files, once: `images` uint8 [N,H,W,3]
(as cv2.imread returns them, at the training resolution), `masks` [N,H,W], `K` [3,3], optional `c2w` [4,4], and the SMPL
parameters of poses/anim_nerf_train.npz (`betas` [1,10], `global_orient` [N,3], `body_pose` [N,69], `transl` [N,3]) -- through
datasets.DeviceFrames and the confs/sampler group, with the plugins built from confs/ like the reference's Hydra run.

    python -m instantavatar_amd.drivers.train --synthetic --steps 200 --ckpt /tmp/avatar/last.ckpt
    python -m instantavatar_amd.drivers.train --frames seq.npz --smpl-dir ./data/SMPLX/smpl --gender male --sampler patch --steps 3000
"""
import argparse
import os
import sys
import time

import torch

from .. import synthetic
from ..pipeline import build_synthetic_model, make_batch
from ..training import GraphedTrainStep, NeRFLoss, configure_optimizer, configure_scheduler
from . import checkpoint as ckpt_io


def synthetic_batches(device, teacher, res=256, n_frames=8, n_rays=4096, seed=1234, rank=0, world_size=1):
    """Endless iterator of training batches: frames rendered once by `teacher`, random rays per step.
    Rank r of a `world_size`-rank job takes frames r, r + W, r + 2W, ... (one frame + its ray batch per rank and step,
    SURVEY.md 8e) and draws its rays from its own generator (seed + r)."""
    poses, tr = synthetic.procedural_pose_track(max(n_frames, 8))
    targets = []
    with torch.no_grad():
        for f in range(n_frames):
            b = make_batch(device, res, poses[f], tr[f])
            rgb, _, alpha, _ = teacher.render_image_fast(b, (res, res))
            targets.append((b, rgb.reshape(1, -1, 3), alpha.reshape(1, -1)))
    g = torch.Generator(device=device).manual_seed(seed + rank)
    i = rank
    while True:
        b, rgb, alpha = targets[i % n_frames]
        sel = torch.randint(0, res * res, (n_rays,), device=device, generator=g)
        batch = dict(b)
        for k in ("rays_o", "rays_d", "near", "far"):
            batch[k] = b[k][:, sel]
        batch["rgb"], batch["alpha"] = rgb[:, sel], alpha[:, sel]
        batch["bg_color"] = torch.ones_like(batch["rgb"])
        yield batch
        i += world_size


def frame_batches(frames, seed=42, rank=0, world_size=1):
    """Endless iterator over a DeviceFrames sequence the way the reference's DataLoader(shuffle=True, batch_size=1) walks it:
    a fresh permutation of the frames per epoch (the same on every rank: pl.seed_everything), rank r taking positions r, r + W, ...;
    sampler draws from the rank's own device generator."""
    g_perm = torch.Generator().manual_seed(seed)
    g_dev = torch.Generator(device=frames.images.device).manual_seed(seed + 1 + rank)
    n = len(frames)
    while True:
        order = torch.randperm(n, generator=g_perm).tolist()
        for k in range(-(-n // world_size)):     # every rank takes the same number of steps per epoch (the last one wraps around)
            yield frames.batch(order[(k * world_size + rank) % n], generator=g_dev)


def load_frames(path, sampler, device):
    """`--frames`: a pre-decoded sequence (see the module docstring) -> datasets.DeviceFrames"""
    import numpy as np
    from ..datasets.device_frames import DeviceFrames
    z = np.load(path)
    need = ("images", "masks", "K", "betas", "global_orient", "body_pose", "transl")
    missing = [k for k in need if k not in z.files]
    if missing:
        raise SystemExit("--frames %s: missing arrays %s (have %s)" % (path, missing, z.files))
    imgs, masks = z["images"], z["masks"]
    if imgs.dtype != np.uint8 or imgs.ndim != 4 or imgs.shape[-1] != 3 or masks.shape != imgs.shape[:3]:
        raise SystemExit("--frames: images must be uint8 [N,H,W,3] and masks [N,H,W] (got %s %s, %s)" % (imgs.dtype, imgs.shape, masks.shape))
    n = imgs.shape[0]
    smpl = dict(betas=z["betas"].reshape(1, 10).astype(np.float32), global_orient=z["global_orient"].reshape(n, 3).astype(np.float32),
                body_pose=z["body_pose"].reshape(n, 69).astype(np.float32), transl=z["transl"].reshape(n, 3).astype(np.float32))
    c2w = z["c2w"] if "c2w" in z.files else np.eye(4)
    return DeviceFrames.from_arrays(imgs, masks, z["K"], c2w, smpl, sampler, device)


def synthetic_val_batch(device, teacher, res=256, frame=0):
    """A whole frame with its target image: what the "val" split's __getitem__ yields (peoplesnapshot.py:112-125)."""
    poses, tr = synthetic.procedural_pose_track(8)
    b = make_batch(device, res, poses[frame], tr[frame])
    with torch.no_grad():
        rgb, _, alpha, _ = teacher.render_image_fast(dict(b), (res, res))
    b["rgb"], b["alpha"] = rgb.reshape(1, -1, 3), alpha.reshape(1, -1)
    return b


def fit(model, batches, steps, optimizer=None, loss_fn=None, log_every=50, out=sys.stdout, scheduler=None,
        steps_per_epoch=None, max_epochs=None, world_size=1, graphed=True, check_val_every_n_epoch=10, on_validation=None):
    """`steps` iterations of training_step; returns (last losses (device tensors), optimizer, scheduler).
    The reference steps its LambdaLR `(1 - k / max_epochs) ** 1.5` in `on_validation_epoch_end` (DNeRF.py:163-166; manual
    optimisation, so Lightning does not step it): once per validation run, every `check_val_every_n_epoch` epochs.  Pass
    `steps_per_epoch` (= frames of the sequence) and `max_epochs` to get the same decay; `on_validation(model)` is called at
    those epochs before the scheduler steps (validation_step, DNeRF.py:171-188).
    graphed: the step is replayed from a captured HIP graph (training.GraphedTrainStep; it runs the occupancy-update steps
    eagerly and falls back to eager steps altogether when capture is not possible).
    world_size > 1 (one process per GPU, every rank calls this with ITS batches): replicas are made identical to rank 0
    first, every step averages the gradients over RCCL (bucketed, started from inside the backward; captured into the graph
    with the kernels), the occupancy update MAX-reduces the cached densities.  An epoch is `steps_per_epoch` steps of the
    JOB (pass frames / world_size); `on_validation` runs on every rank that passes one (pass it on rank 0 only)."""
    from ..parallel import broadcast_module_state
    broadcast_module_state(model, world_size)   # replicas equal rank 0's (parameters, buffers, occupancy caches), not "equal by seed"
    optimizer = optimizer or configure_optimizer(model)
    if scheduler is None and steps_per_epoch and max_epochs:
        scheduler = configure_scheduler(optimizer, max_epochs)
    loss_fn = loss_fn or NeRFLoss(dict(w_rgb=1.0, w_alpha=0.1, w_reg=0.1))
    model.train()
    t0 = time.perf_counter()
    losses = None
    step = GraphedTrainStep(model, optimizer, loss_fn, world_size=world_size, enabled=graphed)
    for it, batch in zip(range(steps), batches):
        losses = step(batch)
        if steps_per_epoch and model.global_step % steps_per_epoch == 0 and (model.global_step // steps_per_epoch) % check_val_every_n_epoch == 0:
            if on_validation is not None:
                on_validation(model)
                model.train()
            if scheduler is not None:
                scheduler.step()
        if log_every and (it + 1) % log_every == 0:
            torch.cuda.synchronize()
            print("step %d  loss %.5f  mse %.5f  lr %.2e  %.0f it/s" % (
                model.global_step, float(losses["loss"].detach()), float(losses["mse_loss"].detach()),
                float(optimizer.param_groups[0]["lr"]), (it + 1) / (time.perf_counter() - t0)), file=out)
    return losses, optimizer, scheduler


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--synthetic", action="store_true",
                    help="synthetic SMPL-like body and targets (the only data source shipped with this package)")
    ap.add_argument("--frames", help="npz of a pre-decoded sequence: images uint8 [N,H,W,3], masks [N,H,W], K [3,3], (c2w [4,4]), betas, "
                                     "global_orient, body_pose, transl (see the module docstring)")
    ap.add_argument("--sampler", default="patch", help="confs/sampler group for --frames: patch (SNARF_NGP.yaml) or edge (SNARF_NGP_refine.yaml)")
    ap.add_argument("--smpl-dir", default="./data/SMPLX/smpl")
    ap.add_argument("--gender", default="neutral")
    ap.add_argument("--synthetic-body", action="store_true", help="--frames: the synthetic SMPL-like body instead of a SMPL pickle from --smpl-dir")
    ap.add_argument("--confs", default=os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "confs"))
    ap.add_argument("--deformer", default="fast_snarf")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--ckpt", default="checkpoints/last.ckpt")
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--steps-per-epoch", type=int, default=8, help="frames per epoch")
    ap.add_argument("--max-epochs", type=int, default=30, help="confs/SNARF_NGP.yaml train.max_epochs (= scheduler.max_epochs)")
    ap.add_argument("--check-val-every-n-epoch", type=int, default=10,
                    help="confs/SNARF_NGP.yaml train.check_val_every_n_epoch: a validation_step and ONE step of the LR schedule every that many epochs")
    args = ap.parse_args(argv)
    if bool(args.synthetic) == bool(args.frames):
        ap.error("exactly one of --synthetic / --frames <npz> is required")
    from .launch import Launch
    launch = Launch.from_env(who="train")
    try:
        return _run(args, launch)
    finally:
        launch.close()


def _run(args, launch):
    """train.py:27-41 for rank `launch.rank` of `launch.world_size` (the reference is `pl.Trainer(gpus=1)`, train.py:29-30)."""
    device, world, main = launch.device, launch.world_size, launch.is_main
    say = print if main else (lambda *a, **k: None)
    frames = None
    if args.frames:
        # DNeRFModel.__init__ from the conf groups (DNeRF.py:22-28) + the datamodule's trainset as device-resident frames
        from . import config as cfg
        from ..pipeline import AvatarModel
        sampler = cfg.instantiate(cfg.load_group(args.confs, "sampler", args.sampler, {}))
        frames = load_frames(args.frames, sampler, device)
        kw = dict(model_path=args.smpl_dir)
        if args.synthetic_body:
            from ..deformers.smplx import SMPL
            kw = dict(body_model=SMPL.from_dict(synthetic.make_body()).to(device))
        deformer, net, renderer = cfg.build_plugins(args.confs, args.deformer, gender=args.gender, deformer_kwargs=kw)
        model = AvatarModel(deformer, net, renderer).to(device)
        renderer.initialize(len(frames))
        deformer.initialize(frames.smpl_params["betas"][:1], device)
        deformer.initialized = True
        args.res = frames.H
        args.steps_per_epoch = len(frames)
        teacher = None
        say("%d frames %dx%d from %s, sampler %s" % (len(frames), frames.W, frames.H, args.frames, type(sampler).__name__))
    else:
        teacher, _, _ = build_synthetic_model(device)
        model, _, _ = build_synthetic_model(device)
    model.net_coarse.reset_parameters()
    opt = configure_optimizer(model)
    sched = configure_scheduler(opt, args.max_epochs)
    if args.resume and os.path.exists(args.ckpt):
        # parameters, buffers, global_step AND the optimiser moments / LR-scheduler epoch: a resumed run
        # continues the interrupted one instead of restarting Adam from zero moments (every rank reads the same file)
        ckpt_io.load_checkpoint(model, args.ckpt, map_location=device, optimizer=opt, scheduler=sched)
        say("resumed from %s at step %d (lr %.2e)" % (args.ckpt, model.global_step, float(opt.param_groups[0]["lr"])))
    from ..evaluation import validation_step
    val_batch = frames.frame(0) if frames is not None else synthetic_val_batch(device, teacher, res=args.res)
    val_size = (frames.H, frames.W) if frames is not None else (args.res, args.res)

    def validate(m):
        m.eval()
        val = validation_step(m, dict(val_batch), val_size)                                                # DNeRF.py:171-188
        print("step %d  val/rgb_loss %.6f  val/counter_avg %.2f  val/counter_max %.0f" % (m.global_step, float(val["rgb_loss"]), float(val["counter_avg"]),
                                                                                        float(val["counter_max"])))
    # W ranks consume W frames per step: an epoch (one pass over the frames) is ceil(frames / W) steps.  The learning-rate
    # schedule is per epoch, so it is unchanged; the global batch is W x 4 096 rays (the reference has no multi-GPU mode to
    # compare with: DESIGN.md section 6)
    steps_per_epoch = -(-args.steps_per_epoch // world)
    if frames is not None:
        batches = frame_batches(frames, rank=launch.rank, world_size=world)
    else:
        batches = synthetic_batches(device, teacher, res=args.res, n_frames=max(args.steps_per_epoch, 1), rank=launch.rank, world_size=world)
    losses, opt, sched = fit(model, batches, args.steps, optimizer=opt, scheduler=sched, log_every=50 if main else 0,
                             steps_per_epoch=steps_per_epoch, max_epochs=args.max_epochs, world_size=world,
                             check_val_every_n_epoch=args.check_val_every_n_epoch, on_validation=validate if main else None)
    if main:
        validate(model)
        os.makedirs(os.path.dirname(os.path.abspath(args.ckpt)), exist_ok=True)
        ckpt_io.save_checkpoint(model, args.ckpt, optimizer=opt, scheduler=sched, epoch=sched.last_epoch)   # replicas are identical: rank 0's is THE state
        print("saved %s (step %d, mse %.5f, %d rank(s))" % (args.ckpt, model.global_step, float(losses["mse_loss"]), world))
    launch.barrier()
    return 0


if __name__ == "__main__":
    sys.exit(main())
