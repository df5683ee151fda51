"""`train.py` without Lightning / Hydra, for the part of it that lies on the hot path: the optimisation
loop of DNeRFModel.training_step (DNeRF.py:112-161) over per-frame ray batches, with Lightning-layout
checkpoints, followed by one validation_step (DNeRF.py:171-188).  Image files are not read by this package (the samplers
and `__getitem__` run on device-resident frames: datasets.DeviceFrames); the loop takes any iterable of batches with the
reference's keys (`rays_o`, `rays_d`, `near`, `far`, `rgb`, `alpha`, `bg_color`, SMPL parameters).  `--synthetic` supplies one: targets rendered
from the synthetic field, 4 096 random rays of a random frame per step (what bench.py times).

    python -m instantavatar_amd.drivers.train --synthetic --steps 200 --ckpt /tmp/avatar/last.ckpt
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
    ap.add_argument("--synthetic", action="store_true", required=True,
                    help="synthetic SMPL-like body and targets (the only data source shipped with this package)")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--ckpt", default="checkpoints/last.ckpt")
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--steps-per-epoch", type=int, default=8, help="frames per epoch")
    ap.add_argument("--max-epochs", type=int, default=30, help="confs/SNARF_NGP.yaml train.max_epochs (= scheduler.max_epochs)")
    ap.add_argument("--check-val-every-n-epoch", type=int, default=10,
                    help="confs/SNARF_NGP.yaml train.check_val_every_n_epoch: a validation_step and ONE step of the LR schedule every that many epochs")
    args = ap.parse_args(argv)
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
    val_batch = synthetic_val_batch(device, teacher, res=args.res)

    def validate(m):
        m.eval()
        val = validation_step(m, dict(val_batch), (args.res, args.res))                                    # DNeRF.py:171-188
        print("step %d  val/rgb_loss %.6f  val/counter_avg %.2f  val/counter_max %.0f" % (m.global_step, float(val["rgb_loss"]), float(val["counter_avg"]),
                                                                                        float(val["counter_max"])))
    # W ranks consume W frames per step: an epoch (one pass over the frames) is ceil(frames / W) steps.  The learning-rate
    # schedule is per epoch, so it is unchanged; the global batch is W x 4 096 rays (the reference has no multi-GPU mode to
    # compare with: DESIGN.md section 6)
    steps_per_epoch = -(-args.steps_per_epoch // world)
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
