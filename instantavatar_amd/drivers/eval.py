"""`eval.py` without Lightning / Hydra (reference eval.py:37-118): refine the SMPL parameters of the TEST frames with
everything else frozen (confs/SNARF_NGP_refine.yaml: EdgeSampler, NGPLoss, Adam lr 1e-5 on the SMPL tables, 20 epochs), render
the test frames with the refined parameters, write `test/<i>.png` = [ground truth | rendering | error map], and measure
PSNR / SSIM / LPIPS(alex) on the written 8-bit images -> `results.txt`.

    python -m instantavatar_amd.drivers.eval --synthetic --frames 4 --res 128 --epochs 2 --out /tmp/eval

`--synthetic`: the subject is the synthetic SMPL-like body + field; the "test images" are its renderings at the true poses,
the checkpoint under evaluation has the same field, and its SMPL tables start `--pose-noise` rad / `--transl-noise` m off
(the situation eval.py is written for: a trained field, test-frame poses that still have to be fitted).  With `--ckpt`
the field comes from a Lightning-layout checkpoint, `SMPL_param.*` entries skipped as eval.py:64-67 does.
LPIPS needs the two pretrained weight files (`--lpips-lin third_parties/lpips/weights/v0.1/alex.pth`, `--lpips-trunk` = a saved
torchvision `alexnet().features.state_dict()`); without them the LPIPS line is omitted rather than computed on random features."""
import argparse
import os
import sys

import numpy as np
import torch

from .. import evaluation as ev
from .. import synthetic
from ..pipeline import build_synthetic_model, make_batch
from ..training import GraphedTrainStep, NGPLoss, configure_optimizer, configure_scheduler
from . import checkpoint as ckpt_io


def synthetic_test_set(device, teacher, res, n_frames, pose_noise, transl_noise, seed=7):
    """Test frames of the synthetic subject (8-bit images in the model's channel order + masks, as a data set on disk would
    hold them), the camera, the TRUE SMPL parameters and a perturbed copy of them (the tables to refine)."""
    from ..datasets.device_frames import DeviceFrames
    from ..utils.sampler import EdgeSampler
    poses, tr = synthetic.procedural_pose_track(max(n_frames, 8))
    imgs, masks = [], []
    with torch.no_grad():
        for f in range(n_frames):
            rgb, _, alpha, _ = teacher.render_image_fast(make_batch(device, res, poses[f], tr[f]), (res, res))
            imgs.append(ev.to_u8(rgb[0].clamp(0, 1) * 255))
            masks.append((alpha[0] > 0.5).float())
    true = dict(betas=np.zeros((1, 10), np.float32), body_pose=poses[:n_frames, 3:].copy(), global_orient=poses[:n_frames, :3].copy(),
                transl=tr[:n_frames].copy())
    rs = np.random.RandomState(seed)
    start = dict(true, body_pose=(true["body_pose"] + pose_noise * rs.randn(n_frames, 69)).astype(np.float32),
                 global_orient=(true["global_orient"] + pose_noise * rs.randn(n_frames, 3)).astype(np.float32),
                 transl=(true["transl"] + transl_noise * rs.randn(n_frames, 3)).astype(np.float32))
    K = np.array([[2000.0 * res / 1080, 0, res / 2], [0, 2000.0 * res / 1080, res / 2], [0, 0, 1]])
    sampler = EdgeSampler(num_sample=4096, ratio_mask=0.6, ratio_edge=0.3, kernel_size=16)          # confs/sampler/edge.yaml
    frames = DeviceFrames(torch.stack(imgs), torch.stack(masks), K, np.eye(4), start, sampler)
    return frames, true, start


def refine_test_frames(model, frames, epochs, smpl_lr=1e-5, seed=42, log=None, check_val_every_n_epoch=10):
    """eval.py:70-91 (`trainer.fit` on the test split with only `SMPL_param` trainable): `epochs` passes over the frames,
    training_step with is_refine; every `check_val_every_n_epoch` epochs a validation_step on the first frame and ONE step of
    the LR schedule (DNeRF.py:163-188; training.configure_scheduler).  Returns the number of steps taken."""
    n_train = ev.freeze_all_but_smpl(model)
    if n_train == 0:
        raise ValueError("refine_test_frames: the model has no SMPL_param tables")
    model.is_refine = True
    opt = configure_optimizer(model, smpl_lr=smpl_lr)
    sched = configure_scheduler(opt, epochs)
    loss_fn = NGPLoss(dict(w_rgb=1.0, w_alpha=0.1, w_reg=0.1))
    model.train()
    stepper = GraphedTrainStep(model, opt, loss_fn, is_refine=True)
    g = torch.Generator(device=frames.images.device).manual_seed(seed)
    steps = 0
    for epoch in range(epochs):
        for i in torch.randperm(len(frames)).tolist():                  # DataLoader(shuffle=True) of the train split
            out = stepper(frames.batch(i, generator=g, out=stepper.inputs))
            steps += 1
        if (epoch + 1) % check_val_every_n_epoch == 0:
            model.eval()
            b = frames.frame(0)
            b["idx"] = b["idx"].to(frames.images.device)
            val = ev.validation_step(model, b, (frames.H, frames.W))
            model.train()
            sched.step()
            if log is not None:
                print("epoch %d  val/rgb_loss %.6f  val/counter_avg %.2f" % (epoch, float(val["rgb_loss"]), float(val["counter_avg"])), file=log)
        if log is not None:
            print("epoch %d  loss %.5f  lr(SMPL) %.2e" % (epoch, float(out["loss"].detach()), float(opt.param_groups[2]["lr"])), file=log)
    model.eval()
    return steps


def render_test_images(model, frames, out_dir):
    """`trainer.test` (DNeRF.py:226-239): one [gt | rendering | error map] PNG per test frame."""
    os.makedirs(out_dir, exist_ok=True)
    size = (frames.H, frames.W)
    for i in range(len(frames)):
        b = frames.frame(i)
        b["idx"] = b["idx"].to(frames.images.device)
        ev.write_png_bgr(os.path.join(out_dir, "%d.png" % i), ev.test_step(model, b, size))
    return len(frames)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--synthetic", action="store_true", required=True,
                    help="synthetic SMPL-like body and test frames (the only data source shipped with this package)")
    ap.add_argument("--ckpt", help="Lightning-layout checkpoint of the field to evaluate (default: the synthetic field itself)")
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--epochs", type=int, default=20, help="confs/SNARF_NGP_refine.yaml train.max_epochs")
    ap.add_argument("--check-val-every-n-epoch", type=int, default=10, help="confs/SNARF_NGP_refine.yaml train.check_val_every_n_epoch")
    ap.add_argument("--smpl-lr", type=float, default=1e-5, help="optimize_SMPL.lr")
    ap.add_argument("--pose-noise", type=float, default=0.03)
    ap.add_argument("--transl-noise", type=float, default=0.01)
    ap.add_argument("--lpips-lin", help="third_parties/lpips/weights/v0.1/alex.pth of a reference checkout")
    ap.add_argument("--lpips-trunk", help="a saved torchvision alexnet().features.state_dict()")
    ap.add_argument("--out", default="eval_out")
    args = ap.parse_args(argv)
    from .launch import Launch
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # the refinement steps ONE optimiser over the test frames in sequence (eval.py:70-97) and the metrics are means over
        # all files: a few seconds of work that does not shard without changing Adam's step counts -- one process
        raise SystemExit("eval: a single-process driver (start it without torch.distributed.run)")
    device = Launch.from_env(who="eval").device     # cuda:LOCAL_RANK
    torch.manual_seed(42)
    teacher, _, _ = build_synthetic_model(device)
    model, _, _ = build_synthetic_model(device)
    if args.ckpt:
        missing, unexpected = ckpt_io.load_checkpoint(model, args.ckpt, map_location=device, skip_prefixes=("SMPL_param",), strict_self_check=True)
        print("checkpoint %s loaded (step %d), SMPL_param entries skipped" % (args.ckpt, model.global_step))
    frames, true, start = synthetic_test_set(device, teacher, args.res, args.frames, args.pose_noise, args.transl_noise)
    from ..models.structures.body_model_param import SMPLParamEmbedding
    model.SMPL_param = SMPLParamEmbedding(**{k: torch.as_tensor(v.copy()) for k, v in start.items()}).to(device)
    field_before = [p.detach().clone() for n, p in model.named_parameters() if not n.startswith("SMPL_param")]
    steps = refine_test_frames(model, frames, args.epochs, smpl_lr=args.smpl_lr, log=sys.stdout, check_val_every_n_epoch=args.check_val_every_n_epoch)
    frozen = all(torch.equal(a, p.detach()) for a, (n, p) in zip(field_before, ((n, p) for n, p in model.named_parameters() if not n.startswith("SMPL_param"))))
    if not frozen:
        raise RuntimeError("eval: a parameter outside SMPL_param changed during the refinement (eval.py:70-73 freezes them)")
    moved = {k: float((getattr(model.SMPL_param, k).weight.detach().cpu() - torch.as_tensor(start[k])).abs().max()) for k in ("body_pose", "global_orient", "transl")}
    print("refined %d frames in %d steps (field parameters untouched); largest change of the tables: %s" % (len(frames), steps, moved))
    if max(moved.values()) == 0.0:
        raise RuntimeError("eval: the SMPL tables did not move -- no gradient reached them")
    test_dir = os.path.join(args.out, "test")
    n = render_test_images(model, frames, test_dir)
    lp = None
    if args.lpips_lin and args.lpips_trunk:
        from ..utils.lpips import LPIPS
        lp = LPIPS(net="alex").load_lin_weights(args.lpips_lin).load_trunk_weights(args.lpips_trunk).to(device)
    elif args.lpips_lin or args.lpips_trunk:
        ap.error("--lpips-lin and --lpips-trunk are both needed for the LPIPS figure")
    from ..utils.metrics import Evaluator
    results, n_eval = ev.evaluate_folder(test_dir, Evaluator(lpips=lp).to(device), device)
    ev.write_results(os.path.join(args.out, "results.txt"), results)
    print("PSNR: %.2f" % results["psnr"])
    print("SSIM: %.4f" % results["ssim"])
    print(("LPIPS: %.4f" % results["lpips"]) if "lpips" in results else "LPIPS: -- (no pretrained weights given)")
    print("wrote %d test images to %s and %s" % (n, test_dir, os.path.join(args.out, "results.txt")))
    return 0


if __name__ == "__main__":
    sys.exit(main())
