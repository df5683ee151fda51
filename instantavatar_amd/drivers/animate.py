"""`animate.py` without Lightning / Hydra (reference animate.py:13-118): render a novel-pose sequence
with a trained (or synthetic) avatar and write PNG frames + a GIF.

    python -m instantavatar_amd.drivers.animate --poses data/animation/aist_demo.npz \\
        --ckpt checkpoints/last.ckpt --smpl-dir ./data/SMPLX/smpl --gender male --out animation/aist_demo
    python -m instantavatar_amd.drivers.animate --synthetic --max-frames 8 --downscale 8 --out /tmp/anim

Camera, ray construction and the pose-track conventions are the reference's AnimateDataset
(animate.py:13-80): 1080x1080 pinhole with f = 2000, c2w = I, `trans - trans[0] + (0, 0.15, 5)`,
near/far = |transl| -+ 1.  Frames are replayed from one captured HIP graph (pipeline.GraphedRenderer)."""
import argparse
import os
import sys

import numpy as np
import torch

from .. import synthetic
from ..pipeline import AvatarModel, GraphedRenderer, build_synthetic_model
from . import checkpoint as ckpt_io
from . import config as cfg


def make_rays(K, c2w, H, W):
    """animate.py:13-25"""
    x, y = np.meshgrid(np.arange(W), np.arange(H), indexing="xy")
    xy = np.stack([x, y, np.ones_like(x)], axis=-1).reshape(-1, 3).astype(np.float32)
    d_c = xy @ np.linalg.inv(K).T
    d_w = d_c @ c2w[:3, :3].T
    d_w = d_w / np.linalg.norm(d_w, axis=1, keepdims=True)
    o_w = np.tile(c2w[:3, 3], (len(d_w), 1))
    return o_w.astype(np.float32), d_w.astype(np.float32)


class AnimateSequence:
    """AnimateDataset (animate.py:27-80) as device-resident batches."""

    def __init__(self, poses72, trans, betas, device, downscale=2):
        H = W = 1080
        K = np.eye(3)
        K[0, 0] = K[1, 1] = 2000
        K[0, 2] = H // 2
        K[1, 2] = W // 2
        if downscale > 1:
            H, W = H // downscale, W // downscale
            K[:2] /= downscale
        self.H, self.W = H, W
        o, d = make_rays(K, np.eye(4), H, W)
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device)
        self.rays_o, self.rays_d = t(o)[None], t(d)[None]
        self.thetas = t(poses72[..., :72])
        self.transl = t(trans - trans[0:1] + np.array([0, 0.15, 5], np.float32))
        self.betas = t(betas).reshape(1, 10)

    def __len__(self):
        return self.transl.shape[0]

    def batch(self, idx):
        dist = torch.sqrt((self.transl[idx] ** 2).sum())
        ones = torch.ones(1, self.rays_d.shape[1], device=self.rays_d.device)
        return {"rays_o": self.rays_o, "rays_d": self.rays_d, "betas": self.betas,
                "global_orient": self.thetas[idx:idx + 1, :3], "body_pose": self.thetas[idx:idx + 1, 3:],
                "transl": self.transl[idx:idx + 1], "near": ones * (dist - 1), "far": ones * (dist + 1)}


def build_model(args, device):
    if args.synthetic:
        model, _, _ = build_synthetic_model(device)
        if args.ckpt:  # synthetic body, trained weights (e.g. written by drivers.train --synthetic)
            missing, unexpected = ckpt_io.load_checkpoint(model, args.ckpt, map_location=device)
            print("checkpoint %s loaded (step %d)" % (args.ckpt, model.global_step))
        return model, np.zeros(10, np.float32)
    deformer, net, renderer = cfg.build_plugins(args.confs, args.deformer, args.network, args.renderer, gender=args.gender,
                                                deformer_kwargs=dict(model_path=args.smpl_dir))
    model = AvatarModel(deformer, net, renderer).to(device)
    renderer.initialize(1)
    betas = np.zeros(10, np.float32)
    if args.betas:
        betas = np.load(args.betas)["betas"].reshape(-1)[:10].astype(np.float32)
    missing, unexpected = ckpt_io.load_checkpoint(model, args.ckpt, map_location=device)
    print("checkpoint %s: %d tensors not on the path ignored, %d own tensors kept at init" % (args.ckpt, len(unexpected), len(missing)))
    if not getattr(deformer, "initialized", False):
        deformer.initialize(torch.as_tensor(betas, device=device).reshape(1, 10), device)
        deformer.initialized = True
    # The reference's animate.py never calls net.initialize: `center` / `scale` come from the checkpoint
    # (buffers written by training, DNeRF.py:134).  Only a checkpoint without them falls back to the bbox.
    if "net_coarse.center" in missing or "net_coarse.scale" in missing:
        net.initialize(deformer.bbox)
    else:
        net.bbox = deformer.bbox    # attribute only (NeRFNGPNet.initialize is a no-op once `bbox` exists)
    return model, betas


def write_frames(frames, out_dir, gif=None):
    """8-bit [H, W, 4] frames in the MODEL's channel order -> `<i>.png` (+ optionally one GIF) with the colours the reference's
    files have.  The reference writes with cv2.imwrite (animate.py:113), which takes channel 0 as BLUE: the model's channels are
    in the (B, G, R) order of the cv2.imread training images (peoplesnapshot.py:100), so its files hold the intended colours;
    for the GIF it converts BGRA -> RGBA first (:115).  PIL takes channel 0 as RED: the same reorder serves both."""
    from PIL import Image
    os.makedirs(out_dir, exist_ok=True)
    frames = [np.ascontiguousarray(np.asarray(f)[..., [2, 1, 0, 3]]) for f in frames]
    for i, f in enumerate(frames):
        Image.fromarray(f, "RGBA").save(os.path.join(out_dir, "%d.png" % i))
    if gif and frames:
        ims = [Image.fromarray(f, "RGBA") for f in frames]
        ims[0].save(os.path.join(out_dir, gif), save_all=True, append_images=ims[1:], duration=33, loop=0)


def render_sequence(model, seq, out_dir, gif="animation.gif"):
    """animate.py:104-118 / novel_view.py:117-127: every batch of `seq` through render_image_fast (replayed from one captured
    HIP graph), RGBA = [rgb, alpha] * 255 as 8-bit PNGs `<i>.png`, optionally a GIF of all frames.  Returns the frame count."""
    size = (seq.H, seq.W)
    renderer = GraphedRenderer(model, seq.batch(0), size)
    frames = []
    with torch.inference_mode():
        for i in range(len(seq)):
            rgb, _, alpha, _ = renderer(seq.batch(i))
            img = torch.cat([rgb, alpha[..., None]], dim=-1)[0]
            frames.append((img.clamp(0, 1) * 255).to(torch.uint8).cpu().numpy())     # animate.py:109-113
    renderer.finish()
    for i in renderer.incomplete_calls:  # frames whose loop needed more iterations than the graph holds
        rgb, _, alpha, _ = model.render_image_fast(seq.batch(i), size)
        frames[i] = (torch.cat([rgb, alpha[..., None]], dim=-1)[0].clamp(0, 1) * 255).to(torch.uint8).cpu().numpy()
    write_frames(frames, out_dir, gif)
    return len(frames)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--poses", help="npz with `poses` [n,>=72] and `trans` [n,3] (data/animation/aist_demo.npz)")
    ap.add_argument("--ckpt", help="Lightning checkpoint of DNeRFModel")
    ap.add_argument("--betas", help="npz with `betas` (the subject's anim_nerf_train.npz)")
    ap.add_argument("--smpl-dir", default="./data/SMPLX/smpl")
    ap.add_argument("--gender", default="neutral")
    ap.add_argument("--confs", default=os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "confs"))
    ap.add_argument("--deformer", default="fast_snarf")
    ap.add_argument("--network", default="ngp")
    ap.add_argument("--renderer", default="raymarcher_acc")
    ap.add_argument("--synthetic", action="store_true", help="synthetic SMPL-like body + field (no SMPL pickle / checkpoint needed)")
    ap.add_argument("--downscale", type=int, default=2)
    ap.add_argument("--max-frames", type=int, default=0)
    ap.add_argument("--out", default="animation/out")
    ap.add_argument("--no-gif", action="store_true")
    args = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("animate: needs a GPU (the product path has no CPU fallback)")
    if not args.synthetic and not (args.ckpt and args.poses):
        ap.error("--ckpt and --poses are required unless --synthetic is given")
    device = torch.device("cuda", 0)
    model, betas = build_model(args, device)
    model.eval()
    if args.poses:
        z = np.load(args.poses)
        poses, trans = z["poses"].astype(np.float32), z["trans"].astype(np.float32)
    else:
        poses, trans = synthetic.procedural_pose_track(64)
    if args.max_frames:
        poses, trans = poses[:args.max_frames], trans[:args.max_frames]
    seq = AnimateSequence(poses, trans, betas, device, args.downscale)
    n = render_sequence(model, seq, args.out, gif=None if args.no_gif else "animation.gif")
    print("wrote %d frames (%dx%d) to %s" % (n, seq.W, seq.H, args.out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
